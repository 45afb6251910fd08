#!/bin/bash
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
for q in 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/many_envs_batch.py 8 8 2>&1 | grep -v amdgpu.ids; GPU_MAX_HW_QUEUES=$q python tools/many_envs_batch.py 16 8 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2d/queues.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2d/trace8 -o t -- python tools/many_envs_batch.py 8 8 > gpurun_out/r2d/trace8.log 2>&1
python tools/trace_overlap.py $(find gpurun_out/r2d/trace8 -name '*kernel_trace.csv' | head -1)

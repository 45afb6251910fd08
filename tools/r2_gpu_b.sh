#!/bin/bash
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
python tools/many_envs_batch.py 1 1 > gpurun_out/r2b/batch_1.log 2>&1
python tools/many_envs_batch.py 8 8 > gpurun_out/r2b/batch_8.log 2>&1
python tools/many_envs_batch.py 8 4 >> gpurun_out/r2b/batch_8.log 2>&1
python tools/many_envs_batch.py 16 8 >> gpurun_out/r2b/batch_8.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/trace1 -o t -- python tools/many_envs_batch.py 1 1 > gpurun_out/r2b/trace1.log 2>&1
cat gpurun_out/r2b/batch_1.log gpurun_out/r2b/batch_8.log | grep -v amdgpu.ids
f=$(find gpurun_out/r2b/trace1 -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f 5

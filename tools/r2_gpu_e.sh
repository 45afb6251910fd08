#!/bin/bash
mkdir -p gpurun_out/r2e
export OMP_NUM_THREADS=1
timeout 600 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -q -x -k "env_batch or fused_cycle or full_size_inv or host_and_device" > gpurun_out/r2e/t.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r2e/t.log
unset OMP_NUM_THREADS
for a in "1 1" "2 8" "4 8" "8 8" "16 8" "32 8" "64 8" "128 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2e/batch.log
MAGENT_BATCH_CYCLE=0 GPU_MAX_HW_QUEUES=16 python tools/many_envs_batch.py 8 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2e/batch.log

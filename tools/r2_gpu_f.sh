#!/bin/bash
mkdir -p gpurun_out/r2f
export OMP_NUM_THREADS=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x > gpurun_out/r2f/t.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r2f/t.log
unset OMP_NUM_THREADS
python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2f/marks.log
for a in "1 1" "8 8" "32 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2f/batch.log
MAGENT_SOLO_LDS=0 python tools/many_envs_batch.py 1 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2f/batch.log

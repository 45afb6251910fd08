#!/bin/bash
mkdir -p gpurun_out/r2c
export OMP_NUM_THREADS=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_cycle" > gpurun_out/r2c/fused.log 2>&1
echo "fused rc=$?"; tail -15 gpurun_out/r2c/fused.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_fused_cycle_matches_oracle > gpurun_out/r2c/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r2c/pytest.log
unset OMP_NUM_THREADS
for a in "1 1" "8 8" "8 4" "16 8" "32 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2c/batch.log
python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c/marks.log

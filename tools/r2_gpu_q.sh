#!/bin/bash
mkdir -p gpurun_out/r2q
export OMP_NUM_THREADS=1
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2q/pytest.log 2>&1
echo "pytest rc=$?"; tail -14 gpurun_out/r2q/pytest.log
unset OMP_NUM_THREADS
timeout 200 python bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-330
for a in "1 1" "8 8" "32 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2q/batch.log

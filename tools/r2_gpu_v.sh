#!/bin/bash
mkdir -p gpurun_out/r2v
export OMP_NUM_THREADS=1
timeout 600 python tools/gpu_check.py 2>&1 | grep -v "^OK" | tail -4
timeout 300 python tools/fuzz_parity.py oracle hip 0 500 2>/dev/null | tail -2
FUZZ_TURN=2 FUZZ_SECTOR=1 timeout 300 python tools/fuzz_parity.py oracle hip 500 900 2>/dev/null | tail -2
unset OMP_NUM_THREADS
python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2v/marks.log
for a in "1 1" "8 8" "64 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2v/batch.log
timeout 200 python bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-330

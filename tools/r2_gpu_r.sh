#!/bin/bash
mkdir -p gpurun_out/r2r
export OMP_NUM_THREADS=1
timeout 600 python tools/gpu_check.py 2>&1 | grep -v "^OK" | tail -8 | tee gpurun_out/r2r/check.log
timeout 300 python tools/fuzz_parity.py oracle hip 0 600 2>/dev/null | tail -3 | tee gpurun_out/r2r/fuzz.log
FUZZ_TURN=2 timeout 300 python tools/fuzz_parity.py oracle hip 0 400 2>/dev/null | tail -3 | tee -a gpurun_out/r2r/fuzz.log
unset OMP_NUM_THREADS
python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2r/marks.log
MAGENT_SOLO_BATCH=0 python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tail -20 > gpurun_out/r2r/marks_unbatched.log
for a in "1 1" "8 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2r/batch.log

#!/bin/bash
mkdir -p gpurun_out/r2s
export OMP_NUM_THREADS=1
timeout 600 python tools/gpu_check.py battle_small_dense battle_brawl battle_brawl_big battle_turn tri_rect pursuit bodies arrange_live 2>&1 | grep -v "^OK" | tail -4
unset OMP_NUM_THREADS
python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2s/marks.log
MAGENT_SOLO_BATCH=0 python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tail -19 | tee gpurun_out/r2s/marks_unbatched.log
for a in "1 1" "8 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2s/batch.log

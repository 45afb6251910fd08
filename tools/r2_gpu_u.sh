#!/bin/bash
mkdir -p gpurun_out/r2u
export OMP_NUM_THREADS=1
timeout 600 python tools/gpu_check.py battle_small_dense battle_brawl battle_brawl_big battle_grow battle_events tri_rect pursuit bodies arrange_live 2>&1 | grep -v "^OK" | tail -4
unset OMP_NUM_THREADS
python tools/solo_marks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2u/marks.log
for a in "1 1" "8 8"; do python tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2u/batch.log
timeout 200 python bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-330

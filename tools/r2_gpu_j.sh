#!/bin/bash
mkdir -p gpurun_out/r2j
export OMP_NUM_THREADS=1
T="battle_turn bodies_turn pursuit_turn bodies_turn_large arrange_turn tri_turn gather_turn pursuit bodies arrange_live bodies_large"
timeout 600 python tools/gpu_check.py $T 2>&1 | tail -14 | tee gpurun_out/r2j/turn_solo.log
MAGENT_SOLO_STEP=0 timeout 600 python tools/gpu_check.py $T 2>&1 | tail -14 | tee gpurun_out/r2j/turn_multi.log

#!/bin/bash
mkdir -p gpurun_out/r2i
export OMP_NUM_THREADS=1
timeout 600 python tools/gpu_check.py battle_turn battle_turn_large gather_turn tri_turn battle_brawl pursuit bodies 2>&1 | tail -12 | tee gpurun_out/r2i/turn.log

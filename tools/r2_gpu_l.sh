#!/bin/bash
mkdir -p gpurun_out/r2l
export OMP_NUM_THREADS=1
timeout 600 python tools/gpu_check.py sector sector_turn sector_turn_large battle_small_dense 2>&1 | tail -6 | tee gpurun_out/r2l/sector.log
MAGENT_SOLO_STEP=0 timeout 600 python tools/gpu_check.py sector sector_turn 2>&1 | tail -4 | tee -a gpurun_out/r2l/sector.log

#!/bin/bash
mkdir -p gpurun_out/r2m
export OMP_NUM_THREADS=1
FUZZ_SECTOR=1 FUZZ_TURN=2 timeout 900 python tools/fuzz_parity.py oracle hip 0 2000 2>/dev/null | tail -4 | tee gpurun_out/r2m/fuzz_sector_turn.log
FUZZ_SECTOR=1 FUZZ_TURN=2 FUZZ_RULES=2 MAGENT_SOLO_STEP=0 timeout 900 python tools/fuzz_parity.py oracle hip 2000 3000 2>/dev/null | tail -4 | tee gpurun_out/r2m/fuzz_all_multi.log

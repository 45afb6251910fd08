#!/bin/bash
mkdir -p gpurun_out/r2k
export OMP_NUM_THREADS=1
FUZZ_TURN=2 timeout 900 python tools/fuzz_parity.py oracle hip 0 3000 2>/dev/null | tail -4 | tee gpurun_out/r2k/fuzz_turn.log
FUZZ_TURN=2 MAGENT_SOLO_STEP=0 timeout 900 python tools/fuzz_parity.py oracle hip 3000 4500 2>/dev/null | tail -4 | tee gpurun_out/r2k/fuzz_turn_multi.log

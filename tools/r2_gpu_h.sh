#!/bin/bash
mkdir -p gpurun_out/r2h
export OMP_NUM_THREADS=1
FUZZ_RULES=2 timeout 900 python tools/fuzz_parity.py oracle hip 1000 2500 2>/dev/null | tail -3 | tee gpurun_out/r2h/fuzz_rules2.log
timeout 900 python tools/fuzz_parity.py oracle hip 0 1500 2>/dev/null | tail -3 | tee gpurun_out/r2h/fuzz_default.log
timeout 600 python tools/gpu_check.py 2>&1 | tail -4 | tee gpurun_out/r2h/check.log

#!/bin/bash
mkdir -p gpurun_out/r2g
export OMP_NUM_THREADS=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "rules_search or fuzz_rule or driver_variants" > gpurun_out/r2g/t.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r2g/t.log
FUZZ_RULES=2 timeout 900 python tools/fuzz_parity.py oracle hip 1000 1600 2>/dev/null | tail -3 | tee gpurun_out/r2g/fuzz_rules.log

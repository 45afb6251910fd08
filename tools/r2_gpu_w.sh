#!/bin/bash
mkdir -p gpurun_out/r2w
export OMP_NUM_THREADS=1
MAGENT_SOLO_STEP=0 timeout 600 python tools/gpu_check.py 2>&1 | grep -v "^OK" | tail -4
MAGENT_SOLO_STEP=0 FUZZ_TURN=2 timeout 300 python tools/fuzz_parity.py oracle hip 0 600 2>/dev/null | tail -2
MAGENT_SOLO_STEP=0 MAGENT_CHECKED_STEP=1 timeout 300 python tools/fuzz_parity.py oracle hip 600 900 2>/dev/null | tail -2
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3
unset OMP_NUM_THREADS
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('now', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['breakdown'])"

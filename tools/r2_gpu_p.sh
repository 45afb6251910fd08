#!/bin/bash
mkdir -p gpurun_out/r2p
export TMPDIR=/tmp
for k in 1 2; do
  (cd ab/r1 && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r1 ', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline']['frac'])")
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('now', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done | tee gpurun_out/r2p/ab.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2p/c2 -o t -- python bench.py --map-size 200 --agents 2000 --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2p/c2.log 2>&1
python tools/step_timeline.py $(find gpurun_out/r2p/c2 -name '*kernel_trace.csv' | head -1) 5

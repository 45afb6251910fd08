#!/bin/bash
mkdir -p gpurun_out/r2o
for k in 1 2; do
  (cd ab/r1 && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r1 ', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['breakdown'])")
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('now', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['breakdown'])"
done | tee gpurun_out/r2o/ab.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --gather obs --map-size 200 --agents 2000 --no-cpu-baseline > gpurun_out/r2o/gloo2.json 2> gpurun_out/r2o/gloo2.err
echo "gloo2 rc=$?"; cut -c1-700 gpurun_out/r2o/gloo2.json; grep -v "Gloo\|socket.cpp" gpurun_out/r2o/gloo2.err | tail -5
timeout 200 python bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-400

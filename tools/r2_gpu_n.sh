#!/bin/bash
mkdir -p gpurun_out/r2n
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2n/bench_default.json 2> gpurun_out/r2n/bench_default.err
echo "default rc=$?"; cut -c1-1500 gpurun_out/r2n/bench_default.json; tail -3 gpurun_out/r2n/bench_default.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --gather obs --map-size 200 --agents 2000 --no-cpu-baseline > gpurun_out/r2n/gloo2.json 2> gpurun_out/r2n/gloo2.err
echo "gloo2 rc=$?"; cut -c1-900 gpurun_out/r2n/gloo2.json; tail -5 gpurun_out/r2n/gloo2.err

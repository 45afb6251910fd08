#!/bin/bash
mkdir -p gpurun_out/r2x
timeout 200 python bench.py --workload battle_fill --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2x/fill.log 2>&1; echo "fill rc=$?"; tail -1 gpurun_out/r2x/fill.log | cut -c1-200
timeout 200 python bench.py --workload test_1m --agents 500000 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r2x/t1m.log 2>&1; echo "t1m rc=$?"; tail -1 gpurun_out/r2x/t1m.log | cut -c1-200
timeout 200 python -c "
import sys, time, json; sys.path.insert(0,'.')
import torch, magent_amd, bench
print(json.dumps(bench.small_world_extras(torch, magent_amd, torch.device('cuda',0), steps=100, warmup=10)))
" > gpurun_out/r2x/small.log 2>&1; echo "small rc=$?"; tail -1 gpurun_out/r2x/small.log | cut -c1-300

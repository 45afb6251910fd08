#!/bin/bash
mkdir -p gpurun_out/r2y
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2y/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r2y/bench.log | cut -c1-200
timeout 200 python tools/selfplay_rate.py 400000 3 bf16 > gpurun_out/r2y/selfplay.log 2>&1; echo "selfplay rc=$?"; tail -2 gpurun_out/r2y/selfplay.log
export OMP_NUM_THREADS=1
timeout 600 python -m pytest tests/test_gpu_properties.py tests/test_training.py tests/test_c_client.py -m gpu -q 2>&1 | tail -3

#!/bin/bash
# round-2 GPU session A: sanity of the one-launch step, then the whole GPU suite, then benches (small world on / off, default)
mkdir -p gpurun_out/r2a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export OMP_NUM_THREADS=1
timeout 300 python tools/gpu_check.py battle_small_dense battle_brawl pursuit bodies arrange_live rules_mix duo tri_rect quad > gpurun_out/r2a/sanity.log 2>&1
echo "sanity rc=$?" | tee -a gpurun_out/r2a/sanity.log
tail -15 gpurun_out/r2a/sanity.log
if grep -q "failures: 0" gpurun_out/r2a/sanity.log; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r2a/pytest.log 2>&1
  echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
  tail -40 gpurun_out/r2a/pytest.log
fi
unset OMP_NUM_THREADS
timeout 300 python bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r2a/c2_solo.json 2> gpurun_out/r2a/c2_solo.err
MAGENT_SOLO_STEP=0 timeout 300 python bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r2a/c2_multi.json 2> gpurun_out/r2a/c2_multi.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/c3.json 2> gpurun_out/r2a/c3.err
cat gpurun_out/r2a/c2_solo.json gpurun_out/r2a/c2_multi.json gpurun_out/r2a/c3.json | cut -c1-900

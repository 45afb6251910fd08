# GPU legs of the literal loop and of goal_mode: the new tests, the new scenarios through every parity leg, fuzz
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_set_action_twice.py tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "twice or repeated or goal or literal or render or scenario or oracle or moving or cycle" 2>&1 | grep -v amdgpu.ids | tail -8 > $O/tests.log; tail -5 $O/tests.log
export OMP_NUM_THREADS=1
for k in "FUZZ_TWICE=1 FUZZ_TURN=1" "FUZZ_GOALS_ACT=1 FUZZ_TURN=1" "FUZZ_GOAL=1" "FUZZ_TWICE=1 FUZZ_RULES=2" "FUZZ_GOALS_ACT=1 FUZZ_CYCLE=1 FUZZ_TURN=1"; do
  echo "== $k"; env $k python tools/fuzz_parity.py oracle hip 0 300 2>&1 | tail -1
done > $O/fuzz.log 2>&1; cat $O/fuzz.log

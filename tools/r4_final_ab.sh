# last A/B of the round: the report inside the commit's launch (default) against behind the moves (early_report=0), same box, then a parity subset
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4fab; mkdir -p $O
cd /tmp
for i in 1 2; do
for t in "" "early_report=0"; do
MAGENT_TUNE=$t timeout 300 python $R/bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$t]', d['ms_per_step'], d['repeats_ms_per_step'], d['roofline']['frac'], d['breakdown'])" | tee -a $O/ab.txt
done; done
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "c3 or c5 or scenario or oracle or variants" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e3; mkdir -p $O
cd /tmp
for i in 1 2 3; do
for t in "" "early_report=0"; do
MAGENT_TUNE=$t timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$t]', d['ms_per_step'], d['repeats_ms_per_step'])" | tee -a $O/early.txt
done; done

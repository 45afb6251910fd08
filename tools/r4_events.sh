R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e2; mkdir -p $O
cd /tmp
for i in 1 2; do
for f in "--event-every 4" "--event-every 1" "--no-profile"; do
timeout 300 python $R/bench.py --no-cpu-baseline --no-extras $f 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('[$f]', d['ms_per_step'], d['repeats_ms_per_step'], r.get('frac'), r.get('avg_launch_ms'), r.get('launches'))" | tee -a $O/events.txt
done; done

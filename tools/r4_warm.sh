R=$GRAFT_REPO_ROOT; cd /tmp
for a in "--preheat-ms 40" "--preheat-ms 100" "--preheat-ms 250" "--preheat-ms 40" "--preheat-ms 0"; do
timeout 300 python $R/bench.py --no-cpu-baseline --no-extras $a 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$a', d['ms_per_step'], '%.4e'%d['value'], d['repeats_ms_per_step'], r['frac'], r['avg_launch_ms'], r.get('kernel_alone',{}).get('frac'), r.get('behind_a_drained_stream',{}).get('frac'), d['config']['preheat'])"
done

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ov; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ov in 0 1 2 3 0 3; do MAGENT_OVERLAP=$ov timeout 300 python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-1500 > $O/bench_ov$ov.log; python - <<PY
import json
d=json.loads(open("$O/bench_ov$ov.log").read())
print("overlap=$ov", "%.3e"%d["value"], "%.4f"%d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
done

# development (GPU box): bench line + per-kernel stats of the bench workload;  usage: bash tools/prof_head.sh <outdir>
out=${1:-gpurun_out/prof}; mkdir -p $out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench.json
python -c "
import json; r=json.load(open('$out/bench.json')); print('value %.4g ms/step %.4f frac %.4f render %.4f' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'])); print(r['breakdown']); print(r['config']['steps_finished_by_host_driver'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/rocprof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-60s %6s %9s %9s" % ("kernel", "calls", "avg us", "per step us"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-60s %6s %9.1f %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / 17))
PY
cp $f $out/kernel_stats.csv

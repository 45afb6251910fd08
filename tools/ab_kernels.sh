# development A/B (GPU box): per-kernel averages of the bench workload under engine knobs;  usage: bash tools/ab_kernels.sh <outdir> "VAR=val" "VAR=val" ...
out=$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp$i -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$out/rp$i.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/rp$i -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"
  python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:${TOP:-14}]:
    print("%-60s %6s %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  cp $f $GRAFT_REPO_ROOT/$out/kernel_stats_$i.csv; rm -rf $GRAFT_REPO_ROOT/$out/rp$i
done

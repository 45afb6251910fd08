# development (GPU box): policy kernels -- parity test, rate, per-kernel times at sustained clocks;  usage: bash tools/policy_prof.sh [reps]
reps=${1:-1000}
python -m pytest tests/test_policy.py -x -q 2>&1 | tail -2
python tools/policy_rate.py 131072 $reps cells 2>&1 | grep "HIP policy"
python tools/policy_rate.py 131072 $reps 2>&1 | grep "HIP policy"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/polprof -o p -- python $GRAFT_REPO_ROOT/tools/policy_rate.py 131072 $reps cells > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/polprof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:3]:
    print("%-50s calls %s avg %.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/polprof

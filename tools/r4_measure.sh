# round 4 measurement call: quick parity (the scenarios + the full-size digests), a bench line, kernel stats.  usage: bash tools/r4_measure.sh <tag> [pytest -k expr]
R=$GRAFT_REPO_ROOT; TAG=${1:-r4m}; K=${2:-"c3 or c5 or c2 or c4"}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "$K or scenario or oracle" 2>&1 | grep -v amdgpu.ids | tail -6 > $O/tests.log; tail -4 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --no-cpu-baseline --no-extras > $O/bench.log 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.log").read().strip().splitlines()[-1])
print("%.4e"%d["value"], d["ms_per_step"], d["repeats_ms_per_step"], d["roofline"]["frac"], d["breakdown"], d["config"]["steps_finished_by_host_driver"])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/bench_kernel_stats.csv")))
tot=0
for r in rows[:26]:
    n=r["Name"].split("(")[0].replace("magent_amd::","").replace("void ","")[:40]
    print("%-42s %5s %8.1f"%(n, r["Calls"], float(r["AverageNs"])/1e3))
PY

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --workload test_1m --agents 500000 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/bench_kernel_stats.csv")))
for r in rows[:30]:
    n=r["Name"].split("(")[0].replace("magent_amd::","").replace("void ","")[:40]
    print("%-42s %5s %8.1f %8.2f"%(n, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY

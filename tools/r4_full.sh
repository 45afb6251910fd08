# round 4: the GPU suite subset that touches the step + the full bench line (extras included) + kernel stats
R=$GRAFT_REPO_ROOT; TAG=${1:-r4f}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_properties.py tests/test_training.py tests/test_bench_multi.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|^batch" | tail -8 > $O/tests.log; tail -5 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench.log 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.log").read().strip().splitlines()[-1])
print("%.4e"%d["value"], d["ms_per_step"], d["repeats_ms_per_step"], d["roofline"]["frac"], d["breakdown"], d["config"]["steps_finished_by_host_driver"])
e=d["extra"]
print({k:(v.get("ms_per_step") if isinstance(v,dict) else None) for k,v in e.items()})
print(json.dumps(e.get("c5_train_round_1m"))[:1500]); print(e.get("error")); print(d["cpu_baseline"])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/bench_kernel_stats.csv")))
for r in rows[:20]:
    n=r["Name"].split("(")[0].replace("magent_amd::","").replace("void ","")[:40]
    print("%-42s %5s %8.1f"%(n, r["Calls"], float(r["AverageNs"])/1e3))
PY

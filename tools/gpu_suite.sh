# Development helper (GPU box, via gpurun): the whole GPU suite, two bench lines and the launch timeline of one cycle.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/suite; mkdir -p $O
cd $R; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for k in 1 2; do timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-1700 > $O/bench.log; python - <<PY
import json
d=json.loads(open("$O/bench.log").read())
print("%.3e"%d["value"], "%.4f"%d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["breakdown"])
PY
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python $R/tools/step_timeline.py $O/stats/bench_kernel_trace.csv | tail -40

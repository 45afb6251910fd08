R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4multi; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_replicas.py tests/test_bench_multi.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 > $O/tests.log; tail -3 $O/tests.log
cd /tmp; timeout 600 python $R/bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --repeats 1 2>$O/two.err | tail -1 > $O/two_ranks_gloo.log; python - <<PY
import json
d=json.loads(open("$O/two_ranks_gloo.log").read()); print(d["n_gpus"], "%.3e"%d["value"], d["ms_per_step"], d["config"]["backend"], json.dumps(d["extra"]["c4_gather_rccl"])[:500])
PY
tail -3 $O/two.err

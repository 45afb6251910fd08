# Round-end measurements (GPU box): everything profiles/r04_* and DESIGN.md section 6 quote.  ~6 GPU-minutes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>$O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python $R/bench.py --no-cpu-baseline --no-extras --obs bf16 2>&1 | tail -1 | cut -c1-1200 > $O/bf16.log
python $R/bench.py --no-cpu-baseline --no-extras --workload gather --map-size 500 --agents 100000 2>&1 | tail -1 | cut -c1-1600 > $O/gather.log
python $R/bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-extras 2>&1 | tail -1 > $O/c2.log
for a in "1 1" "8 8" "32 8" "128 8"; do python $R/tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done > $O/batch.log
python $R/bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --repeats 1 2>/dev/null | tail -1 > $O/two_ranks_gloo.log
cut -c1-500 $O/gather.log; cut -c1-400 $O/c2.log; cat $O/batch.log; cut -c1-500 $O/bf16.log; python - <<PY
import json
d=json.loads(open("$O/two_ranks_gloo.log").read()); print(d["value"], d["ms_per_step"], json.dumps(d["extra"]["c4_gather_rccl"])[:900])
PY
bash $R/tools/step_pmc.sh r04final_pmc > $O/pmc.log 2>&1; tail -1 $O/pmc.log

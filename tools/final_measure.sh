# Round-end measurements (GPU box): everything profiles/r03_* and DESIGN.md section 6 quote.  ~4 GPU-minutes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $O/write.log 2>&1
python $R/bench.py --no-cpu-baseline --no-extras --obs bf16 2>&1 | tail -1 | cut -c1-900 > $O/bf16.log
python $R/bench.py --no-cpu-baseline --no-extras --workload gather --map-size 500 --agents 100000 2>&1 | tail -1 | cut -c1-330 > $O/gather.log
python $R/bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-extras 2>&1 | tail -1 > $O/c2.log
for a in "1 1" "8 8" "32 8" "128 8"; do python $R/tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done > $O/batch.log
python $R/tools/solo_marks.py 200 2000 2>&1 | grep -v amdgpu > $O/solo_marks.log
python $R/bench.py --gpus 2 --backend gloo --gather obs --check-gather --steps 5 --warmup 2 --map-size 400 --agents 50000 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/two_ranks_gloo.log
cat $O/gather.log; cut -c1-400 $O/c2.log; cat $O/batch.log; cut -c1-600 $O/bf16.log; cut -c1-700 $O/two_ranks_gloo.log

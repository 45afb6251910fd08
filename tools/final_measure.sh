# Round-end measurements (GPU box): everything profiles/r02_* and DESIGN.md section 6 quote.  ~4 GPU-minutes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-extras > $O/write.log 2>&1
python $R/bench.py --no-cpu-baseline --no-extras --workload gather --map-size 500 --agents 100000 2>&1 | tail -1 | cut -c1-330 > $O/gather.log
python $R/bench.py --map-size 200 --agents 2000 --steps 300 --warmup 20 --no-extras 2>&1 | tail -1 > $O/c2.log
for a in "1 1" "8 8" "32 8" "128 8"; do python $R/tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done > $O/batch.log
python $R/tools/host_abi_rate.py > $O/host_abi.log 2>&1
python $R/tools/selfplay_rate.py 400000 6 bf16 > $O/selfplay_bf16.log 2>&1; python $R/tools/selfplay_rate.py 400000 6 bf16 cells > $O/selfplay_cells.log 2>&1
python $R/tools/policy_rate.py 131072 20 torch 2>&1 | grep "HIP policy\|torch bf16" > $O/policy_rate.log; python $R/tools/policy_rate.py 131072 20 cells 2>&1 | grep "HIP policy" >> $O/policy_rate.log
cat $O/gather.log; cut -c1-400 $O/c2.log; cat $O/batch.log; tail -1 $O/host_abi.log; tail -1 $O/selfplay_bf16.log; tail -1 $O/selfplay_cells.log; cat $O/policy_rate.log

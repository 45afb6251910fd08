set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-profile > $O/write.log 2>&1
python $R/bench.py --no-cpu-baseline --workload test_1m --agents 500000 2>&1 | tail -1 | cut -c1-200 > $O/test_1m.log
python $R/bench.py --no-cpu-baseline --workload gather --map-size 500 --agents 100000 2>&1 | tail -1 | cut -c1-200 > $O/gather.log
python $R/bench.py --no-cpu-baseline --map-size 200 --agents 2000 --steps 300 --warmup 20 2>&1 | tail -1 | cut -c1-200 > $O/c2.log
python $R/tools/host_abi_rate.py > $O/host_abi.log 2>&1
python $R/tools/many_envs.py > $O/many_envs.log 2>&1
cat $O/test_1m.log $O/gather.log $O/c2.log; tail -3 $O/host_abi.log; tail -3 $O/many_envs.log

# round 4, first GPU call: the new parity tests, the step kernels' counters, the new bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_training.py tests/test_policy.py tests/test_replicas.py tests/test_bench_multi.py -x -q -m gpu -k "c5 or bf16_policy or rccl or two_ranks or bench" -s 2>&1 | grep -v "amdgpu.ids\|^batch\|^batches" | tail -25 > $O/tests.log
tail -12 $O/tests.log
bash tools/step_pmc.sh r4a_pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-1500

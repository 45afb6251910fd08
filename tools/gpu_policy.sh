# Development helper (GPU box, via gpurun): policy kernel tests, rate, and per-kernel times under rocprofv3.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/policy; mkdir -p $O
cd $R; timeout 600 python -m pytest tests/test_policy.py -x -q -m gpu 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o pr -- python $R/tools/policy_rate.py 131072 20 cells > $O/pr.log 2>&1
grep -v "rocprof\|amdgpu.ids" $O/pr.log | tail -3
grep "k_dqn" $O/stats/pr_kernel_stats.csv | cut -c1-140

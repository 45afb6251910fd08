R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2pol; mkdir -p $O
cd $R; timeout 600 python -m pytest tests/test_policy.py -x -q -m gpu 2>&1 | tail -15
cd /tmp; timeout 300 python $R/tools/selfplay_rate.py 400000 4 bf16 2>&1 | tail -2

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2pol2; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_policy.py tests/test_training.py -x -q -m gpu 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/selfplay_rate.py 400000 6 bf16 2>&1 | tail -1 | tee $O/selfplay_hip.log
MAGENT_HIP_POLICY=0 timeout 300 python $R/tools/selfplay_rate.py 400000 4 bf16 2>&1 | tail -1 | tee $O/selfplay_torch.log
timeout 300 python $R/tools/policy_rate.py 131072 20 torch 2>&1 | grep "HIP policy\|torch bf16" | tee $O/policy_rate.log

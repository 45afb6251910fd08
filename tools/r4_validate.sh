# round 4, final tree: the whole GPU suite, the fuzz campaign, the watchdog of the N > 1 extra
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids\|^batch" | tail -6 > $O/tests.log; tail -3 $O/tests.log
bash tools/fuzz_campaign.sh > $O/fuzz.log 2>&1; cat $O/fuzz.log
cd /tmp; timeout 600 python $R/bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --repeats 1 --extra-timeout 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('watchdog:', d['n_gpus'], d['value']>0, d['extra']['c4_gather_rccl'])"

# the five-channel sweep: parity at the reference's own 1M recipe, its bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_properties.py -x -q -m gpu -k "1m or pursuit or scenario or oracle or invariants" 2>&1 | grep -v amdgpu.ids | tail -5 > $O/tests.log; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for t in "" "render=0"; do
MAGENT_TUNE=$t timeout 600 python $R/bench.py --workload test_1m --agents 500000 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tune=[$t]', '%.4e'%d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['breakdown'])" | tee -a $O/bench_1m.txt
done

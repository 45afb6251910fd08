R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2batch; mkdir -p $O
cd $R; timeout 600 python -m pytest tests/test_gpu_properties.py -x -q -m gpu -k "batch or fused" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cycle" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for a in "1 1" "8 8" "32 8" "128 8" "128 1" "256 8"; do timeout 300 python $R/tools/many_envs_batch.py $a 2>&1 | grep -v amdgpu.ids; done > $O/batch.log
cat $O/batch.log

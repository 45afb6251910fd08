cd $GRAFT_REPO_ROOT
run() { echo "-- $*"; timeout 120 env "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c150-420; }
for k in 1 2; do
run python tools/many_envs_pipe.py battle 200 2000 32 200
run MAGENT_TUNE=pipe_sweep=0 python tools/many_envs_pipe.py battle 200 2000 32 200
done
run python tools/many_envs_pipe.py battle 200 2000 32 600
run python tools/many_envs_pipe.py battle 200 2000 8 200
run MAGENT_TUNE=pipe_sweep=0 python tools/many_envs_pipe.py battle 200 2000 8 200
run python tools/many_envs_pipe.py battle 200 2000 128 100
run python tools/many_envs_pipe.py battle 600 20000 8 60
run MAGENT_TUNE=pipe_sweep=0 python tools/many_envs_pipe.py battle 600 20000 8 60
run python tools/many_envs_pipe.py gather 500 100000 8 20
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "pipeline or batch" 2>&1 | tail -2

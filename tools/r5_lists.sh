R=$GRAFT_REPO_ROOT; cd $R
K="test_1m or pursuit or bodies or arrange or forest or double or tri or turn or sector or duo or chase or quad" bash tools/measure.sh r5f parity
bash tools/measure.sh r5f_1m line,stats -- --workload test_1m --agents 500000 --repeats 3 --steps 10 --warmup 3
python tools/step_timeline.py gpurun_out/r5f_1m/stats/bench_kernel_trace.csv 3 k_step_report
(run() { echo "-- $*"; env "$@" 2>&1 | tail -1; }
 run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 3000 3800
 run MAGENT_TUNE=solo_step=0,move_batches=0 python tools/fuzz_parity.py oracle hip 3800 4200
 run FUZZ_TURN=1 MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 300 700
 run python tools/fuzz_parity.py oracle hip 4200 4800)

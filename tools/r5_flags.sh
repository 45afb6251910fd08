R=$GRAFT_REPO_ROOT; cd $R
K="test_1m or pursuit or bodies or arrange or forest or double or tri or turn or sector or duo or chase or quad or food or rules or gather" bash tools/measure.sh r5g parity
bash tools/measure.sh r5g_1m line,stats -- --workload test_1m --agents 500000 --repeats 3 --steps 10 --warmup 3
(run() { echo "-- $*"; env "$@" 2>&1 | tail -1; }
 run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 5000 6000
 run MAGENT_TUNE=solo_step=0,attack_pairs=0 python tools/fuzz_parity.py oracle hip 6000 6400
 run FUZZ_TURN=1 MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 700 1100
 run FUZZ_RULES=2 MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 0 400
 run FUZZ_CYCLE=1 python tools/fuzz_parity.py oracle hip 300 700
 run python tools/fuzz_parity.py oracle hip 6400 7000)

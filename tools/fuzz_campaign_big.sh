export OMP_NUM_THREADS=1
run() { echo "== $*"; env "$@" 2>&1 | tail -1; }
run python tools/fuzz_parity.py oracle hip 10000 16000
run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 16000 20000
run FUZZ_TURN=2 python tools/fuzz_parity.py oracle hip 20000 22000
run FUZZ_RULES=2 python tools/fuzz_parity.py oracle hip 22000 23500
run FUZZ_CYCLE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 23500 25500
run FUZZ_BATCH=4 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 25500 26000

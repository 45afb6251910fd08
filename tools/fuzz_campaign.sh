# round-end differential fuzzing on the GPU box (HIP engine vs CPU oracle); ~4 GPU-minutes
export OMP_NUM_THREADS=1
run() { echo "== $*"; env "$@" 2>&1 | tail -1; }
run python tools/fuzz_parity.py oracle hip 0 1200
run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 1200 2000
run MAGENT_TUNE=solo_step=0,scan_solo_max=64 python tools/fuzz_parity.py oracle hip 2000 2600
run FUZZ_TURN=2 python tools/fuzz_parity.py oracle hip 0 500
run FUZZ_GOAL=1 python tools/fuzz_parity.py oracle hip 0 300
run FUZZ_TWICE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 0 300
run FUZZ_GOALS_ACT=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 0 400
run FUZZ_RULES=2 python tools/fuzz_parity.py oracle hip 0 400
run FUZZ_CYCLE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 0 500
run FUZZ_BATCH=3 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 0 200
run MAGENT_TUNE=solo_step=0,attack_pairs=0 python tools/fuzz_parity.py oracle hip 2600 3200
run MAGENT_TUNE=render=4,render_sweep=3 python tools/fuzz_parity.py oracle hip 3500 3900
run MAGENT_TUNE=render=1 python tools/fuzz_parity.py oracle hip 3900 4200

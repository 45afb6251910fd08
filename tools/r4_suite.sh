# round 4: the whole GPU suite + a fuzz slice on every step driver + counters of the step kernels (TAG = $1)
R=$GRAFT_REPO_ROOT; TAG=${1:-r4s}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|^batch" | tail -8 > $O/tests.log; tail -4 $O/tests.log
export OMP_NUM_THREADS=1
run() { echo "== $*"; env "$@" 2>&1 | tail -1; }
( run python tools/fuzz_parity.py oracle hip 0 600
  run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 600 1600
  run MAGENT_TUNE=solo_step=0,scan_solo_max=64 python tools/fuzz_parity.py oracle hip 1600 2200
  run MAGENT_TUNE=solo_step=0,attack_pairs=0 python tools/fuzz_parity.py oracle hip 2200 2700
  run MAGENT_TUNE=solo_step=0,overlap=3 python tools/fuzz_parity.py oracle hip 2700 3000
  run FUZZ_CYCLE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 0 300 ) > $O/fuzz.log 2>&1
cat $O/fuzz.log
bash tools/step_pmc.sh ${TAG}_pmc > $O/pmc.log 2>&1; tail -2 $O/pmc.log

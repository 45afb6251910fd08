# One GPU-box measurement script (via gpurun), replacing the per-call wrappers of the earlier rounds.
#
#   usage: bash tools/measure.sh <tag> <job>[,<job>...] [-- <bench.py arguments>]
#
# Every job writes under gpurun_out/<tag>/ and prints a few summary lines; the bench arguments after `--` select the workload for
# the jobs that run bench.py (default: the headline workload).  Jobs, in the order given:
#   suite     the whole `pytest -m gpu` suite
#   parity    tests/test_gpu_parity.py + test_gpu_fullsize.py (K=<pytest -k expression> narrows it)
#   callers   tests/test_callers.py -m gpu: the reference's own scripts on the HIP engine (figures -> gpurun_out/callers.json)
#   fuzz      a slice of every fuzz generator on every step driver
#   line      one bench line without extras / CPU baseline         -> line.json
#   full      the driver's default command (extras, CPU baseline)  -> full.json
#   stats     rocprofv3 --kernel-trace --stats of a short run      -> stats/bench_kernel_stats.csv + a table
#   bytes     FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ / TCC_EA0_WRREQ passes (separate --pmc runs) -> bytes.json + a table
#   pmc       the full counter set of tools/step_pmc.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-m}; JOBS=${2:-line}; shift; shift
[ "$1" = "--" ] && shift
ARGS="$*"
O=$R/gpurun_out/$TAG; mkdir -p $O
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-1}
short="--steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-cold"

table() {   # kernel table of a rocprofv3 --stats csv
python - "$1" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    n = r["Name"].split("(")[0].replace("magent_amd::", "").replace("void ", "")[:44]
    print("%-46s %5s %9.1f us %9.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}

for job in ${JOBS//,/ }; do
  echo "== $job"
  case $job in
    suite)   (cd $R && timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|^batch" | tail -8) | tee $O/suite.log ;;
    parity)  (cd $R && timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu ${K:+-k "$K"} 2>&1 | grep -v amdgpu.ids | tail -6) | tee $O/parity.log ;;
    callers) (cd $R && timeout 1800 python -m pytest tests/test_callers.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -16) | tee $O/callers.log
             cp $R/gpurun_out/callers.json $O/ 2>/dev/null ;;
    fuzz)    (cd $R
              run() { echo "-- $*"; env "$@" 2>&1 | tail -1; }
              run python tools/fuzz_parity.py oracle hip 0 600
              run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip 600 1600
              run MAGENT_TUNE=solo_step=0,scan_solo_max=64 python tools/fuzz_parity.py oracle hip 1600 2200
              run MAGENT_TUNE=solo_step=0,attack_pairs=0 python tools/fuzz_parity.py oracle hip 2200 2700
              run FUZZ_CYCLE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip 0 300) 2>&1 | tee $O/fuzz.log ;;
    line)    (cd /tmp && export TMPDIR=/tmp && timeout 900 python $R/bench.py --no-cpu-baseline --no-extras --no-cold $ARGS > $O/line.json 2> $O/line.err)
             python - $O/line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"] or {}
print("%.4e agent-steps/s  %.4f ms/step  repeats %s" % (d["value"], d["ms_per_step"], d["repeats_ms_per_step"]))
print("render:", r.get("kernel"), r.get("avg_launch_ms"), "ms/launch  frac", r.get("frac"), " alone", (r.get("kernel_alone") or {}).get("frac"), " traffic", r.get("traffic"))
print("breakdown:", d["breakdown"])
print("host-finished steps", d["config"]["steps_finished_by_host_driver"], " rounds", d["config"]["attack_round_hist"], " agents", d["config"]["agents_at_start"], "->", d["config"]["agents_at_end"])
PY
             ;;
    full)    (cd /tmp && export TMPDIR=/tmp && timeout 1500 python $R/bench.py $ARGS > $O/full.json 2> $O/full.err); tail -c 600 $O/full.json ;;
    stats)   (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py $short $ARGS > $O/stats.log 2>&1)
             table $O/stats/bench_kernel_stats.csv ;;
    bytes)   for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
               (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o bench -- python $R/bench.py $short --no-profile --preheat-ms 0 $ARGS > $O/$c.log 2>&1)
             done
             python $R/tools/pmc_bytes.py $O $O/bytes.json ;;
    pmc)     bash $R/tools/step_pmc.sh ${TAG}/pmc $ARGS | tail -3 ;;
    *)       echo "unknown job $job" ;;
  esac
done

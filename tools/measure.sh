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
#   ab        `line` once per MAGENT_TUNE setting of AB="set1;set2;..." ("-" = the defaults)   -> ab_<n>.json, one row each
#   timeline  the launches of one step in order, from the trace `stats` wrote (tools/step_timeline.py)
#   default   the driver's default command timed end to end (wall seconds, return code)                       -> full.json
#   n8        the default N = 8 command as a dry run over gloo on this one GPU + tests/test_bench_multi.py      -> n8_gloo.json
#   probes    tools/probe/*.hip + tools/render_locality.py (how stores and window reads behave on this part)
# Environment: K (pytest -k of `parity`), AB (settings of `ab`), FUZZ_FROM / FUZZ_N (first seed and seeds per generator of `fuzz`; default 0 / as listed)
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
              run() { echo "-- $*"; timeout 300 env "$@" 2>&1 | tail -1; }     # (a game whose rules make the reference's search cubic can take minutes in the ORACLE: bounded)
              F=${FUZZ_FROM:-0}; s() { echo $((F + $1)); }
              run python tools/fuzz_parity.py oracle hip $(s 0) $(s 600)
              run MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip $(s 600) $(s 1600)
              run MAGENT_TUNE=solo_step=0,scan_solo_max=64 python tools/fuzz_parity.py oracle hip $(s 1600) $(s 2200)
              run MAGENT_TUNE=solo_step=0,attack_pairs=0 python tools/fuzz_parity.py oracle hip $(s 2200) $(s 2700)
              run MAGENT_TUNE=solo_step=0,move_batches=0 python tools/fuzz_parity.py oracle hip $(s 2700) $(s 3000)
              run FUZZ_TURN=1 MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip $(s 0) $(s 300)
              run FUZZ_RULES=2 MAGENT_TUNE=solo_step=0 python tools/fuzz_parity.py oracle hip $(s 0) $(s 200)
              run FUZZ_CYCLE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip $(s 0) $(s 300)
              # (the one-launch step at the sizes only a batch gives it by default since round 5: 1536 < agents <= 16384)
              run MAGENT_TUNE=solo_max=16384 python tools/fuzz_parity.py oracle hip $(s 3000) $(s 3400)
              run MAGENT_TUNE=solo_max=16384 FUZZ_CYCLE=1 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip $(s 300) $(s 600)
              # (round 6: three environments per plain game through env_cycle_many's batched pipeline, pipe.hip -- every world in the batch, then with one / no optimistic pair)
              run FUZZ_PLAIN=1 FUZZ_BATCH=3 MAGENT_TUNE=batch_pipe_min=1 python tools/fuzz_parity.py oracle hip $(s 0) $(s 400)
              run FUZZ_PLAIN=1 FUZZ_BATCH=3 MAGENT_TUNE=batch_pipe_min=1,attack_pairs=1 python tools/fuzz_parity.py oracle hip $(s 400) $(s 700)
              run FUZZ_PLAIN=1 FUZZ_BATCH=4 MAGENT_TUNE=batch_pipe_min=1,attack_pairs=0 python tools/fuzz_parity.py oracle hip $(s 700) $(s 900)
              run FUZZ_BATCH=3 FUZZ_TURN=1 python tools/fuzz_parity.py oracle hip $(s 0) $(s 200)) 2>&1 | tee $O/fuzz.log ;;
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
    ab)      i=0; IFS=';' read -ra SETS <<< "${AB:--}"
             for t in "${SETS[@]}"; do
               [ "$t" = "-" ] && t=""
               (cd /tmp && export TMPDIR=/tmp && MAGENT_TUNE="$t" timeout 900 python $R/bench.py --no-cpu-baseline --no-extras --no-cold $ARGS > $O/ab_$i.json 2> $O/ab_$i.err)
               python - $O/ab_$i.json "$t" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"] or {}; b = d["breakdown"]
    print("%-44s %.4f ms/step | %s %.4f ms/launch frac %.3f | render %.4f attack %.4f move %.4f clear %.4f order %s" % (
        sys.argv[2] or "(defaults)", d["ms_per_step"], r.get("kernel"), r.get("avg_launch_ms", 0), r.get("frac", 0), b.get("render_ms_per_step", 0),
        b.get("attack_ms_per_step", 0), b.get("move_ms_per_step", 0), b.get("clear_dead_ms_per_step", 0), b.get("map_order_ms_per_step")))
except Exception as e:
    print("%-44s FAILED: %r" % (sys.argv[2], e))
PY
               i=$((i + 1))
             done ;;
    timeline) python $R/tools/step_timeline.py $O/stats/bench_kernel_trace.csv 3 ;;
    default) python - <<PY
import subprocess, time, json
t0 = time.time()
p = subprocess.run(["python", "$R/bench.py"] + "$ARGS".split(), capture_output=True, text=True, timeout=1500, cwd="/tmp")
dt = time.time() - t0
open("$O/full.json", "w").write(p.stdout); open("$O/full.err", "w").write(p.stderr)
rec = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
print("default bench.py: wall %.1f s rc %d" % (dt, p.returncode))
print("%.4e  %.4f ms  no_preheat %s  frac %s" % (rec["value"], rec["ms_per_step"], rec.get("ms_per_step_no_preheat"), rec["roofline"]["frac"]))
print("cpu:", rec["cpu_baseline"])
e = rec.get("extra", {})
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in e.items()})
print(e.get("error"))
PY
             ;;
    n8)      (cd $R && timeout 1500 python -m pytest tests/test_bench_multi.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3) | tee $O/n8_tests.log
             python - <<PY
import subprocess, time, json
t0 = time.time()
p = subprocess.run(["python", "$R/bench.py", "--gpus", "8", "--backend", "gloo", "--extra-timeout", "900"], capture_output=True, text=True, timeout=1500, cwd="/tmp")
dt = time.time() - t0
lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
rec = json.loads(lines[-1])
rec["dry_run"] = {"command": "python bench.py --gpus 8 --backend gloo", "wall_seconds": round(dt, 1), "returncode": p.returncode, "json_lines": len(lines),
                  "note": "8 ranks share ONE MI355X over gloo: the driver's default N = 8 command end to end (headline on config 3 per rank, then extra.c4_gather_rccl with every shard verified); a dry run of the code path and its footprint, not a measurement of xGMI"}
json.dump(rec, open("$O/n8_gloo.json", "w"), indent=1)
print("wall %.1f s, rc %d, %d line(s); value %.3e; extra verified %s" % (dt, p.returncode, len(lines), rec["value"], rec["extra"]["c4_gather_rccl"].get("verified")))
PY
             ;;
    probes)  (cd $R && hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_chunks tools/probe/scatter_chunks.hip && /tmp/scatter_chunks > $O/scatter_chunks.txt 2>&1
              python tools/render_locality.py 2>&1 | grep -v amdgpu.ids | tee $O/render_locality.txt) ;;
    *)       echo "unknown job $job" ;;
  esac
done

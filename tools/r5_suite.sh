R=$GRAFT_REPO_ROOT; cd $R
bash tools/measure.sh r5e suite,fuzz
MAGENT_TUNE=overlap=3 bash tools/measure.sh r5e_overlap line
python - <<PY
import subprocess, time, json
t0 = time.time()
p = subprocess.run(["python", "$R/bench.py"], capture_output=True, text=True, timeout=1500, cwd="/tmp")
dt = time.time() - t0
open("$R/gpurun_out/r5e/full.json", "w").write(p.stdout)
open("$R/gpurun_out/r5e/full.err", "w").write(p.stderr)
rec = json.loads([l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1])
print("default bench.py: wall %.1f s rc %d" % (dt, p.returncode))
print("%.4e  %.4f ms  no_preheat %s  frac %s" % (rec["value"], rec["ms_per_step"], rec["ms_per_step_no_preheat"], rec["roofline"]["frac"]))
print("cpu:", rec["cpu_baseline"])
e = rec["extra"]
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in e.items()})
print(json.dumps(e.get("c5_cycle_3536"))[:1500])
print(e.get("error"))
PY

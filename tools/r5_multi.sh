# round 5: the N > 1 path on the 1-GPU box -- the GPU tests of bench.py's own multi-rank line, then the DEFAULT command at N = 8 as a
# dry run over gloo (8 ranks share the device, config 3 per rank + the config-4 gather extra): wall time and the line, for profiles/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_bench_multi.py -x -q -m gpu > $O/tests.log 2>&1; grep -v amdgpu.ids $O/tests.log | tail -3
cd /tmp; export TMPDIR=/tmp
python - <<PY
import subprocess, time, json
t0 = time.time()
p = subprocess.run(["python", "$R/bench.py", "--gpus", "8", "--backend", "gloo", "--extra-timeout", "900"], capture_output=True, text=True, timeout=1500)
dt = time.time() - t0
lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
rec = json.loads(lines[-1])
rec["dry_run"] = {"command": "python bench.py --gpus 8 --backend gloo", "wall_seconds": round(dt, 1), "returncode": p.returncode, "json_lines": len(lines),
                  "note": "8 ranks share ONE MI355X over gloo: the driver's default N = 8 command end to end (headline on config 3 per rank, then extra.c4_gather_rccl with every shard verified); a dry run of the code path and its footprint, not a measurement of xGMI"}
json.dump(rec, open("$O/n8_gloo.json", "w"), indent=1)
print("wall %.1f s, rc %d, %d line(s); value %.3e; extra verified %s" % (dt, p.returncode, len(lines), rec["value"], rec["extra"]["c4_gather_rccl"].get("verified")))
PY

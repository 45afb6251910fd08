"""Development helper (GPU box) and the body of tests/test_gpu_fullsize.py::test_long_episodes_of_the_plain_pipeline: play full-size
scenarios on the HIP engine and compare every array of every step with the digests of the COMPILED REFERENCE
(tests/golden/digests_fullsize.json).  The engine reads MAGENT_TUNE once per process, so every driver variant is a process of its own.

    python tools/gpu_golden_check.py [--device-io] [--episodes] name [name ...]

Prints one JSON line per scenario: {"name", "steps", "ok", "first_difference", "pipeline_stats", "engine_stats", "seconds"}."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["OMP_NUM_THREADS"] = "1"
import torch  # noqa: F401,E402  (first, so the engine shares torch's HIP runtime)
import helpers as H  # noqa: E402

args = sys.argv[1:]
device_io = "--device-io" in args
names = [a for a in args if not a.startswith("--")]
episodes = "--episodes" in args          # whole episodes (tests/helpers.episode_scenarios, tests/golden/digests_episode.json): by hand, not in the suite
with open(os.path.join(H.GOLDEN_DIR, "digests_episode.json" if episodes else "digests_fullsize.json")) as f:
    GOLD = json.load(f)
FULL = H.episode_scenarios() if episodes else H.fullsize_scenarios()
bad = 0
for name in names:
    t = time.time()
    seen = []
    got = H.run_hashed(FULL[name], H.HIP_LIB, device_io=device_io, env_out=seen)
    diff = None
    try:
        H.assert_same_hashed(GOLD[name], got, name)
    except AssertionError as e:
        diff = str(e)[:300]
        bad += 1
    print(json.dumps({"name": name, "steps": len(got), "ok": diff is None, "first_difference": diff, "pipeline_stats": seen[0].pipeline_stats(),
                      "engine_stats": seen[0].engine_stats(), "seconds": round(time.time() - t, 1)}), flush=True)
sys.exit(1 if bad else 0)

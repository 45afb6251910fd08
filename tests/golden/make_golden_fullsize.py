"""Per-step hashes of BASELINE.json's configurations at their stated sizes, from the COMPILED REFERENCE.

    OMP_NUM_THREADS=1 python tests/golden/make_golden_fullsize.py [name ...]

Writes tests/golden/digests_fullsize.json: {scenario: [{array name: xxh3-128 hex} per step]} (tests/helpers.run_hashed).
The scenarios are tests/helpers.fullsize_scenarios(); the GPU tests compare the HIP engine with these hashes and with
the CPU checkers run beside it.  Takes a few minutes (single-threaded reference, 3.9 GB of observations per C3 step).
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ["OMP_NUM_THREADS"] = "1"

import helpers as H  # noqa: E402


def main():
    assert H.have_ref(), "build oracle/_ref first: make -C oracle ref"
    episodes = "--episodes" in sys.argv          # (whole episodes: tests/helpers.episode_scenarios -> digests_episode.json)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = os.path.join(HERE, "digests_episode.json" if episodes else "digests_fullsize.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    S = H.episode_scenarios() if episodes else H.fullsize_scenarios()
    for name in args or sorted(S):
        t = time.time()
        out[name] = H.run_hashed(S[name], H.REF_LIB)
        print(name, len(out[name]), "steps", "%.1fs" % (time.time() - t), flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()

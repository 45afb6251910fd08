"""Differential fuzzing: random in-scope games, engine A vs engine B, bit for bit.

  python tools/fuzz_parity.py ref oracle 0 300      # here: pins the CPU restatement against the compiled reference
  python tools/fuzz_parity.py oracle hip 0 300      # GPU box: the HIP engine against the oracle
  FUZZ_BATCH=3 python tools/fuzz_parity.py oracle hip 0 300   # three environments per game in one EnvBatch (one launch pair for all)
  FUZZ_CYCLE=1 python tools/fuzz_parity.py oracle hip 0 300   # the HIP leg through env_cycle_many (two launches per cycle), the
                                                              # other leg through the reference call sequence
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["OMP_NUM_THREADS"] = "1"
if "hip" in sys.argv[1:3]:
    import torch  # noqa: F401
import helpers as H  # noqa: E402

LIBS = {"ref": H.REF_LIB, "oracle": H.ensure_oracle(), "hip": H.HIP_LIB}
if "emu" in sys.argv[1:3]:     # the HIP sources run lane by lane on the CPU (tests/hipemu): takes the place of "hip" in a GPU-less container
    LIBS["hip"] = H.HIP_LIB = H.ensure_emu()
    sys.argv[1:3] = ["hip" if v == "emu" else v for v in sys.argv[1:3]]
a, b = sys.argv[1], sys.argv[2]
if "," in sys.argv[3]:       # an explicit list of seeds:  fuzz_parity.py hip oracle 21841,21943
    seeds = [int(x) for x in sys.argv[3].split(",") if x]
else:
    seeds = list(range(int(sys.argv[3]), int(sys.argv[4])))
bad, t0, piped, swept = [], time.time(), 0, 0
for seed in seeds:
    sc = H.fuzz_scenario(seed)
    print("seed", seed, flush=True, file=sys.stderr)
    try:
        nb = int(os.environ.get("FUZZ_BATCH", "0"))
        if nb > 1:      # nb environments of this configuration (different engine and action seeds) in ONE EnvBatch on the HIP side
            import copy
            scs = []
            for k in range(nb):
                c = copy.deepcopy(sc); c.seed, c.action_seed, c.clear_every = sc.seed + 1000 * k, sc.action_seed + k, 1
                scs.append(c)
            seen = []
            hip_side = H.run_cycle_batch(scs, H.HIP_LIB, envs_out=seen)
            piped += any(e.pipeline_stats()[6] > 0 for e in seen)
            swept += any(e.pipeline_stats()[7] > 0 for e in seen)
            other = b if a == "hip" else a
            for k, c in enumerate(scs):
                H.assert_same(H.run_cycle(c, LIBS[other], fused=False), hip_side[k], "%s (batch of %d, env %d)" % (sc.name, nb, k))
        elif os.environ.get("FUZZ_CYCLE", "0") == "1":
            sc.clear_every = 1
            H.assert_same(H.run_cycle(sc, LIBS[a], fused=(a == "hip")), H.run_cycle(sc, LIBS[b], fused=(b == "hip")), sc.name + " (cycle)")
        else:
            H.assert_same(H.run(sc, LIBS[a]), H.run(sc, LIBS[b]), sc.name)
    except AssertionError as e:
        bad.append(seed)
        print("FAIL seed %d: %s" % (seed, str(e)[:300]), flush=True)
if int(os.environ.get("FUZZ_BATCH", "0")) > 1:
    print("games with environments in the batched pipeline (pipe.hip): %d (observations by its sweeping render: %d)" % (piped, swept))
print("%d seeds, %d failures %s, %.1fs" % (len(seeds), len(bad), bad, time.time() - t0))
sys.exit(1 if bad else 0)

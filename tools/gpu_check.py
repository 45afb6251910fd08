"""Development helper (GPU box): run every parity scenario, print the first mismatch of each instead of stopping."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["OMP_NUM_THREADS"] = "1"
import torch  # noqa: F401,E402  (first, so the engine shares torch's HIP runtime)
import helpers as H  # noqa: E402

H.ensure_oracle()
names = sys.argv[1:] or sorted(k for k, sc in H.scenarios().items() if sc.engine)
bad = 0
for name in names:
    sc = H.scenarios()[name]
    t = time.time()
    try:
        got = H.run(sc, H.HIP_LIB)
        want = H.run(sc, H.ORACLE_LIB)
        H.assert_same(want, got, name)
        print("OK  ", name, "%d steps %.2fs" % (len(got), time.time() - t), flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", name, str(e)[:400], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)

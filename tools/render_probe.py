"""Development probe (GPU box): duration of the observation render launch at the bench workload under engine knobs.
usage: render_probe.py            -> runs the matrix below, one subprocess per configuration
       render_probe.py --one      -> one measurement in this process (knobs from the environment)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def one():
    import torch
    import magent_amd
    n = int(os.environ.get("PROBE_N", "400000"))
    env = magent_amd.GridWorld("battle", map_size=1000)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=n)
    dev = torch.device("cuda", 0)
    bf16 = os.environ.get("PROBE_BF16", "0") == "1"
    views = [torch.empty((n, 13, 13, 8), dtype=torch.bfloat16, device=dev) if bf16 else torch.empty((n, 13, 13, 7), device=dev) for _ in hs]
    feats = [torch.empty((n, 34), device=dev) for _ in hs]
    g = env.get_observation_device_bf16 if bf16 else env.get_observation_device
    f = lambda h, v_, f_: g(h, views[h.value], feats[h.value])
    view = feat = None
    for _ in range(5):
        for h in hs: f(h, view, feat)
    env.sync()
    env.profile_enable(2); env.profile_read("render")
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        for h in hs: f(h, view, feat)
    env.sync()
    wall = (time.perf_counter() - t0) / (2 * K)
    k, ms = env.profile_read("render")
    per = ms / k
    bytes_ = n * (13 * 13 * 16 if bf16 else 13 * 13 * 28) + n * 136
    print("%-60s %.4f ms/launch  %.0f GB/s  (wall %.4f)" % (os.environ.get("PROBE_LABEL", ""), per, bytes_ / per / 1e6, wall * 1e3), flush=True)

if "--one" in sys.argv:
    one()
else:
    configs = []
    for rep in range(2):
        configs.append(({"MAGENT_TUNE": "render=0", "PROBE_BF16": "0"}, "generic"))
        for pad in ("0", "24000", "44000", "70000", "150000"):
            for span in ("16", "32", "128"):
                e = {"MAGENT_RENDER_FAST": "1", "PROBE_BF16": "0", "MAGENT_RENDER_SPAN": span, "MAGENT_RENDER_PAD": pad}
                configs.append((e, "fast span=%s pad=%s" % (span, pad)))
    for e, label in configs:
        env = dict(os.environ, PROBE_LABEL=label, **e)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)

"""Is the render's launch time a property of the launch alone?  The same launch (battle 1000 x 1000, 400k agents, float32) back to back
without gaps, with a HIP event pair around every launch, and with idle time between launches (development probe, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import magent_amd
from magent_amd.builtin.config import _games

cfg = _games.make("battle", 1000)
env = magent_amd.GridWorld(cfg)
env.set_seed(12345); env.reset()
hs = env.get_handles()
for h in hs:
    env.add_agents(h, "random", n=400000)
dev = torch.device("cuda", 0)
n = env.get_num(hs[0])
view = torch.empty((n,) + env.get_view_space(hs[0]), device=dev)
feat = torch.empty((n,) + env.get_feature_space(hs[0]), device=dev)
nbytes = view.numel() * 4 + feat.numel() * 4
def loop(k, events, idle_us):
    env.profile_enable(2 if events else 0)
    env.profile_read("render")
    env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        env.get_observation_device(hs[0], view, feat)
        if idle_us:
            env.sync(); t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < idle_us: pass
    env.sync()
    wall = (time.perf_counter() - t0) / k * 1e3
    nl, ms = env.profile_read("render")
    return wall, (ms / nl if nl else None)
for k in (50, 400, 2000):
    for events, idle in ((False, 0), (True, 0), (True, 100), (True, 1000)):
        wall, ev = loop(k, events, idle)
        bw = lambda ms: nbytes / (ms * 1e-3) / 1e12 if ms else None
        print("launches %5d events %-5s idle %5d us : wall %.4f ms/launch (%.2f TB/s incl. gaps)   by events %s ms (%s TB/s)" % (
            k, events, idle, wall, bw(wall), None if ev is None else round(ev, 4), None if ev is None else round(bw(ev), 2)), flush=True)

"""What slows the render inside the cycle (0.328 ms per 400k agents against 0.304 alone)?  Back-to-back launches, wall time per render:
alone, two groups alternating (two 1.9 GB tensors), with set_action between, with a full step + clear_dead between (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import magent_amd
from magent_amd.builtin.config import _games

def world():
    cfg = _games.make("battle", 1000)
    env = magent_amd.GridWorld(cfg)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=400000)
    return env, hs
env, hs = world()
dev = torch.device("cuda", 0)
n = [env.get_num(h) for h in hs]
view = [torch.empty((n[g],) + env.get_view_space(hs[g]), device=dev) for g in range(2)]
feat = [torch.empty((n[g],) + env.get_feature_space(hs[g]), device=dev) for g in range(2)]
acts = [torch.zeros(n[g], dtype=torch.int32, device=dev) for g in range(2)]      # action 0 everywhere: nobody moves, nobody attacks, nobody dies
junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)                     # 256 MB
def timed(k, body):
    env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): body()
    env.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
def ev_render(k, body):      # the renders alone by HIP events inside the same loop
    env.profile_enable(2); env.profile_read("render")
    timed(k, body)
    nl, ms = env.profile_read("render"); env.profile_enable(0)
    return ms / nl
K = 300
r0 = lambda: env.get_observation_device(hs[0], view[0], feat[0])
def r01(): env.get_observation_device(hs[0], view[0], feat[0]); env.get_observation_device(hs[1], view[1], feat[1])
def r0s1s(): 
    env.get_observation_device(hs[0], view[0], feat[0]); env.set_action_device(hs[0], acts[0])
    env.get_observation_device(hs[1], view[1], feat[1]); env.set_action_device(hs[1], acts[1])
    # (set_action twice without a step would append: so a step follows)
def cycle():
    r0s1s(); env.step(); env.clear_dead()
def cycle_nostep_junk():
    env.get_observation_device(hs[0], view[0], feat[0]); junk.add_(1.0)
print("render g0 alone                         : %.4f ms per render" % timed(K, r0))
print("render g0, g1 alternating               : %.4f ms per render" % (timed(K, r01) / 2))
print("render g0 + 512 MB of torch traffic     : %.4f ms per pair (the add_ alone: %.4f)" % (timed(K, cycle_nostep_junk), timed(K, lambda: junk.add_(1.0))))
t_cycle = timed(K, cycle)
print("full cycle (nobody acts: action 0)       : %.4f ms per cycle" % t_cycle)
print("  renders inside that cycle, by events    : %.4f ms per render" % ev_render(K, cycle))
print("render g0 alone again                    : %.4f ms per render;  by events %.4f" % (timed(K, r0), ev_render(K, r0)))

"""Which part of a real step makes the NEXT render slower than the render alone?  Cycles with different action sets; meant to run under
`rocprofv3 --kernel-trace` (kernel begin / end timestamps; no HIP events anywhere).  Prints markers via torch fills of distinct sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import magent_amd
from magent_amd.builtin.config import _games
dev = torch.device("cuda", 0)
which = sys.argv[1]
cfg = _games.make("battle", 1000)
env = magent_amd.GridWorld(cfg)
env.set_seed(12345); env.reset()
hs = env.get_handles()
for h in hs:
    env.add_agents(h, "random", n=400000)
n0 = [env.get_num(h) for h in hs]
view = [torch.empty((n0[g],) + env.get_view_space(hs[g]), device=dev) for g in range(2)]
feat = [torch.empty((n0[g],) + env.get_feature_space(hs[g]), device=dev) for g in range(2)]
rew = [torch.empty(n0[g], device=dev) for g in range(2)]
gen = torch.Generator(device=dev); gen.manual_seed(0)
def actions(g):
    n = n0[g]
    if which == "none": return torch.zeros(n, dtype=torch.int32, device=dev)
    if which == "moves": return torch.randint(0, 13, (n,), dtype=torch.int32, device=dev, generator=gen)
    if which == "attacks": return torch.randint(13, 21, (n,), dtype=torch.int32, device=dev, generator=gen)
    return torch.randint(0, 21, (n,), dtype=torch.int32, device=dev, generator=gen)
sets = [[actions(g) for g in range(2)] for _ in range(8)]
torch.cuda.synchronize()
for s in range(14):
    for g in range(2):
        env.get_observation_device(hs[g], view[g], feat[g])
        env.set_action_device(hs[g], sets[s % 8][g])
    env.step()
    for g in range(2):
        env.get_reward_device(hs[g], rew[g])
    env.clear_dead()
env.sync()
print(which, [env.get_num(h) for h in hs])

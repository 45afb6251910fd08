"""Development probe (GPU box): how much of the step's head is the price of agents being stored in PLACEMENT order?

Same battle worlds, same positions; once with the agents added in the random order `add_agents("random")` produces (the bench
workload), once added in spatial order (tiles of T x T cells, row-major): every per-agent map access is then local.  Prints the
cycle time and the phase breakdown of both.  Upper bound for what a spatially sorted internal slot order could give."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd

def positions(map_size, n):
    env = magent_amd.GridWorld("battle", map_size=map_size)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=n)
    pos = [env.get_pos(h).copy() for h in hs]
    env.close()
    return pos

def run(map_size, pos, steps, warm, label, tile=0):
    env = magent_amd.GridWorld("battle", map_size=map_size)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h, p in zip(hs, pos):
        if tile:
            key = ((p[:, 1] // tile) * 100000 + (p[:, 0] // tile)) * 10000 + (p[:, 1] % tile) * tile + p[:, 0] % tile
            p = p[np.argsort(key, kind="stable")]
        env.add_agents(h, "custom", pos=p)
    dev = torch.device("cuda", 0)
    n0 = [env.get_num(h) for h in hs]
    vs = [env.get_view_space(h) for h in hs]; fs = [env.get_feature_space(h) for h in hs]
    views = [torch.empty((n0[g],) + vs[g], device=dev) for g in range(2)]
    feats = [torch.empty((n0[g],) + fs[g], device=dev) for g in range(2)]
    rew = [torch.empty(n0[g], device=dev) for g in range(2)]
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    acts = [[torch.randint(21, (n0[g],), dtype=torch.int32, device=dev, generator=gen) for g in range(2)] for _ in range(8)]
    torch.cuda.synchronize()
    def cyc(s):
        for g, h in enumerate(hs):
            env.get_observation_device(h, views[g], feats[g]); env.set_action_device(h, acts[s % 8][g])
        env.step()
        for g, h in enumerate(hs):
            env.get_reward_device(h, rew[g])
        env.clear_dead()
    for s in range(warm): cyc(s)
    env.sync(); t0 = time.perf_counter()
    for s in range(warm, warm + steps): cyc(s)
    env.sync(); dt = (time.perf_counter() - t0) / steps
    env.profile_enable(1)
    names = ("render", "features", "paint", "minimap", "attack", "move", "turn", "set_action", "step", "rules", "clear_dead")
    for nm in names: env.profile_read(nm)
    K = 5
    for s in range(K): cyc(s)
    env.sync()
    bd = {}
    for nm in names:
        k, ms = env.profile_read(nm)
        if k: bd[nm] = round(ms / K * 1e3, 1)
    env.profile_enable(False)
    print("%-34s %8.1f us/cycle  breakdown(us) %s  head %.1f" % (label, dt * 1e6, bd, sum(v for k, v in bd.items() if k != "render")), flush=True)
    env.close()

for map_size, n, steps, warm in ((1000, 400000, 20, 5), (200, 2000, 300, 30)):
    pos = positions(map_size, n)
    run(map_size, pos, steps, warm, "%d^2 2x%d placement order" % (map_size, n))
    for tile in (8, 32):
        run(map_size, pos, steps, warm, "%d^2 2x%d sorted, tiles of %d" % (map_size, n, tile), tile)

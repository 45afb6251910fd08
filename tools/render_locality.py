"""Development probe (GPU box): how much of the large-map render's time is the price of agents being stored in PLACEMENT order?

The reference's own 1M harness (scripts/test/test_1m.py: pursuit-like game on a 4472 x 4472 map, 500k 2x2 predators + 500k prey placed at
random): the painted map is 80 MB, every window row of a randomly placed agent is an L2 miss.  Same world, same positions, once in
placement order and once with the agents ADDED in spatial order (tiles of T x T cells): reads become local AND the stores stay in
agent order -- the upper bound for any spatially binned processing order.  Prints ms per render launch and the fraction of the HBM peak."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd
from magent_amd.builtin.config import _games

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
MAP = int((2 * N * 20) ** 0.5)

def world(pos=None, tile=0):
    env = magent_amd.GridWorld(_games.make("pursuit", MAP))
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    if pos is None:
        env.add_walls(method="random", n=2 * N // 10)
        for h in reversed(hs):
            env.add_agents(h, "random", n=N)
    else:
        for h, p in zip(hs, pos):
            if tile:
                key = ((p[:, 1] // tile) * 100000 + (p[:, 0] // tile)) * 100000 + (p[:, 1] % tile) * tile + p[:, 0] % tile
                p = p[np.argsort(key, kind="stable")]
            env.add_agents(h, "custom", pos=p)
    return env, hs

def time_render(env, hs, label):
    dev = torch.device("cuda", 0)
    for g, h in enumerate(hs):
        n = env.get_num(h)
        vs, fs = env.get_view_space(h), env.get_feature_space(h)
        view = torch.empty((n,) + vs, device=dev); feat = torch.empty((n,) + fs, device=dev)
        for _ in range(40):
            env.get_observation_device(h, view, feat)
        env.sync()
        env.profile_enable(2); env.profile_read("render"); env.profile_read("features")
        for _ in range(40):
            env.get_observation_device(h, view, feat)
        env.sync()
        k, ms = env.profile_read("render")
        env.profile_enable(False)
        b = n * 4 * (vs[0] * vs[1] * vs[2] + fs[0])
        print("%-40s group %d: %d agents, %.4f ms per launch, %.0f GB/s = %.3f of peak" % (label, g, n, ms / k, b / (ms / k * 1e-3) / 1e9, b / (ms / k * 1e-3) / 1e9 / 8000), flush=True)
        del view, feat

env, hs = world()
pos = [env.get_pos(h).copy() for h in hs]
time_render(env, hs, "placement order (random)")
env.close()
for tile in (16, 64):
    env, hs = world(pos, tile)
    time_render(env, hs, "added in tiles of %d" % tile)
    env.close()

"""Development helper (GPU box): K independent small environments on ONE GPU driven from ONE host thread:
observations / actions are enqueued per environment (each on its own stream), magent_amd.step_many overlaps the steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
MAP, N, STEPS = 200, 2000, 100
dev = torch.device("cuda", 0)
envs, bufs = [], []
for k in range(K):
    env = magent_amd.GridWorld("battle", map_size=MAP)
    env.set_seed(1000 + k); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=N)
    envs.append(env)
    bufs.append(([torch.empty((N, 13, 13, 7), device=dev) for _ in hs], [torch.empty((N, 34), device=dev) for _ in hs],
                 [torch.empty(N, device=dev) for _ in hs]))
acts = [[torch.randint(21, (N,), dtype=torch.int32, device=dev) for _ in range(2)] for _ in range(8)]
torch.cuda.synchronize()
total, t0 = 0, None
for s in range(STEPS + 10):
    if s == 10:
        for e in envs: e.sync()
        t0 = time.perf_counter(); total = 0
    for e, (views, feats, rew) in zip(envs, bufs):
        for g, h in enumerate(e.get_handles()):
            total += e.get_num(h)
            e.get_observation_device(h, views[g], feats[g])
            e.set_action_device(h, acts[s % 8][g])
    magent_amd.step_many(envs)
    for e, (views, feats, rew) in zip(envs, bufs):
        for g, h in enumerate(e.get_handles()):
            e.get_reward_device(h, rew[g])
        e.clear_dead()
for e in envs: e.sync()
dt = time.perf_counter() - t0
print("K=%d envs, one thread: %.2fM agent-steps/s aggregate, %.3f ms per round of %d env-steps" % (K, total / dt / 1e6, dt / STEPS * 1e3, K))

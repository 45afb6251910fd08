"""Development helper (GPU box): K independent small environments on ONE GPU, one host thread + one HIP stream each.
Small worlds are launch/sync-latency bound; concurrent environments fill the gaps."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
MAP, N, STEPS = 200, 2000, 150
dev = torch.device("cuda", 0)


def worker(k, out):
    env = magent_amd.GridWorld("battle", map_size=MAP)
    env.set_seed(1000 + k); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=N)
    views = [torch.empty((N, 13, 13, 7), device=dev) for _ in hs]
    feats = [torch.empty((N, 34), device=dev) for _ in hs]
    rew = [torch.empty(N, device=dev) for _ in hs]
    acts = [[torch.randint(21, (N,), dtype=torch.int32, device=dev) for _ in hs] for _ in range(STEPS + 10)]
    torch.cuda.synchronize()
    barrier.wait()
    t0 = time.perf_counter(); total = 0
    for s in range(STEPS + 10):
        if s == 10:
            env.sync(); barrier.wait(); t0 = time.perf_counter(); total = 0
        for g, h in enumerate(hs):
            total += env.get_num(h)
            env.get_observation_device(h, views[g], feats[g])
            env.set_action_device(h, acts[s][g])
        env.step()
        for g, h in enumerate(hs):
            env.get_reward_device(h, rew[g])
        env.clear_dead()
    env.sync()
    out[k] = (total, time.perf_counter() - t0)


barrier = threading.Barrier(K)
out = {}
ts = [threading.Thread(target=worker, args=(k, out)) for k in range(K)]
[t.start() for t in ts]; [t.join() for t in ts]
tot = sum(v[0] for v in out.values()); dt = max(v[1] for v in out.values())
print("K=%d envs: %.2fM agent-steps/s aggregate, %.3f ms per env-step" % (K, tot / dt / 1e6, dt / STEPS * 1e3))

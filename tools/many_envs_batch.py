"""Development helper (GPU box): K small environments cycled by magent_amd.EnvBatch (threads inside the library)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
MAP, N, STEPS = 200, 2000, 100
dev = torch.device("cuda", 0)
envs = []
for k in range(K):
    env = magent_amd.GridWorld("battle", map_size=MAP)
    env.set_seed(1000 + k); env.reset()
    for h in env.get_handles():
        env.add_agents(h, "random", n=N)
    envs.append(env)
views = [[torch.empty((N, 13, 13, 7), device=dev) for _ in range(2)] for _ in envs]
feats = [[torch.empty((N, 34), device=dev) for _ in range(2)] for _ in envs]
rews = [[torch.empty(N, device=dev) for _ in range(2)] for _ in envs]
acts = [[[torch.randint(21, (N,), dtype=torch.int32, device=dev) for _ in range(2)] for _ in envs] for _ in range(4)]
batch = magent_amd.EnvBatch(envs, n_threads=T)
batch.order_streams = os.environ.get("ORDER_STREAMS", "0") == "1"   # (this loop orders by env.sync(): no torch work touches the buffers in between)
us = np.zeros(4, dtype=np.float32)
view_p, feat_p, rew_p = batch.pointers(views), batch.pointers(feats), batch.pointers(rews)      # fixed buffers: pointer arrays built once (the tensors stay alive)
act_ptrs = [batch.pointers(a) for a in acts]
torch.cuda.synchronize()
total, t0 = 0, time.perf_counter()
for s in range(STEPS + 10):
    if s == 10:
        for e in envs: e.sync()
        envs[0]._lib.env_get_info(envs[0].game, 0, b"batch_host_us", us.ctypes.data)      # reset: warm-up rounds allocate
        t0 = time.perf_counter(); total = 0
    total += int(batch.nums_array().sum())
    batch.cycle(view_p, feat_p, act_ptrs[s % 4], rew_p)
for e in envs: e.sync()
dt = time.perf_counter() - t0
envs[0]._lib.env_get_info(envs[0].game, 0, b"batch_host_us", us.ctypes.data)
print("   host us per round: prepare %.1f, copy + launches %.1f, wait for the first record %.1f, other records %.1f" % tuple(us))
print("K=%d envs, %d library threads: %.2fM agent-steps/s aggregate, %.3f ms per round" % (K, T, total / dt / 1e6, dt / STEPS * 1e3))

"""Development helper (GPU box): K small environments cycled by magent_amd.EnvBatch (threads inside the library)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
torch.cuda.synchronize()
total, t0 = 0, time.perf_counter()
for s in range(STEPS + 10):
    if s == 10:
        for e in envs: e.sync()
        t0 = time.perf_counter(); total = 0
    total += sum(e.get_num(h) for e in envs for h in e.get_handles())
    batch.cycle(views, feats, acts[s % 4], rews)
for e in envs: e.sync()
dt = time.perf_counter() - t0
print("K=%d envs, %d library threads: %.2fM agent-steps/s aggregate, %.3f ms per round" % (K, T, total / dt / 1e6, dt / STEPS * 1e3))

"""Development probe (GPU box): K battle 200x200 / 2x2000 worlds through EnvBatch.cycle, for rocprofv3 --kernel-trace --stats.
usage: python tools/probe/batch_cycle.py [K=32] [steps=200] [map=200] [agents=2000]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, magent_amd
from magent_amd.builtin.config import _games
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
MAP = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
dev = torch.device("cuda", 0)
envs = []
for k in range(K):
    env = magent_amd.GridWorld(_games.make("battle", MAP)); env.set_seed(5000 + k); env.reset()
    for h in env.get_handles(): env.add_agents(h, "random", n=N)
    envs.append(env)
gen = torch.Generator(device=dev); gen.manual_seed(99)
views = [[torch.empty((N, 13, 13, 7), device=dev) for _ in range(2)] for _ in envs]
feats = [[torch.empty((N, 34), device=dev) for _ in range(2)] for _ in envs]
rews = [[torch.empty(N, device=dev) for _ in range(2)] for _ in envs]
acts = [[[torch.randint(21, (N,), dtype=torch.int32, device=dev, generator=gen) for _ in range(2)] for _ in envs] for _ in range(4)]
torch.cuda.synchronize()
batch = magent_amd.EnvBatch(envs, n_threads=8); batch.order_streams = False
vp, fp, rp = batch.pointers(views), batch.pointers(feats), batch.pointers(rews)
ap = [batch.pointers(a) for a in acts]
for s in range(steps + 20):
    if s == 20:
        for e in envs: e.sync()
        t0 = time.perf_counter()
    batch.cycle(vp, fp, ap[s % 4], rp)
for e in envs: e.sync()
print("%d envs: %.4f ms per round" % (K, (time.perf_counter() - t0) / steps * 1e3))

"""Development helper (GPU box): phase timeline inside the one-launch step (k_step_solo) for a small battle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd

MAP = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
game = sys.argv[3] if len(sys.argv) > 3 else "battle"
dev = torch.device("cuda", 0)
print("sharedMemPerBlock", torch.cuda.get_device_properties(0).shared_memory_per_block, getattr(torch.cuda.get_device_properties(0), "shared_memory_per_block_optin", None))
env = magent_amd.GridWorld(game, map_size=MAP)
env.set_seed(1); env.reset()
hs = env.get_handles()
for h in hs:
    env.add_agents(h, "random", n=N)
names = ["start", "set_action", "draw", "chase", "rank", "eval", "apply", "hit reset", "starve+cand", "claim", "init", "jump", "commit", "rules", "finish", "cycle tail"]
acc = None
for s in range(30):
    for h in hs:
        n = env.get_num(h)
        obs = env.get_observation_device(h); env.sync()   # (the buffers must outlive the render: they come from torch's allocator)
        acts = torch.randint(env.get_action_space(h)[0], (n,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        env.set_action_device(h, acts)
        env.sync()
    env.step()
    buf = np.zeros(48, dtype=np.int32)
    env._lib.env_get_info(env.game, 0, b"step_marks", buf.ctypes.data)
    m = buf[1:1 + buf[0]].astype(np.float64) / 1e3
    if s >= 10:
        acc = m if acc is None else acc + m
    env.clear_dead()
acc /= 20
print("stats (fallback, attack rounds, move rounds)", env.engine_stats())
prev = 0.0
for k, t in enumerate(acc):
    print("%-8s %7.2f us  (+%.2f)" % (names[k] if k < len(names) else "m%d" % k, t, t - prev))
    prev = t

"""Development helper (GPU box): rate of the reference (host-buffer) ABI at the bench workload, PCIe included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import magent_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
env = magent_amd.GridWorld("battle", map_size=1000)
env.set_seed(12345); env.reset()
hs = env.get_handles()
for h in hs:
    env.add_agents(h, "random", n=n)
rs = np.random.RandomState(0)
tot, steps, t_obs = 0.0, 0, 0.0
for step in range(6):
    acts = [rs.randint(21, size=env.get_num(h)).astype(np.int32) for h in hs]
    nn = sum(env.get_num(h) for h in hs)
    t0 = time.perf_counter()
    for h, a in zip(hs, acts):
        t1 = time.perf_counter(); env.get_observation(h); t_obs += (time.perf_counter() - t1) if step >= 2 else 0
        env.set_action(h, a)
    env.step()
    for h in hs:
        env.get_reward(h); env.get_alive(h)
    env.clear_dead()
    dt = time.perf_counter() - t0
    if step >= 2:
        tot += dt; steps += 1; agents = nn
print("host ABI: %.1f ms/step, %.2e agent-steps/s, get_observation %.1f ms per call (%.1f GB/s to numpy)" % (
    tot / steps * 1e3, agents * steps / tot, t_obs / steps / 2 * 1e3, agents / 2 * 4868 / (t_obs / steps / 2) / 1e9))

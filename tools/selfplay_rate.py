"""Development helper (GPU box): BASELINE config 5 in miniature -- battle 1000x1000, 2 x 400k agents, both sides acting
through the PyTorch DQN (inference only, epsilon-greedy), observations and actions staying in HBM.  Prints how the step
time splits between the engine and the policy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import magent_amd
from magent_amd.builtin.torch_model import DeepQNetwork

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp32"
env = magent_amd.GridWorld("battle", map_size=1000, device_obs=True)
env.set_seed(12345); env.reset()
hs = env.get_handles()
for h in hs:
    env.add_agents(h, "random", n=n)
models = [DeepQNetwork(env, h, "m%d" % i, memory_size=16, infer_batch_size=65536, infer_dtype="bf16" if "cells" in sys.argv[4:5] or dtype == "bf16" else "f32")
          for i, h in enumerate(hs)]
cells = len(sys.argv) > 4 and sys.argv[4] == "cells" and all(m._hip is not None for m in models)
env.use_bf16_observations(cells)       # views as bf16 cells: the MFMA kernels' operands, 2.7 KB per agent instead of 4.7
t_env = t_pol = 0.0
agent_steps = 0
for s in range(steps + 2):
    if s == 2:
        torch.cuda.synchronize(); t_env = t_pol = 0.0; agent_steps = 0; t_all = time.perf_counter()
    for h, m in zip(hs, models):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        obs = env.get_observation(h); env.sync()
        t1 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
            acts = m.infer_action(obs, None, policy="e_greedy", eps=0.1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        env.set_action(h, acts)
        agent_steps += env.get_num(h)
        t_env += t1 - t0; t_pol += t2 - t1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.step(); [env.get_reward(h) for h in hs]; env.clear_dead(); env.sync()
    t_env += time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t_all
print("self-play (%s policy%s): %.1f ms/step = engine %.2f ms + policy %.1f ms; %.2e agent-steps/s"
      % (dtype, ", bf16-cell observations" if cells else "", dt / steps * 1e3, t_env / steps * 1e3, t_pol / steps * 1e3, agent_steps / dt))

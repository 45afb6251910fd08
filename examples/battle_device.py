"""A battle played entirely in HBM: observations, policy and actions never leave the GPU.

    python examples/battle_device.py [--map_size 200] [--n 2000] [--steps 100] [--policy random|dqn]

The loop is the reference's (examples/train_battle.py:61-109: get_observation -> infer_action -> set_action per group,
step, get_reward, clear_dead); `device_obs=True` makes get_observation return torch tensors on the engine's GPU, and
set_action takes an int32 tensor that lives there."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import magent_amd  # noqa: E402
from magent_amd.builtin.rule_model import RandomActor  # noqa: E402
from magent_amd.builtin.torch_model import DeepQNetwork  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map_size", type=int, default=200)
    ap.add_argument("--n", type=int, default=2000, help="agents per side")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--policy", choices=["random", "dqn"], default="random")
    ap.add_argument("--infer-dtype", choices=["f32", "bf16"], default="f32", help="bf16: the MFMA inference kernels (magent_amd/csrc/policy.hip)")
    args = ap.parse_args()

    env = magent_amd.GridWorld("battle", map_size=args.map_size, device_obs=True)
    env.set_seed(0)
    env.reset()
    handles = env.get_handles()
    for h in handles:
        env.add_agents(h, "random", n=args.n)
    if args.policy == "dqn":
        models = [DeepQNetwork(env, h, "side%d" % i, memory_size=16, infer_dtype=args.infer_dtype) for i, h in enumerate(handles)]
        # --infer-dtype bf16: the forward pass runs on the hand-written MFMA kernels (magent_amd/csrc/policy.hip); they take the views as bf16 cells of
        # 8 channels, which the engine can render directly (2.7 KB per agent instead of 4.7, nothing to convert)
        env.use_bf16_observations(all(m._hip is not None for m in models))
    else:
        models = [RandomActor(env, h, seed=i) for i, h in enumerate(handles)]

    total = [0.0 for _ in handles]
    agent_steps, t0 = 0, time.perf_counter()
    for step in range(args.steps):
        for h, m in zip(handles, models):
            obs = env.get_observation(h)                       # (view [n, 13, 13, 7] -- or bf16 [n, 13, 13, 8] --, feature [n, 34]) on the GPU
            acts = m.infer_action(obs, None, policy="e_greedy", eps=0.1)
            env.set_action(h, acts)
            agent_steps += env.get_num(h)
        done = env.step()
        for i, h in enumerate(handles):
            total[i] += float(env.get_reward(h).sum())
        env.clear_dead()
        if step % 20 == 0 or done:
            print("step %4d  alive %s  reward so far %s" % (step, [env.get_num(h) for h in handles], [round(t, 1) for t in total]))
        if done:
            break
    env.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d agent-steps in %.2f s = %.2e agent-steps/s (%s policy)" % (agent_steps, dt, agent_steps / dt, args.policy))


if __name__ == "__main__":
    main()

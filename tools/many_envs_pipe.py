"""Development helper (GPU box): K environments of a BASELINE configuration cycled by ONE magent_amd.EnvBatch -- the batched pipeline
(pipe.hip) against the forms it replaces.  The engine reads MAGENT_TUNE once per process: run it once per variant.

    python tools/many_envs_pipe.py battle 200 2000 32 [steps]        # BASELINE config 2 x 32: MAGENT_TUNE=batch_pipe=0 -> one workgroup per world
    python tools/many_envs_pipe.py gather 500 100000 8 [steps]       # BASELINE config 4 x 8 on one GPU: MAGENT_TUNE=batch_pipe=0 -> one by one

Prints one JSON line: ms per round, aggregate agent-steps/s, how many of the environments went through the batched pipeline."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import magent_amd
from magent_amd.builtin.config import _games

game, MAP, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 100
WARM = 10
dev = torch.device("cuda", 0)
envs = []
for k in range(K):
    env = magent_amd.GridWorld(_games.make(game, MAP))
    env.set_seed(1000 + k); env.reset()
    hs = env.get_handles()
    if game == "gather":          # examples/train_gather.py: food, then agents; only the agents act and are observed
        env.add_agents(hs[0], "random", n=N // 5)
        env.add_agents(hs[1], "random", n=N)
    else:
        for h in hs:
            env.add_agents(h, "random", n=N)
    envs.append(env)
hs = envs[0].get_handles()
acting = [1] if game == "gather" else list(range(len(hs)))
cap = [envs[0].get_num(h) for h in hs]
views = [[torch.empty((cap[g],) + envs[0].get_view_space(h), device=dev) if g in acting else None for g, h in enumerate(hs)] for _ in envs]
feats = [[torch.empty((cap[g],) + envs[0].get_feature_space(h), device=dev) if g in acting else None for g, h in enumerate(hs)] for _ in envs]
rews = [[torch.empty(cap[g], device=dev) if g in acting else None for g in range(len(hs))] for _ in envs]
na = [envs[0].get_action_space(h)[0] for h in hs]
acts = [[[torch.randint(na[g], (cap[g],), dtype=torch.int32, device=dev) if g in acting else None for g in range(len(hs))] for _ in envs] for _ in range(4)]
batch = magent_amd.EnvBatch(envs, n_threads=8)
batch.order_streams = os.environ.get("ORDER_STREAMS", "0") == "1"      # (default here: this loop orders by env.sync(), no torch work touches the buffers in between)
view_p, feat_p, rew_p = batch.pointers(views), batch.pointers(feats), batch.pointers(rews)
act_ptrs = [batch.pointers(a) for a in acts]
us = np.zeros(4, dtype=np.float32)
torch.cuda.synchronize()
total, t0 = 0, time.perf_counter()
for s in range(STEPS + WARM):
    if s == WARM:
        for e in envs: e.sync()
        envs[0]._lib.env_get_info(envs[0].game, 0, b"batch_host_us", us.ctypes.data)
        t0 = time.perf_counter(); total = 0
    total += int(batch.nums_array().sum() if len(acting) == len(hs) else batch.nums_array()[:, acting].sum())
    batch.cycle(view_p, feat_p, act_ptrs[s % 4], rew_p)
for e in envs: e.sync()
dt = time.perf_counter() - t0
envs[0]._lib.env_get_info(envs[0].game, 0, b"batch_host_us", us.ctypes.data)
stats = np.array([e.pipeline_stats() for e in envs])
piped = int((stats[:, 6] > 0).sum())
hist = np.array([e.round_hist() for e in envs]).sum(axis=0)
print(json.dumps({"game": game, "map": MAP, "agents_per_env_at_start": cap, "agents_per_env_at_end": batch.nums()[0], "envs": K, "steps": STEPS,
                  "tune": os.environ.get("MAGENT_TUNE", ""), "ms_per_round": round(dt / STEPS * 1e3, 4), "agent_steps_per_s": round(total / dt),
                  "envs_in_batched_pipeline": piped, "host_us_per_round": [round(float(v), 1) for v in us],
                  "env_steps_two_pairs_one_pair_ran_out": [int(stats[:, 1].sum()), int(stats[:, 2].sum()), int(stats[:, 5].sum())], "last_changing_round_hist": [int(v) for v in hist]}))

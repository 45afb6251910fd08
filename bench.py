#!/usr/bin/env python
"""bench.py -- agent-steps/sec (step + observation) of the MI355X grid-world engine on the battle map.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 one rank per GPU under
torch.distributed.run.  One "step" = one pass of the hot path over one environment:

    for each group: get_observation (render into device tensors) ; set_action (device actions)
    step ; for each group: get_reward ; clear_dead

Inputs are resident in HBM when the timed region starts (actions pre-generated on the device, policy excluded;
outputs stay in device tensors).  Every rank owns an independent environment replica (seed 12345 + rank): the
engine does not shard one grid ("replicas only"), so scaling is weak and there is no data-path
collective unless --gather obs is given.

Workload at N=1: BASELINE.json config "battle 1000x1000, 2x500k" in the feasible interpretation C3(i) of
SURVEY.md 8d: 2 x 400k agents placed by the reference's `random` method (2 x 500k does not fit the 996,004
placeable cells).

The JSON line also carries
  roofline     : the observation-render kernel, algorithmic bytes (4*(VH*VW*C + F) per rendered agent) over its
                 HIP-event-timed duration on the engine's stream, against the 8 TB/s HBM peak;
  cpu_baseline : the compiled reference engine (oracle/_ref, kind "reference") or the CPU restatement (kind
                 "port") timed on this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAP_SIZE = 1000
N_PER_GROUP = 400000
WORKLOAD = "battle 1000x1000, 2x400k agents, random placement (C3(i): 2x500k does not fit 996,004 cells), random actions"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline_worker(args):
    """child process: time the CPU engine on a bounded sample of the same workload; prints one JSON line"""
    import numpy as np
    import magent_amd
    # the CPU checker behind the same wrapper (cpu_baseline leg only; the product class never loads anything else than the HIP engine)
    CpuWorld = type("CpuWorld", (magent_amd.GridWorld,), {"_engine_path": args.cpu_lib})
    env = CpuWorld("battle", map_size=args.map_size)
    env.set_seed(12345)
    env.reset()
    handles = env.get_handles()
    for h in handles:
        env.add_agents(h, "random", n=args.agents)
    rs = np.random.RandomState(0)
    agent_steps, elapsed = 0, 0.0
    for step in range(args.cpu_steps + 1):
        acts = [rs.randint(21, size=env.get_num(h)).astype(np.int32) for h in handles]
        n = sum(env.get_num(h) for h in handles)
        t0 = time.time()
        for h, a in zip(handles, acts):
            env.get_observation(h)
            env.set_action(h, a)
        env.step()
        for h in handles:
            env.get_reward(h)
        env.clear_dead()
        dt = time.time() - t0
        if step >= 1:  # first step pays numpy first-touch / OpenMP pool start-up
            agent_steps += n
            elapsed += dt
    print(json.dumps({"agent_steps": agent_steps, "seconds": elapsed}))


def run_cpu_baseline(map_size=MAP_SIZE, agents=N_PER_GROUP, steps=2):
    ref = os.path.join(ROOT, "oracle", "_ref", "libmagent_ref.so")
    port = os.path.join(ROOT, "oracle", "liboracle.so")
    if os.path.exists(ref):
        lib, kind = ref, "reference"
    elif os.path.exists(port):
        lib, kind = port, "port"
    else:
        return None
    ncpu = os.cpu_count() or 1
    best = None
    threads = [1] if kind == "port" else sorted({1, max(1, ncpu // 2)})
    for th in threads:
        env = dict(os.environ, OMP_NUM_THREADS=str(th), MAGENT_AMD_NO_TORCH="1")
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cpu-lib", lib,
                                  "--cpu-steps", str(steps), "--map-size", str(map_size), "--agents", str(agents)],
                                 env=env, capture_output=True, text=True, timeout=600)
            rec = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is a reported extra; never fail the bench for it
            sys.stderr.write("cpu_baseline(%d threads) failed: %r\n" % (th, e))
            continue
        rate = rec["agent_steps"] / rec["seconds"]
        if best is None or rate > best["value"]:
            best = {"value": rate, "unit": "agent-steps/s", "cores": th, "kind": kind,
                    "sample": "%d timed steps (+1 warm-up) of the same workload, %d OpenMP thread(s), host buffers "
                              "(reference ABI); best of threads %s" % (steps, th, threads)}
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--map-size", type=int, default=MAP_SIZE)
    ap.add_argument("--agents", type=int, default=N_PER_GROUP, help="agents per group")
    ap.add_argument("--workload", choices=["battle", "test_1m", "gather"], default="battle",
                    help="test_1m: the reference's own harness (scripts/test/test_1m.py): pursuit-like game, map sqrt(20 N), "
                         "N/10 walls, N/2 prey + N/2 2x2 predators, N = 2 * --agents; "
                         "gather: BASELINE config 4 (examples/train_gather.py: --agents agents + agents/5 food, only agents act)")
    ap.add_argument("--gather", choices=["none", "obs"], default="none",
                    help="obs: all_gather the observation tensors of every replica over RCCL each step")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to dry-run the N > 1 "
                         "code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernels with HIP events")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    ap.add_argument("--cpu-lib", default="")
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)

    import torch
    import torch.distributed as dist
    import magent_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()   # dry run: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the tiny timing reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from magent_amd.builtin.config import _games
    if args.workload == "test_1m":
        args.map_size = int((2 * args.agents * 20) ** 0.5)
        cfg = _games.make("pursuit", args.map_size)
    elif args.workload == "gather":
        cfg = _games.make("gather", args.map_size)
    else:
        cfg = _games.make("battle", args.map_size)
    cfg.set({"device_id": local_rank})
    env = magent_amd.GridWorld(cfg)
    env.set_seed(12345 + rank)
    env.reset()
    handles = env.get_handles()
    acting = list(range(len(handles)))
    if args.workload == "test_1m":
        env.add_walls(method="random", n=2 * args.agents // 10)
        for h in reversed(handles):
            env.add_agents(h, "random", n=args.agents)
    elif args.workload == "gather":   # group 0 = food (never observed, never acts), group 1 = agents
        env.add_agents(handles[0], "random", n=args.agents // 5)
        env.add_agents(handles[1], "random", n=args.agents)
        acting = [1]
    else:
        for h in handles:
            env.add_agents(h, "random", n=args.agents)
    n0 = [env.get_num(h) for h in handles]
    G = len(handles)
    vss = [env.get_view_space(h) for h in handles]
    fss = [env.get_feature_space(h) for h in handles]
    n_actions = [env.get_action_space(h)[0] for h in handles]
    view_bytes = [4 * v[0] * v[1] * v[2] for v in vss]   # per agent: what k_render writes (the dominant kernel)
    feat_bytes = [4 * f[0] for f in fss]                 # per agent: what k_features writes

    # caller-owned device buffers (the reference's ownership convention), sized once for the initial population
    views = [torch.empty((n0[g],) + vss[g], dtype=torch.float32, device=dev) for g in range(G)]
    feats = [torch.empty((n0[g],) + fss[g], dtype=torch.float32, device=dev) for g in range(G)]
    rewards = [torch.empty(n0[g], dtype=torch.float32, device=dev) for g in range(G)]
    total_steps = args.steps + args.warmup
    gen = torch.Generator(device=dev)
    gen.manual_seed(rank)
    actions = [[torch.randint(n_actions[g], (n0[g],), dtype=torch.int32, device=dev, generator=gen) for g in range(G)]
               for _ in range(total_steps + (0 if args.no_profile else 5))]
    from magent_amd import replicas
    do_gather = args.gather == "obs" and world > 1
    torch.cuda.synchronize()

    rendered = {"view": 0, "feat": 0}

    def one_step(s):
        n_now = 0
        for g, h in enumerate(handles):
            if g not in acting:
                continue
            n = env.get_num(h)
            n_now += n
            rendered["view"] += n * view_bytes[g]
            rendered["feat"] += n * feat_bytes[g]
            env.get_observation_device(h, views[g], feats[g])
            env.set_action_device(h, actions[s][g])
        if do_gather:   # the north star's batched-observation gather: every replica's view tensor to every rank
            env.sync()
            for g, h in enumerate(handles):
                src = views[g] if args.backend == "nccl" else views[g].cpu()
                replicas.gather_observations(src, env.get_num(h), capacity=n0[g])
        env.step()
        for g, h in enumerate(handles):
            if g in acting:
                env.get_reward_device(h, rewards[g])
        env.clear_dead()
        return n_now

    for s in range(args.warmup):
        one_step(s)
    env.sync()
    if not args.no_profile:
        # inside the timed region only the dominant kernel carries HIP events (an event pair costs ~10 us of stream time;
        # timing every phase of every step would add ~10 % to the step); the phase breakdown is taken afterwards
        env.profile_enable(2)
        for name in ("render", "features", "paint", "minimap", "attack", "move", "set_action", "starve", "rules", "clear_dead"):
            env.profile_read(name)
    rendered["view"] = rendered["feat"] = 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent_steps = 0
    for s in range(args.warmup, total_steps):
        n_now = one_step(s)
        agent_steps += n_now
    env.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    if world > 1:   # whole-job aggregate over the slowest replica's time
        elapsed = replicas.max_over_replicas(elapsed, device=red_dev)
        agent_steps = replicas.sum_over_replicas(agent_steps, device=red_dev)

    agents_at_end = [env.get_num(h) for h in handles]
    host_finished_steps = env.engine_stats()[0]   # steps whose optimistic rounds ran out (counted since reset, warm-up included)
    roofline, breakdown = None, {}
    if not args.no_profile:
        n_launch, ms = env.profile_read("render")
        n_feat, ms_feat = env.profile_read("features")
        if n_launch and ms > 0:
            # algorithmic bytes of the dominant kernel: every element of the observation written exactly once
            # (B_obs = 4*(VH*VW*C + F) per agent; the feature rows ride in the render launch)
            fused = n_feat == 0
            obs_bytes = rendered["view"] + (rendered["feat"] if fused else 0)
            achieved = obs_bytes / (ms * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "render_pmc.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": "k_render", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "launches": n_launch, "avg_launch_ms": round(ms / n_launch, 4),
                        "algorithmic_bytes_per_launch": int(obs_bytes / n_launch),
                        "obs_total_GBs": round((rendered["view"] + rendered["feat"]) / ((ms + ms_feat) * 1e-3) / 1e9, 1)}
        # phase breakdown: a few more steps of the same episode, OUTSIDE the timed region, with an event pair around every phase
        extra = min(5, len(actions) - total_steps)
        if extra > 0:
            env.profile_enable(1)
            for s in range(total_steps, total_steps + extra):
                one_step(s)
            env.sync()
            for name in ("paint", "minimap", "attack", "move", "set_action", "starve", "rules", "clear_dead"):
                k, t_ms = env.profile_read(name)
                if k:
                    breakdown[name + "_ms_per_step"] = round(t_ms / extra, 4)
            breakdown["note"] = "%d extra steps after the timed region" % extra
            env.profile_read("render"); env.profile_read("features")
        if roofline:
            breakdown["render_ms_per_step"] = round(ms / args.steps, 4)
            breakdown["features_ms_per_step"] = round(ms_feat / args.steps, 4)
        env.profile_enable(False)

    if rank == 0:
        rec = {
            "metric": "agent-steps/sec (step+obs) on battle map; bit-exact vs CPU ref",
            "value": agent_steps / elapsed,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOAD if (args.workload, args.map_size, args.agents) == ("battle", MAP_SIZE, N_PER_GROUP) else
                       ("reference test_1m.py harness: pursuit-like %dx%d, %d walls, %d prey + %d 2x2 predators" % (
                           args.map_size, args.map_size, 2 * args.agents // 10, args.agents, args.agents)
                        if args.workload == "test_1m" else
                        "gather %dx%d (train_gather.py), %d agents + %d food, only the agents act" % (
                            args.map_size, args.map_size, args.agents, args.agents // 5) if args.workload == "gather" else
                        "battle %dx%d, 2x%d agents, random placement, random actions" % (args.map_size, args.map_size, args.agents)),
                       "envs": world, "parallelism": "replicas x%d" % world, "gather": args.gather,
                       "agents_at_start": n0, "agents_at_end": agents_at_end,
                       "io": "device-resident (env_*_device C-ABI)", "steps_finished_by_host_driver": host_finished_steps},
            "roofline": roofline,
            "breakdown": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            small = args.map_size * args.map_size <= 250000
            rec["cpu_baseline"] = run_cpu_baseline(args.map_size, args.agents, steps=200 if small else 2) \
                if args.workload == "battle" else None
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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
PREHEAT_MS = 40.0       # per region, before the warm-up steps: back-to-back renders of step 0's observation (see measure)
EVENT_EVERY = 4         # timed region: the dominant kernel's launches of every 4th step carry HIP events (see measure / one_step)


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


def run_cpu_baseline(map_size=MAP_SIZE, agents=N_PER_GROUP, steps=4):
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
    # best of OMP_NUM_THREADS in {1, nproc//2 (the reference's default, c_lib.py:41), nproc}
    threads = [1] if kind == "port" else sorted({1, max(1, ncpu // 2), ncpu})
    for th in threads:
        env = dict(os.environ, OMP_NUM_THREADS=str(th), MAGENT_AMD_NO_TORCH="1")
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cpu-lib", lib,
                                  "--cpu-steps", str(steps if th > 1 or steps > 50 else min(steps, 3)), "--map-size", str(map_size), "--agents", str(agents)],
                                 env=env, capture_output=True, text=True, timeout=900)
            rec = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # the baseline is a reported extra; never fail the bench for it
            sys.stderr.write("cpu_baseline(%d threads) failed: %r\n" % (th, e))
            continue
        rate = rec["agent_steps"] / rec["seconds"]
        if best is None or rate > best["value"]:
            best = {"value": rate, "unit": "agent-steps/s", "cores": th, "kind": kind,
                    "sample": "%d timed steps (+1 warm-up; the 1-thread leg: 3 -- about 5 s per step there) of the same workload, %d OpenMP thread(s), host buffers "
                              "(reference ABI); best of threads %s; %.1f s of CPU wall time for this leg" % (steps if th > 1 or steps > 50 else min(steps, 3), th, threads, rec["seconds"])}
    return best


def small_world_extras(torch, magent_amd, dev, steps=200, warmup=20):
    """secondary lines for BASELINE config 2 (battle 200x200, 2x2000): the same cycle through the reference call sequence, through
    env_cycle_many (EnvBatch: two launches per cycle) for one environment, and for 8 / 32 environments on this one GPU"""
    from magent_amd.builtin.config import _games
    MAP, N = 200, 2000
    out = {}

    def make(seed):
        env = magent_amd.GridWorld(_games.make("battle", MAP))
        env.set_seed(seed); env.reset()
        for h in env.get_handles():
            env.add_agents(h, "random", n=N)
        return env

    gen = torch.Generator(device=dev); gen.manual_seed(99)
    # (`ordered`: EnvBatch's DEFAULT input ordering -- the environments' streams wait for torch's current stream every cycle; the other
    # lines switch it off because this loop orders by env.sync())
    for label, K, batched, ordered in (("calls", 1, False, False), ("cycle_1env", 1, True, False), ("cycle_8env", 8, True, False),
                                       ("cycle_32env", 32, True, False), ("cycle_32env_default_ordering", 32, True, True)):
        envs = [make(5000 + k) for k in range(K)]
        views = [[torch.empty((N, 13, 13, 7), device=dev) for _ in range(2)] for _ in envs]
        feats = [[torch.empty((N, 34), device=dev) for _ in range(2)] for _ in envs]
        rews = [[torch.empty(N, device=dev) for _ in range(2)] for _ in envs]
        acts = [[[torch.randint(21, (N,), dtype=torch.int32, device=dev, generator=gen) for _ in range(2)] for _ in envs] for _ in range(4)]
        torch.cuda.synchronize()
        batch = magent_amd.EnvBatch(envs, n_threads=8)
        batch.order_streams = ordered
        if batched:      # fixed buffers: the device-pointer arrays are built once, not 8 x K data_ptr() calls per cycle
            views_p, feats_p, rews_p = batch.pointers(views), batch.pointers(feats), batch.pointers(rews)
            acts_p = [batch.pointers(a) for a in acts]
        total, t0 = 0, 0.0
        for s in range(steps + warmup):
            if s == warmup:
                for e in envs:
                    e.sync()
                total, t0 = 0, time.perf_counter()
            total += int(batch.nums_array().sum())
            if batched:
                batch.cycle(views_p, feats_p, acts_p[s % 4], rews_p)
            else:
                e = envs[0]
                for g, h in enumerate(e.get_handles()):
                    e.get_observation_device(h, views[0][g], feats[0][g])
                    e.set_action_device(h, acts[s % 4][0][g])
                e.step()
                for g, h in enumerate(e.get_handles()):
                    e.get_reward_device(h, rews[0][g])
                e.clear_dead()
        for e in envs:
            e.sync()
        dt = time.perf_counter() - t0
        out[label] = {"agent_steps_per_s": total / dt, "ms_per_cycle": dt / steps * 1e3, "envs": K,
                      "envs_in_batched_pipeline": sum(1 for e in envs if e.pipeline_stats()[6] > 0)}     # (pipe.hip: worlds beyond the one-launch step, one launch per phase for all)
        del batch, envs
    return out


def many_worlds_extra(torch, magent_amd, dev, game, map_size, n, K, steps=20, warmup=5):
    import numpy as np
    """K environments of a BASELINE configuration on ONE GPU, cycled by one magent_amd.EnvBatch call per round (env_cycle_many): worlds
    beyond the one-launch step share one chain of launches, one per phase (pipe.hip) -- ms per round, aggregate agent-steps/s"""
    from magent_amd.builtin.config import _games
    envs = []
    for k in range(K):
        env = magent_amd.GridWorld(_games.make(game, map_size))
        env.set_seed(1000 + k); env.reset()
        hs = env.get_handles()
        if game == "gather":          # examples/train_gather.py: food, then agents; only the agents act and are observed
            env.add_agents(hs[0], "random", n=n // 5)
            env.add_agents(hs[1], "random", n=n)
        else:
            for h in hs:
                env.add_agents(h, "random", n=n)
        envs.append(env)
    hs = envs[0].get_handles()
    acting = [1] if game == "gather" else list(range(len(hs)))
    cap = [envs[0].get_num(h) for h in hs]
    mk = lambda shape, dtype=torch.float32: torch.empty(shape, dtype=dtype, device=dev)
    views = [[mk((cap[g],) + envs[0].get_view_space(h)) if g in acting else None for g, h in enumerate(hs)] for _ in envs]
    feats = [[mk((cap[g],) + envs[0].get_feature_space(h)) if g in acting else None for g, h in enumerate(hs)] for _ in envs]
    rews = [[mk(cap[g]) if g in acting else None for g in range(len(hs))] for _ in envs]
    na = [envs[0].get_action_space(h)[0] for h in hs]
    acts = [[[torch.randint(na[g], (cap[g],), dtype=torch.int32, device=dev) if g in acting else None for g in range(len(hs))] for _ in envs] for _ in range(4)]
    batch = magent_amd.EnvBatch(envs, n_threads=8)
    batch.order_streams = False
    view_p, feat_p, rew_p = batch.pointers(views), batch.pointers(feats), batch.pointers(rews)
    act_p = [batch.pointers(a) for a in acts]
    torch.cuda.synchronize()
    total, t0 = 0, 0.0
    for s in range(steps + warmup):
        if s == warmup:
            for e in envs:
                e.sync()
            total, t0 = 0, time.perf_counter()
        total += int(batch.nums_array().sum() if len(acting) == len(hs) else batch.nums_array()[:, acting].sum())
        batch.cycle(view_p, feat_p, act_p[s % 4], rew_p)
    for e in envs:
        e.sync()
    dt = time.perf_counter() - t0
    out = {"envs": K, "game": game, "map": map_size, "agents_per_env_at_start": cap, "agents_per_env_at_end": batch.nums()[0], "ms_per_round": dt / steps * 1e3,
           "agent_steps_per_s": total / dt, "envs_in_batched_pipeline": sum(1 for e in envs if e.pipeline_stats()[6] > 0),
           "observation_bytes_per_round_at_start": K * sum(cap[g] * 4 * (int(np.prod(envs[0].get_view_space(hs[g]))) + envs[0].get_feature_space(hs[g])[0]) for g in acting)}
    for e in envs:
        e.close()
    return out


def selfplay_extra(torch, magent_amd, n=400000, steps=4, infer_dtype="bf16"):
    """BASELINE config 5's shape on one GPU: battle 1000x1000, both sides acting through the reference's DQN (inference only,
    epsilon-greedy), observations and actions staying in HBM; the forward pass runs on the hand-written MFMA kernels
    (magent_amd/csrc/policy.hip).  Whole-cycle time: observe, infer, set_action per group; step; rewards (device-resident); clear_dead."""
    from magent_amd.builtin.torch_model import DeepQNetwork
    env = magent_amd.GridWorld("battle", map_size=MAP_SIZE, device_obs=True)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=n)
    models = [DeepQNetwork(env, h, "m%d" % i, memory_size=16, infer_batch_size=65536, infer_dtype=infer_dtype) for i, h in enumerate(hs)]
    cells = infer_dtype == "bf16" and all(m._hip is not None for m in models)
    hip_f32 = infer_dtype == "f32" and all(type(m._hip).__name__ == "HipDqnPolicyF32" for m in models)
    env.use_bf16_observations(cells)       # the bf16 MFMA kernels take the views as bf16 cells (2.7 KB per agent instead of 4.7)
    total, t0, rew = 0, 0.0, [None] * len(hs)
    for s in range(steps + 2):
        if s == 2:
            torch.cuda.synchronize(); env.sync()
            total, t0 = 0, time.perf_counter()
        acts = []
        for h, m in zip(hs, models):        # (examples/train_battle.py:63-71: observe + infer for every side, then the actions are set;
            obs = env.get_observation(h)    #  the second side's observation is rendered while the first side's policy runs)
            acts.append(m.infer_action(obs, None, policy="e_greedy", eps=0.1))
            total += env.get_num(h)
        for h, a in zip(hs, acts):
            env.set_action(h, a)
        env.step()
        for i, h in enumerate(hs):          # (the rewards stay in HBM like everything else of this loop: env_get_reward_device)
            k = env.get_num(h)
            if rew[i] is None or rew[i].shape[0] < k:
                rew[i] = torch.empty(k, dtype=torch.float32, device=torch.device("cuda", env.device_id))
            env.get_reward_device(h, out=rew[i][:k])
        env.clear_dead()
    torch.cuda.synchronize(); env.sync()
    dt = time.perf_counter() - t0
    out = {"agent_steps_per_s": total / dt, "ms_per_step": dt / steps * 1e3, "agents": [n, n],
           "policy_dtype": "bf16 (inputs, weights, activations; f32 accumulation)" if cells else "f32",
           "policy": "DQN forward pass, bf16 MFMA kernels on bf16-cell observations" if cells else
                     "DQN forward pass, float32 on v_mfma_f32_32x32x2_f32 (policy_f32.hip: the reference's arithmetic)" if hip_f32 else
                     "DQN forward pass, PyTorch float32 (the reference's arithmetic)"}
    env.close()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: re-run this command line as N ranks, one per GPU, the way the driver's
    own launcher would (torch.distributed.run, rendezvous on 127.0.0.1); rank 0's JSON line passes through on stdout.
    The line carries the launcher's return code ("launcher_rc": a rank that died behind the headline is visible in the record).  Returns 0
    once that line has been seen -- the headline is measured before anything that can fail on one rank is tried, and a rank that dies later
    (the launcher then ends the others) costs the extra it died in, not the run (see `relay_sigterm`) -- else the launcher's code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    import threading
    held, lock, done = [], threading.Lock(), {"printed": False}

    def emit(rc):
        """rank 0's line, once: with the launcher's return code when the launcher has ended (a rank that died behind the
        headline is visible to whoever reads the line), with `null` when it has not ended two minutes after the line was seen -- a rank that
        hangs behind the headline must not take the measured line with it"""
        with lock:
            if done["printed"]:
                return
            done["printed"] = True
            for line in held:
                try:
                    rec = json.loads(line)
                    rec["launcher_rc"] = rc
                    line = json.dumps(rec) + "\n"
                except ValueError:
                    pass
                sys.stdout.write(line)
            sys.stdout.flush()
    timer = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            held.append(line)       # (rank 0 prints it last: held for the moment it takes the launcher to end, so that it can carry the launcher's code)
            if timer is None:
                timer = threading.Timer(120.0, emit, args=(None,))
                timer.daemon = True
                timer.start()
            continue
        sys.stderr.write(line)      # (anything else a rank prints: not the line)
        sys.stderr.flush()
    rc = proc.wait()
    if timer is not None:
        timer.cancel()
    emit(rc)
    return 0 if len(held) == 1 else (rc or 1)


def relay_sigterm(on_term):
    """SIGTERM -> `on_term()` on a helper thread, whatever the main thread is blocked in.  torch.distributed.run ends every rank with
    SIGTERM as soon as one rank has died; a Python-level handler only runs when the main thread comes back to the interpreter, which it
    does not while it waits inside a collective.  The C-level handler that `signal.set_wakeup_fd` installs writes the signal's number into
    a pipe at once; the thread reading that pipe acts."""
    import signal
    import threading
    r, w = os.pipe()
    os.set_blocking(w, False)
    signal.signal(signal.SIGTERM, lambda *a: None)         # (a Python handler must exist, else the default action -- die at once -- stays)
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)

    def watch():
        while True:
            b = os.read(r, 1)
            if b and b[0] == signal.SIGTERM:
                on_term()
    t = threading.Thread(target=watch, daemon=True)
    t.start()


def host_abi_extra(magent_amd, n=N_PER_GROUP, steps=3, warm=2):
    """SURVEY.md 8d "report both kernel and API-level rates": the same cycle through the REFERENCE ABI -- env_get_observation
    into caller-owned numpy buffers, host action arrays, host reward buffers (3.9 GB device -> host per step, PCIe included).
    Never the headline value."""
    import numpy as np
    env = magent_amd.GridWorld("battle", map_size=MAP_SIZE)
    env.set_seed(12345); env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=n)
    rs = np.random.RandomState(0)
    tot = t_obs = 0.0
    agent_steps = obs_bytes = 0
    for step in range(steps + warm):
        acts = [rs.randint(21, size=env.get_num(h)).astype(np.int32) for h in hs]
        nn = sum(env.get_num(h) for h in hs)
        t0 = time.perf_counter()
        for h, a in zip(hs, acts):
            t1 = time.perf_counter()
            view, feat = env.get_observation(h)
            if step >= warm:
                t_obs += time.perf_counter() - t1
                obs_bytes += view.nbytes + feat.nbytes
            env.set_action(h, a)
        env.step()
        for h in hs:
            env.get_reward(h)
        env.clear_dead()
        if step >= warm:
            tot += time.perf_counter() - t0
            agent_steps += nn
    env.close()
    return {"agent_steps_per_s": agent_steps / tot, "ms_per_step": tot / steps * 1e3, "get_observation_GBps_to_numpy": obs_bytes / t_obs / 1e9,
            "agents": [n, n], "io": "host buffers (reference ABI: env_get_observation / env_set_action / env_get_reward), PCIe included"}


def train_battle_formation(map_size, gap=3):
    """examples/train_battle.py:15-40 (`generate_map`): two squares of agents on every other cell, side = 2 * int(sqrt(0.04 M^2)) cells,
    `gap` cells either side of the middle column; the script swaps leftID / rightID before it places, so the first round of a process puts
    handles[1] on the left and adds it first.  -> [(group, int32[n, 3] positions)] in placement order.  M = 1000: 2 x 40,000; M = 3536:
    2 x 499,849 (BASELINE config 5's "1M agents")."""
    import math
    import numpy as np
    side = int(math.sqrt(map_size * map_size * 0.04)) * 2
    ys = np.arange((map_size - side) // 2, (map_size - side) // 2 + side, 2)
    out = []
    for g, x0 in ((1, map_size // 2 - gap - side), (0, map_size // 2 + gap)):
        xs = np.arange(x0, x0 + side, 2)
        pos = np.zeros((len(xs) * len(ys), 3), dtype=np.int32)
        pos[:, 0], pos[:, 1] = np.repeat(xs, len(ys)), np.tile(ys, len(xs))      # x outer, y inner: the script's loop order
        out.append((g, pos))
    return out


def train_round_extra(torch, magent_amd, map_size=1000, steps=12, on_step=None, train=True, infer_dtype="bf16"):
    """BASELINE config 5's loop on one GPU: examples/train_battle.py:61-140 (`play_a_round`) at --map_size 1000 -- its
    generate_map puts (int(sqrt(0.04 M^2)))^2 = 40,000 agents on each side -- with the observations staying in HBM
    (device_obs): observe -> infer_action (e-greedy, DQN) -> set_action per side; step; get_reward / get_alive -> sample_step;
    clear_dead; after `steps` steps one train() per model (replay sampling, double-DQN targets, Adam).  Times the three parts
    with the device synchronised around them.  on_step(step, env, handles, obs, acts, rewards, alives) lets a test check the
    engine's outputs against its checker."""
    from magent_amd.builtin.torch_model import DeepQNetwork
    from magent_amd.model import ProcessingModel
    env = magent_amd.GridWorld("battle", map_size=map_size, device_obs=True)
    env.set_seed(12345); env.reset()
    handles = env.get_handles()
    for g, pos in train_battle_formation(map_size):
        env.add_agents(handles[g], method="custom", pos=pos)
    n0 = [env.get_num(h) for h in handles]
    models = [ProcessingModel(env, h, "c5_%d" % i, 20000 + i, 1000, DeepQNetwork, batch_size=512, memory_size=2 ** 17,
                              target_update=1000, train_freq=4, infer_dtype=infer_dtype) for i, h in enumerate(handles)]

    def synced():
        torch.cuda.synchronize(); env.sync()
        return time.perf_counter()
    T = {"env": 0.0, "infer": 0.0, "sample": 0.0, "agent_steps": 0}

    def play(first_step, n_steps):
        for step in range(first_step, first_step + n_steps):
            obs, acts = [None] * len(handles), [None] * len(handles)
            for i, h in enumerate(handles):
                t0 = synced()
                obs[i] = env.get_observation(h)
                ids = env.get_agent_id(h)
                t1 = synced()
                models[i].infer_action(obs[i], ids, "e_greedy", 0.5, block=False)
                t2 = synced()
                T["env"] += t1 - t0; T["infer"] += t2 - t1
            t0 = synced()
            for i, h in enumerate(handles):
                acts[i] = models[i].fetch_action()
                env.set_action(h, acts[i])
                T["agent_steps"] += env.get_num(h)
            env.step()
            rewards = [env.get_reward(h) for h in handles]
            alives = [env.get_alive(h) for h in handles]
            t1 = synced()
            if train:
                for i in range(len(handles)):
                    models[i].sample_step(rewards[i], alives[i], block=False)
            t2 = synced()
            if on_step is not None:
                on_step(step, env, handles, obs, acts, rewards, alives)
            t3 = synced()
            env.clear_dead()
            t4 = synced()
            T["env"] += (t1 - t0) + (t4 - t3); T["sample"] += t2 - t1

    def train_round():
        t0 = synced()
        for m in models:
            m.train(print_every=10 ** 9, block=False)
        res = [m.fetch_train() for m in models]
        return (synced() - t0) * 1e3, res
    # untimed: the first inference of a process loads kernels and sizes workspaces, and the first two steps allocate both sets of cached
    # observation tensors and the episode buffer's blocks (gigabytes of hipMalloc at 2 x 500k agents: 20 ms per step if it is timed)
    for i, h in enumerate(handles):
        models[i].infer_action(env.get_observation(h), env.get_agent_id(h), "e_greedy", 0.5, block=False)
    play(-2, 2)
    for k in T:
        T[k] = 0 if k == "agent_steps" else 0.0
    play(0, steps)
    t_env, t_infer, t_sample, agent_steps = T["env"], T["infer"], T["sample"], T["agent_steps"]
    out = {"map_size": map_size, "agents": n0, "steps": steps, "env_ms_per_step": t_env / steps * 1e3, "infer_ms_per_step": t_infer / steps * 1e3,
           "sample_ms_per_step": t_sample / steps * 1e3,
           "agent_steps_per_s_sampling": agent_steps / (t_env + t_infer + t_sample),
           "policy_dtype": "bf16 inference (MFMA kernels: inputs, weights, activations bf16, f32 accumulation), f32 training" if infer_dtype == "bf16" and all(m.model._hip is not None for m in models)
                           else "f32 inference (policy_f32.hip: float32 in, float32 accumulate, v_mfma_f32_32x32x2_f32), f32 training" if all(type(m.model._hip).__name__ == "HipDqnPolicyF32" for m in models)
                           else "f32 inference (PyTorch), f32 training",
           "policy": "DQN (2 x conv3x3(32) -> dense 256 || dense 256 -> dueling head)"}
    if infer_dtype == "f32":
        # the forward pass against the f32 matrix peak (MI355X_MICROARCH.md: 157.3 TFLOP/s, v_mfma_f32_32x32x2_f32): useful FLOPs of the
        # network per agent (valid convolutions 13x13 -> 11x11 -> 9x9 for the battle shape; generic below) x agents inferred / time
        vh, vw, vc = env.get_view_space(handles[0])
        nf, na = env.get_feature_space(handles[0])[0], env.get_action_space(handles[0])[0]
        flop = 2.0 * ((vh - 2) * (vw - 2) * 32 * 9 * vc + (vh - 4) * (vw - 4) * 32 * 288 + (vh - 4) * (vw - 4) * 32 * 256 + nf * 256 + 512 * (na + 1))
        out["roofline"] = {"bound": "mfma_f32", "achieved": flop * agent_steps / t_infer / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                           "frac": flop * agent_steps / t_infer / 157.3e12, "flop_per_agent": flop, "kernels": "k_dqn_conv_f32 + k_dqn_head_f32",
                           "note": "useful FLOPs of the network x agents inferred / synchronised wall time of infer_action (e-greedy draw included)"}
    if train:
        # the first train() of a process also pays for MIOpen's choice of convolution kernels (seconds); a second round -- the
        # same number of steps played again -- is what a training run pays per round
        out["train_ms_first_round"], _ = train_round()
        play(steps, steps)
        out["train_ms_per_round"], res = train_round()
        out["loss"] = [float(r[0]) for r in res]
        out["value"] = [float(r[1]) for r in res]
    env.close()
    return out


def is_default_workload(args):
    return (args.workload, args.map_size, args.agents) == ("battle", MAP_SIZE, N_PER_GROUP)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--map-size", type=int, default=MAP_SIZE)
    ap.add_argument("--agents", type=int, default=N_PER_GROUP, help="agents per group")
    ap.add_argument("--workload", choices=["battle", "battle_fill", "test_1m", "gather", "battle_c5", "battle_c5_melee"], default="battle",
                    help="battle_fill: SURVEY.md 8d C3(ii), the map filled to capacity by add_agents('fill') (2 x 498,002 at 1000 x 1000; nobody can "
                         "move); test_1m: the reference's own harness (scripts/test/test_1m.py): pursuit-like game, map sqrt(20 N), "
                         "N/10 walls, N/2 prey + N/2 2x2 predators, N = 2 * --agents; "
                         "gather: BASELINE config 4 (examples/train_gather.py: --agents agents + agents/5 food, only agents act); "
                         "battle_c5: BASELINE config 5's world -- examples/train_battle.py's own generate_map at --map-size (3536: 2 x 499,849 "
                         "agents on 12.5 M cells), random actions; battle_c5_melee: the same two lattices interleaved (every agent has hostile "
                         "neighbours at distance 1: the state a self-play episode reaches once the fronts have met)")
    ap.add_argument("--gather", choices=["none", "obs", "obs-padded"], default="none",
                    help="obs: exchange the observation tensors of every replica over RCCL each step (counts first, then sends / receives "
                         "sized by count, on a side stream under the step); obs-padded: one all_gather_into_tensor of capacity rows")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to dry-run the N > 1 "
                         "code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--obs", choices=["f32", "bf16"], default="f32",
                    help="bf16: render the observations as the policy kernels' bf16 cells (env_get_observation_device_bf16, 8 x bf16 per window "
                         "cell) -- a secondary reading; the headline stays on the reference's float32 tensors")
    ap.add_argument("--extra-timeout", type=int, default=180, help="N > 1: seconds the config-4 gather extra may take before the line is printed without it")
    ap.add_argument("--repeats", type=int, default=5, help="identical timed regions of --steps steps; the median one is reported")
    ap.add_argument("--event-every", type=int, default=EVENT_EVERY, help="timed region: HIP events around the render launches of every N-th step (1: all)")
    ap.add_argument("--preheat-ms", type=float, default=PREHEAT_MS,
                    help="milliseconds of untimed render launches before the warm-up steps of every region: the device's sustained state (see measure); 0: none")
    ap.add_argument("--no-cold", action="store_true", help="skip the three extra regions without the preheat (`no_preheat` in the line)")
    ap.add_argument("--force-extra", action="store_true", help="test hook: run the N > 1 extra behind a non-default headline workload too")
    ap.add_argument("--fault", default="", help="test hook (tests/test_bench_multi.py): 'kill-rank-1-in-extra' makes rank 1 die inside the N > 1 extra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernels with HIP events")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (capacity fill, test_1m, small worlds)")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    ap.add_argument("--cpu-lib", default="")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--check-gather", action="store_true",
                    help="with --gather: after the timed region every rank digests its own rendered rows and the shards it received; "
                         "rank 0 prints which shards were bit-identical to the peer's own tensor (config.gather_detail.verified)")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    # stdout carries rank 0's ONE JSON line and nothing else: whatever the legs below print on the way (the reference-style training loops of the
    # config 5 legs report every batch) goes to stderr
    real_stdout, sys.stdout = sys.stdout, sys.stderr

    def emit(record):
        real_stdout.write(json.dumps(record) + "\n")
        real_stdout.flush()

    import torch
    import torch.distributed as dist
    import magent_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    assert args.gpus in (1, world), "--gpus %d under a launcher with WORLD_SIZE=%d" % (args.gpus, world)
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()   # dry run: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The replicas share nothing on the data path, so the job's control plane -- the barriers around the timed region, the MAX / SUM of the
    # ranks' times and counts, the digests of --check-gather -- runs over gloo on CPU tensors whatever --backend says: the headline then
    # does not depend on a collective library that this code has never met on more than one GPU.  --backend names the DATA plane: the
    # exchange of the observation shards (replicas.ObservationGather), over a group of its own -- RCCL over xGMI by default -- that is
    # created where the first gather is asked for (inside the watchdog of extra.c4_gather_rccl on the default run).
    red_dev = torch.device("cpu")
    data_group = {"g": None}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    from magent_amd import replicas

    def gather_group():
        if args.backend == "nccl" and data_group["g"] is None:
            data_group["g"] = dist.new_group(backend="nccl")      # (every rank passes here together: measure() is called in lockstep)
        return data_group["g"]

    def measure(workload, map_size, agents, steps, warmup, profile, gather="none", seed=12345 + rank, check_gather=False, preheat_ms=None):
        """K timed steps of `workload`; returns the fields of the bench line that depend on the run"""
        from magent_amd.builtin.config import _games
        if workload == "test_1m":
            map_size = int((2 * agents * 20) ** 0.5)
            cfg = _games.make("pursuit", map_size)
        elif workload == "gather":
            cfg = _games.make("gather", map_size)
        else:
            cfg = _games.make("battle", map_size)
        cfg.set({"device_id": local_rank})
        env = magent_amd.GridWorld(cfg)
        env.set_seed(seed)
        env.reset()
        handles = env.get_handles()
        acting = list(range(len(handles)))
        if workload == "test_1m":
            env.add_walls(method="random", n=2 * agents // 10)
            for h in reversed(handles):
                env.add_agents(h, "random", n=agents)
        elif workload == "gather":   # group 0 = food (never observed, never acts), group 1 = agents
            env.add_agents(handles[0], "random", n=agents // 5)
            env.add_agents(handles[1], "random", n=agents)
            acting = [1]
        elif workload in ("battle_c5", "battle_c5_melee"):   # examples/train_battle.py:15-40 at --map_size `map_size`
            form = train_battle_formation(map_size)
            if workload == "battle_c5_melee":               # the right square pushed into the left one: odd columns of the same square
                (g_l, left), (g_r, right) = form
                right = right.copy()
                right[:, 0] += left[0, 0] + 1 - right[0, 0]
                form = [(g_l, left), (g_r, right)]
            for g, pos in form:
                env.add_agents(handles[g], method="custom", pos=pos)
        elif workload == "battle_fill":   # the two halves of the inner map, every cell taken
            half = (map_size - 2) // 2
            env.add_agents(handles[0], "fill", pos=(1, 1), size=(half, map_size - 2))
            env.add_agents(handles[1], "fill", pos=(1 + half, 1), size=(map_size - 2 - half, map_size - 2))
        else:
            for h in handles:
                env.add_agents(h, "random", n=agents)
        n0 = [env.get_num(h) for h in handles]
        G = len(handles)
        vss = [env.get_view_space(h) for h in handles]
        fss = [env.get_feature_space(h) for h in handles]
        n_actions = [env.get_action_space(h)[0] for h in handles]
        bf16 = args.obs == "bf16"
        view_bytes = [(16 * v[0] * v[1]) if bf16 else (4 * v[0] * v[1] * v[2]) for v in vss]   # per agent: what k_render writes (the dominant kernel)
        feat_bytes = [4 * f[0] for f in fss]                 # per agent: the feature rows (they ride in the render launch)

        # caller-owned device buffers (the reference's ownership convention), sized once for the initial population
        # with the gather on, two view tensors per group used alternately: the render of step t+1 then only waits for the
        # exchange of step t-1 (the one that read the tensor it overwrites), not for the exchange of step t
        n_buf = 2 if (gather != "none" and world > 1) else 1
        views = [[torch.empty((n0[g],) + (vss[g][:2] + (8,) if bf16 else vss[g]), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
                  for _ in range(n_buf)] for g in range(G)]
        feats = [torch.empty((n0[g],) + fss[g], dtype=torch.float32, device=dev) for g in range(G)]
        rewards = [torch.empty(n0[g], dtype=torch.float32, device=dev) for g in range(G)]
        total_steps = steps + warmup
        gen = torch.Generator(device=dev)
        gen.manual_seed(rank)
        n_sets = min(total_steps + 8, 32)   # action sets are recycled: 32 x 3.2 MB is enough entropy
        actions = [[torch.randint(n_actions[g], (n0[g],), dtype=torch.int32, device=dev, generator=gen) for g in range(G)] for _ in range(n_sets)]
        gathers = None
        if gather != "none" and world > 1:   # the north star's batched-observation gather: every replica's view tensor to every rank
            gdev = dev if args.backend == "nccl" else torch.device("cpu")
            gathers = [replicas.ObservationGather(vss[g], capacity=n0[g], device=gdev, mode="padded" if gather == "obs-padded" else "exact", group=gather_group())
                       if g in acting else None for g in range(G)]
        staged = [torch.empty((n0[g],) + vss[g], dtype=torch.float32).pin_memory() if gathers and args.backend != "nccl" and g in acting else None
                  for g in range(G)]
        last_sent = {}
        torch.cuda.synchronize()
        rendered = {"view": 0, "feat": 0, "launches": 0, "agents": []}
        step_ends = []

        sampling = {"on": False}

        def one_step(s):
            n_now = 0
            # inside the timed region the render launches of every EVENT_EVERY-th step carry a HIP event pair (a pair drains the stream: ~5 us
            # each, measured 9-10 us per step with every launch timed); `rendered` counts the bytes of exactly those launches
            timed = (not sampling["on"]) or (s - warmup) % max(1, args.event_every) == 0
            if sampling["on"]:
                env.profile_enable(2 if timed else 0)
            for g, h in enumerate(handles):
                if g not in acting:
                    continue
                n = env.get_num(h)
                n_now += n
                if timed:
                    rendered["view"] += n * view_bytes[g]
                    rendered["feat"] += n * feat_bytes[g]
                    rendered["launches"] += 1
                    rendered["agents"].append(n)
                view = views[g][s % n_buf]
                if gathers and args.backend == "nccl":   # the exchange of step t-1 may still read the tensor this render overwrites
                    gathers[g].release(view, env.stream)
                if bf16:
                    env.get_observation_device_bf16(h, view, feats[g])
                else:
                    env.get_observation_device(h, view, feats[g])
                env.set_action_device(h, actions[s % n_sets][g])
                if gathers:                      # counts now (behind the render, on the side stream); the rows follow below
                    if args.backend == "nccl":
                        gathers[g].launch(view, n, producer_stream=env.stream)
                    else:                        # gloo dry run: the rows are staged through host memory
                        env.sync()
                        staged[g][:n].copy_(view[:n])
                        gathers[g].launch(staged[g], n)
                    last_sent[g] = (view, n)
            if gathers:                          # the rows travel on the side stream while the step kernels run on the engine's
                for g in acting:
                    gathers[g].post()
            env.step()
            for g, h in enumerate(handles):
                if g in acting:
                    env.get_reward_device(h, rewards[g])
            env.clear_dead()
            step_ends.append(time.perf_counter())
            return n_now

        # Before the W warm-up steps: the device itself is brought to its sustained state.  Every region starts behind the host-side
        # construction of a fresh world (the device idles for half a second), and the first ~10 ms of GPU work after that run in a transient of
        # the device's power management: by kernel timestamps the same render launch takes 0.32 -> 0.38 -> 0.32 ms over the first dozen launches,
        # whatever the actions, and settles at 0.29-0.30 (profiles/r04_summary.md, "The first milliseconds").  So the observation of step 0 is
        # rendered again and again for --preheat-ms of wall time -- the episode does not advance, nothing of it is timed -- and the W warm-up
        # steps and the K timed steps then see the device as a long-running job sees it.
        preheat_launches = 0
        if preheat_ms is None:
            preheat_ms = args.preheat_ms
        if preheat_ms > 0 and acting and env.get_num(handles[acting[0]]) > 0:
            g0 = acting[0]
            t_heat = time.perf_counter()
            while (time.perf_counter() - t_heat) * 1e3 < preheat_ms:
                for _ in range(16):
                    if bf16:
                        env.get_observation_device_bf16(handles[g0], views[g0][0], feats[g0])
                    else:
                        env.get_observation_device(handles[g0], views[g0][0], feats[g0])
                env.sync()
                preheat_launches += 16
        for s in range(warmup):
            one_step(s)
        env.sync()
        if profile:
            # inside the timed region only the dominant kernel carries HIP events (an event pair costs ~10 us of stream time;
            # timing every phase of every step would add ~10 % to the step); the phase breakdown is taken afterwards
            env.profile_enable(2)
            sampling["on"] = True
            for name in ("render", "features", "paint", "minimap", "attack", "move", "turn", "set_action", "step", "rules", "clear_dead"):
                env.profile_read(name)
        rendered["view"] = rendered["feat"] = rendered["launches"] = 0
        del rendered["agents"][:]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        del step_ends[:]
        t0 = time.perf_counter()
        agent_steps = 0
        for s in range(warmup, total_steps):
            agent_steps += one_step(s)
        if gathers:
            for g in acting:
                gathers[g].wait()
        env.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        per_step = sorted(b - a for a, b in zip([t0] + step_ends[:-1], step_ends))
        median_ms = per_step[len(per_step) // 2] * 1e3 if per_step else None

        res = {"elapsed": elapsed, "agent_steps": agent_steps, "median_ms": median_ms, "n0": n0, "agents_at_end": [env.get_num(h) for h in handles],
               "host_finished_steps": env.engine_stats()[0], "preheat_launches": preheat_launches, "attack_round_hist": list(env.round_hist()), "cycles_run": total_steps, "roofline": None, "breakdown": {}, "map_size": map_size}
        sampling["on"] = False
        if profile:
            n_launch, ms = env.profile_read("render")
            n_feat, ms_feat = env.profile_read("features")
            if n_launch and ms > 0:
                # algorithmic bytes of the dominant kernel: every element of the observation written exactly once
                # (B_obs = 4*(VH*VW*C + F) per agent; the feature rows ride in the render launch)
                fused = n_feat == 0
                obs_bytes = rendered["view"] + (rendered["feat"] if fused else 0)
                achieved = obs_bytes / (ms * 1e-3) / 1e9
                traffic, traffic_note = None, None
                pmc = os.path.join(ROOT, "profiles", "render_pmc.json")
                pmc_key = {("battle", 1000): "battle_1000", ("battle_c5", 3536): "battle_c5_3536", ("test_1m", 4472): "test_1m",
                           ("gather", 500): "gather_500"}.get((workload, map_size))
                if os.path.exists(pmc) and pmc_key and not bf16:     # (the PMC passes ran these workloads: tools/measure.sh <tag> bytes)
                    try:     # PMC bytes per rendered AGENT (separate rocprofv3 --pmc passes, profiles/), scaled to this run's launches
                        rec = json.load(open(pmc))["workloads"][pmc_key]
                        agents_per_launch = sum(rendered["agents"]) / float(n_launch)
                        traffic = rec["hbm_bytes_per_agent"] * agents_per_launch
                        traffic_note = "PMC FETCH_SIZE + WRITE_SIZE per rendered agent (%s) x %.0f agents per launch of this run" % (rec.get("source", "profiles/"), agents_per_launch)
                    except Exception:
                        traffic = None
                kname = {0: "k_render_cells16" if bf16 else "k_render", 1: "k_render_fast", 4: "k_render_sweep2"}.get(env.engine_stats()[6], "k_render")
                res["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_scaled_from_profiles_not_this_run": traffic_note,
                                   "launches": n_launch, "launches_note": "HIP events around the render launches of every %d-th step of the timed region" % max(1, args.event_every),
                                   "avg_launch_ms": round(ms / n_launch, 4),
                                   "algorithmic_bytes_per_launch": int(obs_bytes / n_launch)}
            # phase breakdown: a few more steps of the same episode, OUTSIDE the timed region, with an event pair around every phase
            extra = 5
            env.profile_enable(1)
            before = (rendered["view"], rendered["feat"])
            for s in range(total_steps, total_steps + extra):
                one_step(s)
            env.sync()
            for name in ("paint", "minimap", "attack", "move", "turn", "set_action", "step", "rules", "clear_dead"):
                k, t_ms = env.profile_read(name)
                if k:
                    res["breakdown"][name + "_ms_per_step"] = round(t_ms / extra, 4)
            res["breakdown"]["note"] = "%d extra steps after the timed region" % extra
            n2, ms2 = env.profile_read("render")
            n2f, _ = env.profile_read("features")
            if res["roofline"]:
                res["breakdown"]["render_ms_per_step"] = round(ms / n_launch * len(acting), 4)
                if n2 and ms2 > 0:
                    # the same kernel in the breakdown steps, where EVERY launch of every phase sits between event pairs: each launch starts
                    # behind a drained stream (the method of the rounds before; ~2 % faster than in the loop's steady state, profiles/r04_summary.md)
                    b2 = (rendered["view"] - before[0]) + ((rendered["feat"] - before[1]) if n2f == 0 else 0)
                    res["roofline"]["behind_a_drained_stream"] = {"launches": n2, "avg_launch_ms": round(ms2 / n2, 4),
                                                                  "achieved": round(b2 / (ms2 * 1e-3) / 1e9, 1), "frac": round(b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                                  "note": "the %d breakdown steps after the timed region: every launch between event pairs" % extra}
            env.profile_enable(False)
            if res["roofline"] and not gathers:
                # ... and the kernel ALONE: back-to-back launches of the first acting group's render, nothing in between, by HIP events
                # (inside the cycle the same launch is 5-10 % slower by plain kernel timestamps too: profiles/r04_summary.md)
                g0 = acting[0]
                n_alone = env.get_num(handles[g0])
                if n_alone > 0:
                    env.profile_enable(2); env.profile_read("render"); env.profile_read("features")
                    for _ in range(30):
                        if bf16:
                            env.get_observation_device_bf16(handles[g0], views[g0][0], feats[g0])
                        else:
                            env.get_observation_device(handles[g0], views[g0][0], feats[g0])
                    env.sync()
                    n3, ms3 = env.profile_read("render")
                    n3f, _ = env.profile_read("features")
                    env.profile_enable(False)
                    if n3 and ms3 > 0:
                        b3 = n3 * n_alone * (view_bytes[g0] + (feat_bytes[g0] if n3f == 0 else 0))
                        res["roofline"]["kernel_alone"] = {"launches": n3, "agents_per_launch": n_alone, "avg_launch_ms": round(ms3 / n3, 4),
                                                           "achieved": round(b3 / (ms3 * 1e-3) / 1e9, 1), "frac": round(b3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                           "note": "30 back-to-back launches of the same render after the run, nothing in between"}
        if gathers:
            res["gather"] = {"mode": gathers[acting[0]].mode, "payload_bytes_sent_per_step": sum(gathers[g].bytes_sent for g in acting),
                             "view_buffers": n_buf}
            # the payload exchange alone (events on the side stream; gloo: host time), on three more steps of the episode
            ex = []
            base = total_steps + (5 if profile else 0)
            for s in range(base, base + 3):
                one_step(s)
                for g in acting:
                    gathers[g].wait()
                ex.append(sum(gathers[g].exchange_ms() for g in acting))
            env.sync()
            res["gather"]["exchange_ms"] = sorted(ex)[1]
            res["gather"]["rows_sent_to_each_peer"] = sum(gathers[g]._n for g in acting)
            res["gather"]["bytes_to_each_peer"] = sum(gathers[g]._n * gathers[g].row_bytes for g in acting)
            if check_gather:
                # every rank digests the rows it rendered in the last exchanged step and the shards it received; the lists meet on
                # every rank: shard r as received anywhere must be the bytes rank r rendered (HIP engine output, bit for bit)
                import hashlib

                def dig(t):
                    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
                mine = {}
                for g in acting:
                    shards = gathers[g].wait()
                    torch.cuda.synchronize()
                    view, n = last_sent[g]
                    mine[g] = {"own": dig(view[:n]), "n": n, "got": [dig(t) for t in shards], "got_n": [int(t.shape[0]) for t in shards]}
                everyone = [None] * world
                dist.all_gather_object(everyone, mine)
                ok, pairs = True, 0
                for g in acting:
                    for r in range(world):
                        for q in range(world):
                            pairs += 1
                            ok = ok and everyone[q][g]["got"][r] == everyone[r][g]["own"] and everyone[q][g]["got_n"][r] == everyone[r][g]["n"]
                distinct = len({everyone[r][acting[0]]["own"] for r in range(world)})
                res["gather"]["verified"] = {"all_shards_bit_identical": bool(ok), "pairs_checked": pairs, "distinct_replicas": distinct}
        del env
        return res

    # The headline: W untimed warm-up steps, then EXACTLY K timed steps between barriers + synchronisations -- done `--repeats` times
    # on identically built worlds (same seed, same actions: the K-step window is the same work every time), and the repeat with
    # the MEDIAN time is the one reported (a 20-step region is ~17 ms: one region moves +-2 % on nothing).  Every repeat's
    # ms/step is in the line (`repeats_ms_per_step`).
    runs = []
    for rep in range(max(1, args.repeats)):
        R = measure(args.workload, args.map_size, args.agents, args.steps, args.warmup, not args.no_profile, gather=args.gather,
                    check_gather=args.check_gather and rep == 0)
        e, a = R["elapsed"], R["agent_steps"]
        if world > 1:   # whole-job aggregate over the slowest replica's time
            e = replicas.max_over_replicas(e, device=red_dev)
            a = replicas.sum_over_replicas(a, device=red_dev)
        runs.append((e, a, R))
    order = sorted(range(len(runs)), key=lambda k: runs[k][0])
    elapsed, agent_steps, R = runs[order[len(order) // 2]]
    if args.check_gather and "gather" in runs[0][2] and "verified" in runs[0][2]["gather"]:
        R.setdefault("gather", {})["verified"] = runs[0][2]["gather"]["verified"]
    repeats_ms = [round(r[0] / args.steps * 1e3, 4) for r in runs]

    # ... and the same region WITHOUT the preheat, beside it (ADVICE / VERDICT round 4): what a caller sees whose device was idle a moment
    # ago -- the figure every line of rounds 1-3 carried.  Three regions, the median one; N = 1 only (no barriers involved).
    cold = None
    if world == 1 and args.preheat_ms > 0 and not args.no_cold:
        cr = [measure(args.workload, args.map_size, args.agents, args.steps, args.warmup, not args.no_profile, preheat_ms=0.0) for _ in range(3)]
        cr.sort(key=lambda r: r["elapsed"])
        c = cr[1]
        cold = {"ms_per_step": c["elapsed"] / args.steps * 1e3, "value": c["agent_steps"] / c["elapsed"], "regions_ms_per_step": [round(r["elapsed"] / args.steps * 1e3, 4) for r in cr],
                "render_frac_in_region": (c["roofline"] or {}).get("frac"), "render_avg_launch_ms": (c["roofline"] or {}).get("avg_launch_ms"),
                "note": "the same %d-step region without the preheat renders (--preheat-ms 0): the device's first milliseconds after an idle period; "
                        "VERDICT / BASELINE comparisons use the sustained figure (`value`), lines of rounds 1-3 carried this one" % args.steps}
    if rank == 0:
        is_default = is_default_workload(args)
        names = {"test_1m": "reference test_1m.py harness: pursuit-like %dx%d, %d walls, %d prey + %d 2x2 predators" % (
                     R["map_size"], R["map_size"], 2 * args.agents // 10, args.agents, args.agents),
                 "gather": "gather %dx%d (train_gather.py), %d agents + %d food, only the agents act" % (args.map_size, args.map_size, args.agents, args.agents // 5),
                 "battle_fill": "battle %dx%d filled to capacity by add_agents('fill'): %s agents (SURVEY.md 8d C3(ii)), random actions" % (args.map_size, args.map_size, R["n0"]),
                 "battle_c5": "battle %dx%d, examples/train_battle.py's generate_map formation: %s agents, random actions" % (args.map_size, args.map_size, R["n0"]),
                 "battle_c5_melee": "battle %dx%d, the two lattices of train_battle.py's formation interleaved: %s agents, random actions" % (args.map_size, args.map_size, R["n0"]),
                 "battle": "battle %dx%d, 2x%d agents, random placement, random actions" % (args.map_size, args.map_size, args.agents)}
        rec = {
            "metric": "agent-steps/sec (step+obs) on battle map; bit-exact vs CPU ref",
            "value": agent_steps / elapsed,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": R["median_ms"],
            "repeats": len(runs), "repeats_ms_per_step": repeats_ms,
            "value_is": "the repeat with the median time of %d identical %d-step regions, each behind %.0f ms of untimed renders (the device's sustained state)" % (len(runs), args.steps, args.preheat_ms),
            "ms_per_step_no_preheat": cold["ms_per_step"] if cold else None, "no_preheat": cold,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOAD if is_default else names[args.workload],
                       "envs": world, "parallelism": "replicas x%d" % world, "gather": args.gather,
                       "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                       "backend": (("control plane gloo; observation gather: " + ("RCCL (nccl)" if args.backend == "nccl" else "gloo")) if world > 1 else None),
                       "agents_at_start": R["n0"], "agents_at_end": R["agents_at_end"],
                       "io": "device-resident (env_*_device C-ABI)", "tune": os.environ.get("MAGENT_TUNE") or None, "steps_finished_by_host_driver": R["host_finished_steps"],
                       "host_driver_rate": R["host_finished_steps"] / float(R["cycles_run"]),
                       # steps of the run by the last round of the death-rank fixed point that still changed something (index; one more round confirms)
                       "attack_round_hist": R["attack_round_hist"],
                       # untimed, before the W warm-up steps of every region: step 0's observation rendered over and over for --preheat-ms, so that
                       # the region sees the device's sustained state and not the power-management transient of its first milliseconds of work
                       "preheat": {"ms": args.preheat_ms, "render_launches": R["preheat_launches"]}},
            "roofline": R["roofline"],
            "breakdown": R["breakdown"],
        }
        if "gather" in R:
            rec["config"]["gather_detail"] = R["gather"]
        if world == 1 and not args.no_cpu_baseline:
            small = args.map_size * args.map_size <= 250000
            rec["cpu_baseline"] = run_cpu_baseline(args.map_size, args.agents, steps=200 if small else 20) \
                if args.workload == "battle" else None
        else:
            rec["cpu_baseline"] = None
        if world == 1 and is_default and not args.no_extras:
            # the other readings of "battle 1000x1000 / 1M agents" the survey sanctions, and BASELINE config 2, each a short run
            extra = {}
            try:
                F = measure("battle_fill", MAP_SIZE, 0, 10, 3, False)
                extra["battle_fill_2x498002"] = {"agent_steps_per_s": F["agent_steps"] / F["elapsed"], "ms_per_step": F["elapsed"] / 10 * 1e3, "agents": F["n0"]}
                # every other BASELINE configuration with the roofline of ITS render kernel (algorithmic bytes 4 * (VH*VW*C + F) per
                # rendered agent: test_1m 5 channels without minimap, gather 15 x 15 x 7 = 6444 B, C2 the battle shape at 2 x 2000)
                T = measure("test_1m", 0, 500000, 24, 4, True)       # (24 steps: 12 render launches between event pairs)
                extra["test_1m_2x500k"] = {"agent_steps_per_s": T["agent_steps"] / T["elapsed"], "ms_per_step": T["elapsed"] / 24 * 1e3, "agents": T["n0"], "map": T["map_size"],
                                           "roofline": T["roofline"]}
                # BASELINE config 5's world measured like the headline: battle 3536 x 3536, 2 x 499,849 agents in
                # examples/train_battle.py's own formation (two squares 6 columns apart), and the same two lattices interleaved (the melee a
                # self-play episode reaches once the fronts have met); each leg with the roofline of ITS render launches, the phase breakdown,
                # the death-rank rounds it needed and the steps the host had to finish
                extra["c5_cycle_3536"] = {}
                for leg, wl in (("formation", "battle_c5"), ("melee", "battle_c5_melee")):
                    C5 = measure(wl, 3536, 0, 24, 4, True)
                    extra["c5_cycle_3536"][leg] = {"agent_steps_per_s": C5["agent_steps"] / C5["elapsed"], "ms_per_step": C5["elapsed"] / 24 * 1e3,
                                                   "agents": C5["n0"], "agents_at_end": C5["agents_at_end"], "roofline": C5["roofline"], "breakdown": C5["breakdown"],
                                                   "attack_round_hist": C5["attack_round_hist"], "steps_finished_by_host_driver": C5["host_finished_steps"]}
                C4 = measure("gather", 500, 100000, 20, 5, True)     # (the agents of this game starve within a few dozen steps: a longer region would time a smaller world)
                extra["gather_500_100k"] = {"agent_steps_per_s": C4["agent_steps"] / C4["elapsed"], "ms_per_step": C4["elapsed"] / 20 * 1e3, "agents": C4["n0"],
                                            "workload": "BASELINE config 4, one replica: gather 500x500 (train_gather.py), 100k agents + 20k food, only the agents act",
                                            "roofline": C4["roofline"], "breakdown": C4["breakdown"]}
                extra["gather_500_100k"]["eight_replicas_one_gpu"] = many_worlds_extra(torch, magent_amd, dev, "gather", 500, 100000, 8, steps=20)
                extra["battle_600_2x20000_8env"] = many_worlds_extra(torch, magent_amd, dev, "battle", 600, 20000, 8, steps=40)
                C2 = measure("battle", 200, 2000, 200, 20, True)
                extra["battle_200_2x2000"] = small_world_extras(torch, magent_amd, dev)
                extra["battle_200_2x2000"]["calls_with_events"] = {"ms_per_step": C2["elapsed"] / 200 * 1e3, "roofline": C2["roofline"], "breakdown": C2["breakdown"]}
                extra["battle_selfplay_2x400k"] = selfplay_extra(torch, magent_amd)
                extra["battle_selfplay_2x400k_f32_policy"] = selfplay_extra(torch, magent_amd, steps=2, infer_dtype="f32")
                extra["host_abi_2x400k"] = host_abi_extra(magent_amd)
                extra["c5_train_round_2x40k"] = train_round_extra(torch, magent_amd)
                # BASELINE config 5 at the size it names: train_battle.py --map_size 3536, 2 x 499,849 agents in its own formation
                extra["c5_train_round_1m"] = {"bf16_policy": train_round_extra(torch, magent_amd, map_size=3536, steps=6),
                                              "f32_policy": train_round_extra(torch, magent_amd, map_size=3536, steps=4, infer_dtype="f32")}
                os.environ["MAGENT_POLICY_F32"] = "torch"      # (the same loop with PyTorch's float32 forward pass: what round 5 measured)
                try:
                    extra["c5_train_round_1m"]["f32_policy_pytorch"] = train_round_extra(torch, magent_amd, map_size=3536, steps=2, train=False, infer_dtype="f32")
                finally:
                    del os.environ["MAGENT_POLICY_F32"]
            except Exception as e:     # secondary lines never fail the bench
                extra["error"] = repr(e)
            rec["extra"] = extra
    if world > 1 and (is_default_workload(args) or args.force_extra) and not args.no_extras:
        # north-star configuration 4 on the ranks of this job: gather 500 x 500, 100k agents + 20k food per replica, every replica's
        # observation tensor to every rank each step (counts first, rows sized by count, on a side stream under the step) -- timed
        # without and with the exchange, every gathered shard checked bit for bit against the rows its owner rendered.
        # The headline above is already measured: whatever happens in here -- an exception on some rank, a collective that never
        # returns (this leg has never run on xGMI) -- rank 0 still prints its ONE line: a watchdog on every rank ends the process
        # after --extra-timeout seconds, rank 0 printing the headline with the failure noted first.
        import threading
        XGMI_LINK_GBS = 153.0          # MI355X_MICROARCH.md: one xGMI link, per direction
        done_flag = threading.Event()

        def bail(why=None):
            if done_flag.is_set():
                return
            done_flag.set()
            if rank == 0:
                rec.setdefault("extra", {})["c4_gather_rccl"] = {"error": why or "did not finish within %d s (a rank failed, or a collective did not return)" % args.extra_timeout}
                emit(rec)
            os._exit(0)
        watchdog = threading.Timer(args.extra_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        # a rank that DIES in here (not: raises) makes the launcher end the others with SIGTERM: rank 0 answers it with its one line
        relay_sigterm(lambda: bail("the launcher ended this rank with SIGTERM inside the extra: another rank died (the headline above was already measured)"))
        try:
            if args.fault == "kill-rank-1-in-extra" and rank == 1:
                os._exit(17)
            A = measure("gather", 500, 100000, args.steps, args.warmup, False, gather="none")
            B = measure("gather", 500, 100000, args.steps, args.warmup, False, gather="obs", check_gather=True)
            ea = replicas.max_over_replicas(A["elapsed"], device=red_dev)
            eb = replicas.max_over_replicas(B["elapsed"], device=red_dev)
            na = replicas.sum_over_replicas(A["agent_steps"], device=red_dev)
            nb = replicas.sum_over_replicas(B["agent_steps"], device=red_dev)
            ex = replicas.max_over_replicas(B["gather"]["exchange_ms"], device=red_dev)
            if rank == 0:
                gb = B["gather"]
                link = gb["bytes_to_each_peer"] / (ex * 1e-3) / 1e9 if ex > 0 else None
                rec.setdefault("extra", {})["c4_gather_rccl"] = {
                    "workload": "gather 500x500 (train_gather.py), 100k agents + 20k food per replica, %d replicas, only the agents act" % world,
                    "backend": args.backend + (" (RCCL over xGMI)" if args.backend == "nccl" else " (dry run: rows staged through host memory)"),
                    "ms_per_step_without_gather": ea / args.steps * 1e3, "agent_steps_per_s_without_gather": na / ea,
                    "ms_per_step_with_gather": eb / args.steps * 1e3, "agent_steps_per_s_with_gather": nb / eb,
                    "exchange_ms": ex, "payload_bytes_to_each_peer": gb["bytes_to_each_peer"], "payload_bytes_sent_per_step": gb["payload_bytes_sent_per_step"],
                    "GBps_per_link": link, "xgmi_link_peak_GBps": XGMI_LINK_GBS, "link_frac": (link / XGMI_LINK_GBS if link else None),
                    "view_buffers": gb["view_buffers"], "verified": gb.get("verified")}
            done_flag.set()
            watchdog.cancel()
        except Exception as e:             # (the other ranks are now waiting in a collective: everybody leaves through the watchdog)
            sys.stderr.write("rank %d: extra.c4_gather_rccl failed: %r\n" % (rank, e))
            if rank == 0:
                rec.setdefault("extra", {})["c4_gather_rccl"] = {"error": repr(e)}
                emit(rec)
            done_flag.set()
            os._exit(0)
    if rank == 0:
        emit(rec)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

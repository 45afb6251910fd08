"""GPU tests at BASELINE.json's full size through size-independent properties of the domain, plus equivalence of the
host-buffer ABI and the device-resident ABI.  (Bit-exact comparison with the oracle at full size is
tests/test_gpu_parity.py::test_full_size_battle_two_steps.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

ROOT = H.ROOT

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _make(game, size, n, seed=12345, over=None):
    import magent_amd
    env = H.gridworld(H.config_for(game, size, **(over or {})), lib=H.HIP_LIB)
    env.set_seed(seed)
    env.reset()
    for h in env.get_handles():
        env.add_agents(h, "random", n=n)
    return env


def test_full_size_invariants():
    """battle 1000x1000, 2x400k agents, 6 steps on the device-resident path"""
    torch = _torch()
    dev = torch.device("cuda", 0)
    env = _make("battle", 1000, 400000, over={"small": {"hp": 4, "damage": 3}})   # low hp: many deaths per step
    handles = env.get_handles()
    W = 1000
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    total_dead = 0
    for step in range(6):
        cells_all, n_alive_all = [], 0
        for g, h in enumerate(handles):
            n = env.get_num(h)
            view, feat = env.get_observation_device(h)
            view2, feat2 = env.get_observation_device(h)
            pos = env.get_info_device(h, "pos", torch.empty((n, 2), dtype=torch.int32, device=dev))
            hp = env.get_info_device(h, "hp", torch.empty(n, dtype=torch.float32, device=dev))
            env.sync()
            # idempotence: rendering twice gives the same bits
            assert torch.equal(view.view(torch.int32), view2.view(torch.int32)) and torch.equal(feat, feat2)
            # every agent sees itself in the centre: own "has" = 1, own hp channel = hp / type.hp (all alive here)
            assert bool((view[:, 6, 6, 1] == 1).all())
            assert torch.equal(view[:, 6, 6, 2], hp / 4.0)
            # wall channel: 1 exactly where the window leaves the inner map (x or y == 0 or W-1), inside the view range
            xs, ys = pos[:, 0].long(), pos[:, 1].long()
            left = view[:, 6, :, 0]                                   # centre row, 13 columns
            col = torch.arange(13, device=dev)[None, :] - 6 + xs[:, None]
            assert torch.equal(left, ((col == 0) | (col == W - 1)).float())
            # minimap channels: own-group map sums to 1 (+1 self marker), float tolerance 1e-4
            s_own = view[:, :, :, 3].sum(dim=(1, 2))
            assert torch.allclose(s_own, torch.full_like(s_own, 2.0), atol=1e-3)
            # features: x / w and y / h in the last two slots
            # (expected values by numpy on the host: torch's device division is not IEEE-rounded, the engine's is)
            fx = pos[:, 0].cpu().numpy().astype(np.float32) / np.float32(W)
            fy = pos[:, 1].cpu().numpy().astype(np.float32) / np.float32(W)
            assert np.array_equal(feat[:, 32].cpu().numpy(), fx) and np.array_equal(feat[:, 33].cpu().numpy(), fy)
            cells_all.append(ys * W + xs)
            n_alive_all += n
            acts = torch.randint(21, (n,), dtype=torch.int32, device=dev, generator=gen)
            env.order_after_torch()          # the engine's stream waits for torch's stream to have produced `acts`
            env.set_action_device(h, acts)
            torch.cuda.synchronize(); env.sync()
        # no two agents share a cell
        cells = torch.cat(cells_all)
        assert torch.unique(cells).numel() == cells.numel() == n_alive_all
        env.step()
        expect = []
        for h in handles:
            n = env.get_num(h)
            alive = env.get_info_device(h, "alive", torch.empty(n, dtype=torch.uint8, device=dev))
            hp = env.get_info_device(h, "hp", torch.empty(n, dtype=torch.float32, device=dev))
            rew = env.get_reward_device(h)
            env.sync()
            assert bool(torch.isfinite(rew).all())
            assert torch.equal(alive.bool(), hp >= 0)                 # dead iff hp < 0 (GridWorld.h:205)
            # dead_penalty overwrites what was accumulated (GridWorld.h:207); a victim whose own attack landed before it
            # died still collects the 'attack' rule bonus afterwards (RewardEngine loops over dead agents too)
            rd = rew[~alive.bool()]
            lo, hi = np.float32(-0.1), np.float32(-0.1) + np.float32(0.2)
            assert bool(((rd == float(lo)) | (rd == float(hi))).all())
            dead = int((~alive.bool()).sum())
            total_dead += dead
            expect.append(n - dead)
        env.clear_dead()
        assert [env.get_num(h) for h in handles] == expect            # conservation
        env.clear_dead()                                              # a second clear_dead removes nobody
        assert [env.get_num(h) for h in handles] == expect
    assert total_dead > 10000


def test_host_and_device_abi_agree():
    """the reference ABI (numpy buffers) and the device-resident ABI return the same bits"""
    torch = _torch()
    dev = torch.device("cuda", 0)
    env = _make("gather", 120, 3000)
    handles = env.get_handles()
    rs = np.random.RandomState(3)
    for step in range(5):
        for h in handles:
            n = env.get_num(h)
            v_host, f_host = env.get_observation(h)
            v_dev, f_dev = env.get_observation_device(h)
            env.sync()
            assert v_dev.cpu().numpy().tobytes() == v_host.tobytes() and f_dev.cpu().numpy().tobytes() == f_host.tobytes()
        acts = rs.randint(33, size=env.get_num(handles[1])).astype(np.int32)
        d_acts = torch.from_numpy(acts).to(dev)
        torch.cuda.synchronize()                 # (the copy runs on torch's stream: before the hand-over)
        env.set_action_device(handles[1], d_acts)
        env.sync()
        env.step()
        for h in handles:
            r_dev = env.get_reward_device(h)
            env.sync()
            assert r_dev.cpu().numpy().tobytes() == env.get_reward(h).tobytes()
            n = env.get_num(h)
            p_dev = env.get_info_device(h, "pos", torch.empty((n, 2), dtype=torch.int32, device=dev))
            env.sync()
            assert np.array_equal(p_dev.cpu().numpy(), env.get_pos(h))
        env.clear_dead()


def test_two_environments_in_one_process_are_independent():
    """no global state: two engines stepped alternately give what each gives alone"""
    a = H.run(H.scenarios()["battle_brawl"], H.HIP_LIB)
    sc1, sc2 = H.scenarios()["battle_brawl"], H.scenarios()["gather"]
    env1, h1 = sc1.build(H.HIP_LIB)
    env2, h2 = sc2.build(H.HIP_LIB)
    rs1, rs2 = np.random.RandomState(sc1.action_seed), np.random.RandomState(sc2.action_seed)
    for step in range(10):
        for g, h in enumerate(h1):
            v, f = env1.get_observation(h)
            assert v.tobytes() == a[step]["view%d" % g].tobytes()
            env1.get_agent_id(h)
            env1.set_action(h, rs1.randint(21, size=env1.get_num(h)).astype(np.int32))
        env2.get_observation(h2[1])
        env2.set_action(h2[1], rs2.randint(33, size=env2.get_num(h2[1])).astype(np.int32))
        env2.step()
        env1.step()
        for g, h in enumerate(h1):
            assert env1.get_reward(h).tobytes() == a[step]["reward%d" % g].tobytes()
        env2.clear_dead()
        env1.clear_dead()


def test_unaligned_output_pointers_take_the_scalar_store_path():
    """a caller may hand a view / feature pointer that is not 16-byte aligned: same bits through the scalar path"""
    torch = _torch()
    dev = torch.device("cuda", 0)
    env = _make("battle", 60, 777)
    h = env.get_handles()[0]
    n = env.get_num(h)
    v_ref, f_ref = env.get_observation_device(h)
    S, F = int(np.prod(env.get_view_space(h))), env.get_feature_space(h)[0]
    big_v = torch.zeros(n * S + 8, dtype=torch.float32, device=dev)
    big_f = torch.zeros(n * F + 8, dtype=torch.float32, device=dev)
    for off in (1, 2, 3):
        v = big_v[off:off + n * S].view((n,) + env.get_view_space(h))
        f = big_f[off:off + n * F].view(n, F)
        assert v.data_ptr() % 16 != 0
        env.get_observation_device(h, v, f)
        env.sync()
        assert torch.equal(v.view(torch.int32), v_ref.view(torch.int32)) and torch.equal(f, f_ref)
        assert float(big_v[:off].abs().sum()) == 0 and float(big_v[off + n * S:].abs().sum()) == 0   # nothing outside
        big_v.zero_(); big_f.zero_()
        torch.cuda.synchronize()


def test_reference_1m_methodology_small():
    """the reference's own throughput harness (scripts/test/test_1m.py:66-74): pursuit-like game on a
    sqrt(20 N) map with N/10 random walls, N/2 prey, N/2 2x2 predators -- here N = 20000, compared with the oracle"""
    N = 20000
    size = int(np.sqrt(N * 20))
    sc = H.Scenario("test_1m_small", "pursuit", size, walls=N // 10,
                    place=[(1, "random", {"n": N // 2}), (0, "random", {"n": N // 2})], steps=4, action_seed=31, obs_every=2)
    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, H.HIP_LIB), "test_1m_small")


def test_info_queries_match_oracle():
    """the cold get_info names of the interactive tools: walls_info, groups_info, view2attack, global_minimap"""
    sc = H.scenarios()["battle_walls"]
    a, ha = sc.build(H.HIP_LIB)
    b, hb = sc.build(H.ensure_oracle())
    wa, wb = a._get_walls_info(), b._get_walls_info()
    assert wa.shape == wb.shape and np.array_equal(wa, wb) and len(wa) > 150
    assert np.array_equal(a._get_groups_info(), b._get_groups_info())
    for x, y in zip(ha, hb):
        assert a.get_view2attack(x)[0] == b.get_view2attack(y)[0] and np.array_equal(a.get_view2attack(x)[1], b.get_view2attack(y)[1])
    assert a.get_global_minimap(7, 9).tobytes() == b.get_global_minimap(7, 9).tobytes()


def test_env_batch_equals_standalone_environments():
    """magent_amd.EnvBatch (env_cycle_many: every small world of the batch in ONE pair of launches, the others one by one) and
    step_many give every environment exactly what it computes when driven alone.  Environment 2 is too large for the one-launch
    step (> 16384 agents): it leaves a skip marker in the batch and runs through the ordinary calls inside the same round."""
    torch = _torch()
    import magent_amd
    dev = torch.device("cuda", 0)
    K, STEPS = 5, 8
    SIZE = [(36, 300), (36, 300), (150, 8400), (36, 300), (40, 350)]     # (map, agents per group)

    def make(k):
        env = H.gridworld(H.config_for("battle", SIZE[k][0], small={"hp": 4, "damage": 3}), lib=H.HIP_LIB)
        env.set_seed(100 + k); env.reset()
        for h in env.get_handles():
            env.add_agents(h, "random", n=SIZE[k][1])
        return env

    gen = torch.Generator(device=dev); gen.manual_seed(5)
    acts = [[[torch.randint(21, (SIZE[k][1],), dtype=torch.int32, device=dev, generator=gen) for _ in range(2)] for k in range(K)] for _ in range(STEPS)]
    torch.cuda.synchronize()

    def buffers():
        return ([[torch.zeros((SIZE[k][1], 13, 13, 7), device=dev) for _ in range(2)] for k in range(K)],
                [[torch.zeros((SIZE[k][1], 34), device=dev) for _ in range(2)] for k in range(K)],
                [[torch.zeros(SIZE[k][1], device=dev) for _ in range(2)] for k in range(K)])

    # (a) alone, one call at a time
    solo, log_a = [make(k) for k in range(K)], []
    va, fa, ra = buffers()
    for s in range(STEPS):
        for k, e in enumerate(solo):
            for g, h in enumerate(e.get_handles()):
                e.get_observation_device(h, va[k][g], fa[k][g]); e.set_action_device(h, acts[s][k][g])
            e.step()
            for g, h in enumerate(e.get_handles()):
                e.get_reward_device(h, ra[k][g])
            e.sync()
            log_a.append([t.clone() for t in va[k] + fa[k] + ra[k]] + [e.get_pos(h).copy() for h in e.get_handles()])
            e.clear_dead()
    # (b) batched; the second half of the rounds hands over prebuilt pointer arrays instead of tensor lists
    envs, log_b = [make(k) for k in range(K)], []
    batch = magent_amd.EnvBatch(envs, n_threads=3)
    vb, fb, rb = buffers()
    vp, fp, rp = batch.pointers(vb), batch.pointers(fb), batch.pointers(rb)
    for s in range(STEPS):
        assert batch.nums() == [[e.get_num(h) for h in e.get_handles()] for e in envs]
        # positions must be read before clear_dead: cycle() includes it, so compare the post-clear state instead
        if s < STEPS // 2:
            batch.cycle(vb, fb, acts[s], rb)
        else:
            batch.cycle(vp, fp, batch.pointers(acts[s]), rp)
        for e in envs:
            e.sync()
        log_b.append([[t.clone() for t in vb[k] + fb[k] + rb[k]] for k in range(K)])
    i = 0
    for s in range(STEPS):
        for k in range(K):
            for ta, tb in zip(log_a[i][:6], log_b[s][k]):
                assert torch.equal(ta.view(torch.int32), tb.view(torch.int32)), (s, k)
            i += 1
    for a, b in zip(solo, envs):
        for ha, hb in zip(a.get_handles(), b.get_handles()):
            assert np.array_equal(a.get_pos(ha), b.get_pos(hb)) and a.get_num(ha) == b.get_num(hb)
    assert min(min(x) for x in batch.nums()) < 300      # agents did die: compaction ran inside the batched launches
    # (c) step_many
    e1, e2 = make(0), make(1)
    for e in (e1, e2):
        for g, h in enumerate(e.get_handles()):
            e.set_action_device(h, acts[0][0][g])
    dones = magent_amd.step_many([e1, e2])
    assert dones == [False, False]


def _play_actions_first(lib, map_size, n, steps, device_api):
    """set_action for every group BEFORE the observations of the step are asked for (a legal order: the reference's feature rows
    then already show the new last_action, GridWorld.cc:386-396 / Agent::set_action); the attack list's order does not depend on it"""
    env = H.gridworld(H.config_for("battle", map_size, small={"hp": 4, "damage": 3}), lib=lib)
    env.set_seed(77); env.reset()
    handles = env.get_handles()
    for h in handles:
        env.add_agents(h, "random", n=n)
    rs = np.random.RandomState(5)
    out = []
    for step in range(steps):
        rec = {}
        for g, h in enumerate(handles):
            acts = rs.randint(21, size=env.get_num(h)).astype(np.int32)
            if device_api:
                torch = _torch()
                d = torch.from_numpy(acts).to(torch.device("cuda", env.device_id))
                torch.cuda.synchronize()
                env.set_action_device(h, d)
            else:
                env.set_action(h, acts)
        for g, h in enumerate(handles):
            if device_api:
                v, f = env.get_observation_device(h)
                env.sync()
                rec["view%d" % g], rec["feat%d" % g] = v.cpu().numpy(), f.cpu().numpy()
            else:
                v, f = env.get_observation(h)
                rec["view%d" % g], rec["feat%d" % g] = v.copy(), f.copy()
        rec["done"] = np.array([env.step()], dtype=np.int32)
        for g, h in enumerate(handles):
            rec["reward%d" % g] = env.get_reward(h)
            rec["alive%d" % g] = env.get_alive(h).astype(np.uint8)
        env.clear_dead()
        for g, h in enumerate(handles):
            rec["pos%d" % g] = env.get_pos(h)
        out.append(rec)
    return out


@pytest.mark.parametrize("map_size,n", [(40, 300), (160, 9000)])
def test_observation_between_set_action_and_step(map_size, n):
    """the engine stores last_action lazily (with MAGENT_TUNE=overlap=3 set_action runs on a side stream under the renders of large
    worlds): an observation asked for after set_action must still show it.  Small world (one-launch step) and large world, host and
    device API; the large world once more in a process with the side stream on."""
    want = _play_actions_first(H.ensure_oracle(), map_size, n, 5, False)
    H.assert_same(want, _play_actions_first(H.HIP_LIB, map_size, n, 5, False), "actions first, host API")
    H.assert_same(want, _play_actions_first(H.HIP_LIB, map_size, n, 5, True), "actions first, device API")
    if n > 8192 and "overlap" not in os.environ.get("MAGENT_TUNE", ""):
        env = H.merge_env(os.environ, {"MAGENT_TUNE": "overlap=3"})
        out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__, "-k", "test_observation_between_set_action_and_step"],
                             env=env, capture_output=True, text=True, timeout=900, cwd=H.ROOT)
        assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])


def test_moving_goals_take_the_literal_loop():
    """a can_absorb group that is given actions (rounds 1-3 refused a goal that moves): the step runs the reference's own loops on the
    device and matches the oracle -- `arrange_goals_move*` in the parity suites pin it on the compiled reference's digests; here: that
    nothing aborts and the goals really moved"""
    sc = H.scenarios()["arrange_goals_move"]
    env, handles = sc.build(H.HIP_LIB)
    before = env.get_pos(handles[0]).copy()
    for _ in range(3):
        for h in handles:
            env.set_action(h, np.random.RandomState(1).randint(env.get_action_space(h)[0], size=env.get_num(h)).astype(np.int32))
        env.step()
    assert (env.get_pos(handles[0]) != before).any()

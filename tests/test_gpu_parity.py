"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle, bit for bit.

Every observable output of every step is compared: view and feature tensors, ids, rewards, alive masks, positions,
group sizes, the done flag.  Floats are compared as raw 32-bit patterns (tolerance: none -- the path is integer /
index work plus float adds, subtracts and divisions replayed in the reference's order).
"""
import json
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

SCENARIOS = H.scenarios()
with open(os.path.join(H.GOLDEN_DIR, "digests.json")) as f:
    DIGESTS = json.load(f)


@pytest.fixture(scope="module")
def oracle():
    return H.ensure_oracle()


@pytest.mark.parametrize("name", sorted(k for k in SCENARIOS if SCENARIOS[k].engine))
def test_hip_matches_oracle_and_golden(name, oracle):
    got = H.run(SCENARIOS[name], H.HIP_LIB)
    want = H.run(SCENARIOS[name], oracle)
    H.assert_same(want, got, name)
    assert H.digest(got) == DIGESTS[name]["sha256"]     # the committed vectors from the compiled reference


def test_mean_info_matches_oracle_and_reference(oracle):
    """get_info("mean_info") (GridWorld.cc:765-786: mean position, the share of every action; float sums in agent order): the HIP engine
    against the oracle -- and against the compiled reference where it travelled -- between set_action and step, behind the step and behind
    clear_dead; scenarios of both step drivers (the one-launch step, the pipeline of large_map_mode worlds) and a large group (2 x 40,000)"""
    names = ["battle_small_dense", "battle_walls", "battle_largemap_odd", "gather", "pursuit_dense", "tri_rect", "bodies", "bodies_turn", "battle_events",
             "battle_brawl_dense_big"]
    for name in names:
        sc = SCENARIOS[name]
        want, got = H.mean_info_trace(sc, oracle), H.mean_info_trace(sc, H.HIP_LIB)
        assert len(want) == len(got) and len(want) > 0, name
        for k, (a, b) in enumerate(zip(want, got)):
            assert a.tobytes() == b.tobytes(), (name, k, a, b)
        if H.have_ref():
            ref = H.mean_info_trace(sc, H.REF_LIB)
            assert [a.tobytes() for a in ref] == [a.tobytes() for a in got], name


@pytest.mark.parametrize("name", sorted(k for k in SCENARIOS if SCENARIOS[k].engine and SCENARIOS[k].clear_every == 1))
def test_fused_cycle_matches_oracle(name, oracle):
    """env_cycle_many / EnvBatch.cycle: a whole environment cycle in two launches for small worlds (k_render_multi, then
    set_action + step + get_reward + clear_dead + the next minimap inside k_step_solo) -- against the oracle driven call by call"""
    got = H.run_cycle(SCENARIOS[name], H.HIP_LIB, fused=True)
    want = H.run_cycle(SCENARIOS[name], oracle, fused=False)
    H.assert_same(want, got, name + " (cycle)")


def test_batch_cycle_of_worlds_given_their_actions_beforehand(oracle):
    """env_set_action_device on worlds of 2000 agents (beyond the one-launch step's limit for an environment on its own:
    the tiled set_action), then EnvBatch.cycle with no actions over the two of them (within the batch's limit: one launch per step)"""
    scs = H.preset_batch_scenarios()
    got = H.run_cycle_batch(scs, H.HIP_LIB, preset=True)
    for sc, g in zip(scs, got):
        H.assert_same(H.run_cycle(sc, oracle, fused=False, preset=True), g, sc.name + " (preset actions, batch of 2)")


def test_batched_pipeline(oracle):
    """env_cycle_many over worlds beyond the one-launch step: one launch per phase of the plain pipeline for all of them (pipe.hip), beside a
    world on the two-launch cycle and one that goes alone in the same call; every environment against the oracle driven alone through the
    reference call sequence.  Then two worlds of 80,000 agents with kills from the first step (their renders are launches of their own:
    the sweeping kernel) beside a 2000-agent one."""
    import copy
    scs = H.pipe_batch_scenarios()
    seen = []
    got = H.run_cycle_batch(scs, H.HIP_LIB, envs_out=seen)
    for sc, g in zip(scs, got):
        H.assert_same(H.run_cycle(sc, oracle, fused=False), g, sc.name + " (batched pipeline)")
    stats = [e.pipeline_stats() for e in seen]
    assert all(s[6] >= 8 for s in stats[:3]) and stats[3][6] == 0 and stats[4][6] == 0, stats
    assert all(s[7] == s[6] for s in stats[:3]), stats      # battle-shaped observations: the batch's render was the sweeping kernel in every cycle
    big = []
    for k in range(2):
        c = copy.deepcopy(SCENARIOS["battle_brawl_dense_big"]); c.seed, c.action_seed, c.obs_every = 777 + k, 50 + k, 1
        big.append(c)
    big.append(H.pipe_batch_scenarios()[0])
    seen = []
    got = H.run_cycle_batch(big, H.HIP_LIB, envs_out=seen)
    for sc, g in zip(big, got):
        H.assert_same(H.run_cycle(sc, oracle, fused=False), g, sc.name + " (batched pipeline, 2 x 80,000 agents + 2000)")
    assert all(e.pipeline_stats()[6] >= 6 for e in seen), [e.pipeline_stats() for e in seen]


def test_batched_pipeline_over_a_long_episode(oracle):
    """the batched pipeline over what only acts with the length of an episode (the claim words' epoch window, refilled every 63 steps by every
    environment of the batch for itself; the carried round stamps; the batch's budget of optimistic rounds: three, four while some environment's is raised):
    two 200 x 200 worlds of 2 x 4500 agents with different seeds -- `battle300_long`'s recipe at the same density: hp 4 / damage 3,
    reinforcements at steps 70 and 130, kills in every step -- for 150 steps through ONE EnvBatch, every environment against the oracle driven
    alone through the reference call sequence (observations compared every 10th step: the trajectories are held in memory)"""
    rnd = lambda g, n: (g, "random", {"n": n})
    scs = [H.Scenario("battle200_long_%d" % k, "battle", 200, seed=4000 + 17 * k, place=[rnd(0, 4500), rnd(1, 4500)], steps=150, action_seed=300 + k, obs_every=10,
                      over={"small": {"hp": 4, "damage": 3}},
                      events={70: [("add", 0, "random", {"n": 4000}), ("add", 1, "random", {"n": 4000})],
                              130: [("add", 0, "random", {"n": 4000}), ("add", 1, "random", {"n": 4000})]}) for k in range(2)]
    seen = []
    got = H.run_cycle_batch(scs, H.HIP_LIB, envs_out=seen)
    for sc, g in zip(scs, got):
        assert len(g) == 150
        H.assert_same(H.run_cycle(sc, oracle, fused=False), g, sc.name + " (batched pipeline, 150 steps)")
    for e in seen:
        st = e.pipeline_stats()
        assert st[6] == 150 and st[3] >= 3, st          # every cycle through the batch; the window refilled at steps 1, 63, 126


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) did not travel")
@pytest.mark.parametrize("name", ["battle_brawl", "battle_largemap", "gather"])
def test_hip_matches_compiled_reference(name):
    H.assert_same(H.run(SCENARIOS[name], H.REF_LIB), H.run(SCENARIOS[name], H.HIP_LIB), name)


def test_full_size_battle_two_steps(oracle):
    """BASELINE.json's headline size (battle 1000x1000, 2x400k agents, the workload bench.py times): two full steps,
    every output compared with the oracle bit for bit (3.9 GB of observations per step)."""
    import hashlib
    sc = H.Scenario("battle_c3", "battle", 1000, place=[(0, "random", {"n": 400000}), (1, "random", {"n": 400000})], steps=2)
    state = {}

    def keep(step, rec):  # hash step by step so that only one step of one engine is held in memory
        state.setdefault("steps", []).append({k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in rec.items()})
        rec.clear()

    H.run(sc, H.HIP_LIB, record=keep)
    got, state = state["steps"], {}
    H.run(sc, oracle, record=keep)
    want = state["steps"]
    assert len(got) == len(want) == 2
    for s in range(2):
        for k in sorted(want[s]):
            assert got[s][k] == want[s][k], "step %d %s differs" % (s, k)


def test_render_text_dump_matches_reference(tmp_path):
    """env.render(): config.json and video_N.txt byte-identical to what the reference's RenderGenerator wrote
    (tests/golden/render_battle16, generated from the compiled reference), attack events included"""
    got = H.render_episode(H.HIP_LIB, str(tmp_path))
    gold_dir = os.path.join(H.GOLDEN_DIR, "render_battle16")
    want = {name: open(os.path.join(gold_dir, name), "rb").read() for name in sorted(os.listdir(gold_dir))}
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name


def test_render_text_dump_with_repeated_set_action(tmp_path):
    """the same with a group that is given actions twice before every other step: the attack events of the literal loop (k_step_serial)
    in the shuffled list's order, byte-identical to the reference's dump (tests/golden/render_battle16_twice)"""
    got = H.render_episode(H.HIP_LIB, str(tmp_path), twice=True)
    gold_dir = os.path.join(H.GOLDEN_DIR, "render_battle16_twice")
    want = {name: open(os.path.join(gold_dir, name), "rb").read() for name in sorted(os.listdir(gold_dir))}
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name


@pytest.mark.parametrize("block", range(4))
def test_fuzz_random_games(block, oracle):
    """differential fuzzing (tests/helpers.fuzz_scenario): random in-scope games -- group count, body sizes, ranges,
    hp / damage / recover / kill_supply, in-group attack, random rules, map shape, walls, density, clear_dead cadence"""
    for seed in range(block * 25, block * 25 + 25):
        sc = H.fuzz_scenario(seed)
        H.assert_same(H.run(sc, oracle), H.run(sc, H.HIP_LIB), sc.name)

"""CPU tests (no GPU): pin the oracle restatement (oracle/liboracle.so).

1. against the committed golden vectors generated from the compiled reference (tests/golden/make_golden.py);
2. against the compiled reference itself when oracle/_ref is present (build container; travels to the GPU box);
3. against the hand-checked known answers of SURVEY.md Appendix B.
"""
import json
import os

import numpy as np
import pytest

import helpers as H

ORACLE = H.ensure_oracle()
with open(os.path.join(H.GOLDEN_DIR, "digests.json")) as f:
    DIGESTS = json.load(f)
SCENARIOS = H.scenarios()


def test_golden_covers_every_scenario():
    assert sorted(DIGESTS) == sorted(SCENARIOS)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_oracle_matches_golden_digest(name):
    traj = H.run(SCENARIOS[name], ORACLE)
    assert len(traj) == DIGESTS[name]["steps"]
    assert [int(traj[-1][k][0]) for k in sorted(traj[-1]) if k.startswith("num")][:2] == DIGESTS[name]["final_num"]
    assert H.digest(traj) == DIGESTS[name]["sha256"]


@pytest.mark.parametrize("name", ["battle_tiny", "battle_one_side"])
def test_oracle_matches_golden_arrays(name):
    gold = np.load(os.path.join(H.GOLDEN_DIR, "kat_%s.npz" % name))
    traj = H.run(SCENARIOS[name], ORACLE)
    ref = [{k.rsplit("_s", 1)[0]: gold[k] for k in gold.files if k.endswith("_s%d" % s)} for s in range(len(traj))]
    H.assert_same(ref, traj, name)


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("name", sorted(k for k in SCENARIOS if k not in ("battle_brawl_dense_big", "battle_brawl_big")))
def test_oracle_matches_compiled_reference(name):
    # (the two largest brawls -- 80,000 and 9,000 agents, two minutes of the single-threaded reference -- are pinned to it through the digests
    # tests/golden/make_golden.py took from the compiled reference: test_oracle_matches_golden_digest)
    H.assert_same(H.run(SCENARIOS[name], H.REF_LIB), H.run(SCENARIOS[name], ORACLE), name)


def test_appendix_b_known_answers():
    """SURVEY.md Appendix B, values hand-checked against the reference semantics"""
    import magent_amd
    env = H.gridworld("battle", lib=ORACLE, map_size=30)
    env.reset()
    h0, h1 = env.get_handles()
    env.add_agents(h0, "custom", pos=[(10, 12), (3, 3)])
    env.add_agents(h1, "custom", pos=[(11, 12), (10, 14)])
    assert env.get_view_space(h0) == (13, 13, 7) and env.get_feature_space(h0) == (34,) and env.get_action_space(h0) == (21,)
    assert list(env.get_agent_id(h0)) == [0, 1] and list(env.get_agent_id(h1)) == [2, 3]
    base, table = env.get_view2attack(h0)
    expect = -np.ones((13, 13), np.int32)
    expect[5:8, 5:8] = [[0, 1, 2], [3, -1, 4], [5, 6, 7]]
    assert base == 13 and np.array_equal(table, expect)
    view, feat = env.get_observation(h0)
    v = view[0]
    assert v[:, :, 0].sum() == 0                                  # no wall in sight
    assert v[6, 6, 1] == 1 and v[6, 6, 2] == 1.0                  # itself: has + hp
    assert v[6, 7, 4] == 1 and v[8, 6, 4] == 1 and v[6, 7, 5] == 1  # the two enemies
    assert v[1, 1, 3] == 0.5 and v[4, 3, 3] == 1.5 and v[4, 3, 6] == 2.0  # minimaps + self marker on BOTH
    assert np.count_nonzero(feat[0]) == 2 and feat[0, 32] == np.float32(10) / np.float32(30) and feat[0, 33] == np.float32(0.4)
    assert feat[1, 0] == 1 and feat[1, 32] == np.float32(0.1)
    hp_seen = []
    for s in range(6):
        env.set_action(h0, np.array([17, 6], np.int32))
        env.set_action(h1, np.array([6, 6], np.int32))
        done = env.step()
        r0, r1 = env.get_reward(h0), env.get_reward(h1)
        if s < 5:
            assert r0[0] == np.float32(-0.005) + (np.float32(0.0) + np.float32(-0.1)) + np.float32(0.2)
            assert list(env.get_alive(h1)) == [True, True]
        else:  # the kill: kill_reward + attack_penalty, no 'attack' bonus; victim reward overwritten by dead_penalty
            assert r0[0] == np.float32(-0.005) + (np.float32(5) + np.float32(-0.1))
            assert r1[0] == np.float32(-0.1) and list(env.get_alive(h1)) == [False, True] and env.get_num(h1) == 2
        assert not done
        env.clear_dead()
        if s < 5:
            hp_seen.append(env.get_observation(h1)[0][0, 6, 6, 2])
    assert env.get_num(h1) == 1
    assert np.allclose(hp_seen, [0.81, 0.62, 0.43, 0.24, 0.05], atol=1e-6)
    _, feat = env.get_observation(h0)
    assert feat[0, 10 + 17] == 1 and feat[0, 31] == np.float32(-0.005) + (np.float32(5) + np.float32(-0.1))


def test_appendix_b_golden_arrays():
    import golden.make_golden as mg
    gold = np.load(os.path.join(H.GOLDEN_DIR, "kat_appendix_b.npz"))
    got = mg.appendix_b(ORACLE)
    assert sorted(got) == sorted(gold.files)
    for k in gold.files:
        a, b = gold[k], got[k]
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), k


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_matches_compiled_reference_on_random_games():
    """differential fuzzing of the restatement itself (tools/fuzz_parity.py ran 3000 seeds clean in the build container)"""
    for seed in range(60):
        sc = H.fuzz_scenario(seed)
        H.assert_same(H.run(sc, H.REF_LIB), H.run(sc, ORACLE), sc.name)


MEAN_INFO = ["battle_small_dense", "battle_walls", "battle_largemap_odd", "gather", "pursuit_dense", "tri_rect", "bodies", "bodies_turn", "battle_events"]


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_mean_info_matches_compiled_reference():
    """get_info("mean_info") -- mean position and the share of every action, float sums in agent order -- between set_action and step,
    behind the step and behind clear_dead (GridWorld.cc:765-786; the reference under one OpenMP thread, as every parity run)"""
    for name in MEAN_INFO:
        sc = H.scenarios()[name]
        want, got = H.mean_info_trace(sc, H.REF_LIB), H.mean_info_trace(sc, ORACLE)
        assert len(want) == len(got) and len(want) > 0, name
        for k, (a, b) in enumerate(zip(want, got)):
            assert a.tobytes() == b.tobytes(), (name, k, a, b)


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_goal_mode_matches_compiled_reference(monkeypatch):
    """goal_mode (two feature slots nothing writes, GridWorld.cc:926-934) and set_goal between steps (two draws of the engine's
    generator per agent, GridWorld.cc:667-679) in random games (FUZZ_GOAL=1)"""
    monkeypatch.setenv("FUZZ_GOAL", "1")
    for seed in range(50):
        sc = H.fuzz_scenario(seed)
        H.assert_same(H.run(sc, H.REF_LIB), H.run(sc, ORACLE), sc.name)


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("knob", ["FUZZ_TWICE", "FUZZ_GOALS_ACT"])
def test_oracle_literal_loop_cases_match_compiled_reference(monkeypatch, knob):
    """groups given actions twice before a step (GridWorld.cc:403-454 appends) and goals given actions (they move, Map.cc:313-358) in random
    games, 60 % of them turn_mode"""
    monkeypatch.setenv(knob, "1")
    monkeypatch.setenv("FUZZ_TURN", "1")
    for seed in range(60):
        sc = H.fuzz_scenario(seed)
        H.assert_same(H.run(sc, H.REF_LIB), H.run(sc, ORACLE), sc.name)


@pytest.mark.skipif(not H.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_rule_search_matches_compiled_reference(monkeypatch):
    """random rule expressions over 'any' / 'all' / fixed-index symbols, in_a_line, several iterated symbols (FUZZ_RULES=2): the
    restated recursive search (RewardEngine.cc:216-443), the reference's Agent::index quirk included
    (tools/fuzz_parity.py ref oracle ran 6000 such seeds clean in the build container)"""
    monkeypatch.setenv("FUZZ_RULES", "2")
    for seed in list(range(40)) + [223, 358, 366, 688, 1024, 1131]:    # (the six that found the Agent::index quirk)
        sc = H.fuzz_scenario(seed)
        H.assert_same(H.run(sc, H.REF_LIB), H.run(sc, ORACLE), sc.name)

"""A group that is given actions more than once before a step.

GridWorld::set_action appends to the step's action lists (/root/reference/src/gridworld/GridWorld.cc:403-454): every agent of
such a group acts once per call -- two entries in the shuffled attack list, two moves in list order, the second from wherever
the first one ended; `last_action` is the latest call's.  Rounds 1-2 refused this; the engine serves it with the
reference's own sequential loops on one lane of the device (k_step_serial: exact by construction, slow, taken only when it
happens) -- for one-cell bodies since round 3, for every game since round 4.  Pinned three ways: compiled reference == oracle (here, when oracle/_ref is present), oracle == the HIP sources on
the emulator (here), oracle == the HIP engine (GPU)."""
import numpy as np
import pytest

import helpers as H

PATTERNS = [[(0, 1, 0)], [(0, 0, 1, 1)], [(1, 0, 1, 0, 1)], [(0, 1), (0, 1, 1), (0, 0), (1,)]]   # groups given actions, per step (cycled)
WORLDS = ((20, 60), (45, 400), (120, 1500))     # (map size, agents per group); 120 > 99: large_map_mode, moves run stripe by stripe


def play(lib, map_size, n, steps, seed, pattern, game="battle"):
    env = H.gridworld(game, lib=lib, map_size=map_size)
    env.set_seed(seed)
    env.reset()
    hs = env.get_handles()
    for h in hs:
        env.add_agents(h, "random", n=n)
    rs = np.random.RandomState(seed)
    out = []
    for step in range(steps):
        rec = {}
        order = pattern[step % len(pattern)]
        for k, g in enumerate(order):
            env.set_action(hs[g], rs.randint(env.get_action_space(hs[g])[0], size=env.get_num(hs[g])).astype(np.int32))
            if k == len(order) - 1:      # observed BEFORE the step: the feature rows show the latest call's action
                for gg, hh in enumerate(hs):
                    if env.get_num(hh):
                        rec["view%d" % gg], rec["feat%d" % gg] = [a.copy() for a in env.get_observation(hh)]
        rec["done"] = np.array([env.step()])
        for gg, hh in enumerate(hs):
            rec["reward%d" % gg] = env.get_reward(hh)
            rec["alive%d" % gg] = env.get_alive(hh).astype(np.uint8)
            rec["pos%d" % gg] = env.get_pos(hh)
        env.clear_dead()
        for gg, hh in enumerate(hs):
            if env.get_num(hh):
                rec["view_after%d" % gg] = env.get_observation(hh)[0].copy()
        out.append(rec)
    return out


@pytest.mark.parametrize("world", WORLDS, ids=lambda w: "map%d" % w[0])
def test_repeated_set_action_oracle_is_the_reference_and_the_kernels_are_the_oracle(world):
    emu = H.ensure_emu()
    for pi, pat in enumerate(PATTERNS):
        want = play(H.ensure_oracle(), world[0], world[1], 6, 3 + pi, pat)
        if H.have_ref():
            H.assert_same(play(H.REF_LIB, world[0], world[1], 6, 3 + pi, pat), want, "reference vs oracle, pattern %d" % pi)
        H.assert_same(want, play(emu, world[0], world[1], 6, 3 + pi, pat), "oracle vs emulated kernels, pattern %d" % pi)


def test_repeated_set_action_with_starvation():
    """gather: the agents lose hp every step (step_recover < 0) and starve; the food group never acts -- the serial step's starve loop"""
    emu = H.ensure_emu()
    for pat in ([(1, 1)], [(1,), (1, 1, 1)]):
        want = play(H.ensure_oracle(), 40, 120, 12, 5, pat, game="gather")
        if H.have_ref():
            H.assert_same(play(H.REF_LIB, 40, 120, 12, 5, pat, game="gather"), want, "reference vs oracle (gather)")
        H.assert_same(want, play(emu, 40, 120, 12, 5, pat, game="gather"), "oracle vs emulated kernels (gather)")


@pytest.mark.gpu
@pytest.mark.parametrize("world", WORLDS + ((300, 20000),), ids=lambda w: "map%d" % w[0])
def test_repeated_set_action_on_the_gpu(world):
    for pi, pat in enumerate(PATTERNS):
        H.assert_same(play(H.ensure_oracle(), world[0], world[1], 5, 3 + pi, pat), play(H.HIP_LIB, world[0], world[1], 5, 3 + pi, pat),
                      "oracle vs HIP engine, pattern %d" % pi)


# every game family the reference plays: bodies larger than one cell, turn_mode, food_mode, goals (given actions too: they move), sector
# ranges, kill_supply, three groups -- the scenarios' own configurations and placements, their groups given actions by PATTERNS
GAMES = ["bodies", "bodies_turn", "battle_food", "bodies_food", "arrange_live", "sector_turn", "duo", "pursuit_dense", "tri_turn", "battle_turn"]


def play_scenario(lib, name, steps, seed, pattern):
    sc = H.scenarios()[name]
    env, hs = sc.build(lib)
    rs = np.random.RandomState(seed)
    out = []
    for step in range(steps):
        rec = {}
        order = [g % len(hs) for g in pattern[step % len(pattern)]] + ([2] if len(hs) > 2 and step % 2 else [])
        for g in order:
            env.set_action(hs[g], rs.randint(env.get_action_space(hs[g])[0], size=env.get_num(hs[g])).astype(np.int32))
        rec["done"] = np.array([env.step()])
        for gg, hh in enumerate(hs):
            rec["reward%d" % gg] = env.get_reward(hh)
            rec["alive%d" % gg] = env.get_alive(hh).astype(np.uint8)
            rec["pos%d" % gg] = env.get_pos(hh)
        if step % 3 != 2:
            env.clear_dead()
        for gg, hh in enumerate(hs):
            if env.get_num(hh):
                rec["view%d" % gg], rec["feat%d" % gg] = [a.copy() for a in env.get_observation(hh)]
        out.append(rec)
    return out


@pytest.mark.parametrize("name", GAMES)
def test_repeated_set_action_in_every_game(name):
    emu = H.ensure_emu()
    for pi, pat in enumerate(PATTERNS[1:]):
        want = play_scenario(H.ensure_oracle(), name, 7, 11 + pi, pat)
        if H.have_ref():
            H.assert_same(play_scenario(H.REF_LIB, name, 7, 11 + pi, pat), want, "%s: reference vs oracle, pattern %d" % (name, pi))
        H.assert_same(want, play_scenario(emu, name, 7, 11 + pi, pat), "%s: oracle vs emulated kernels, pattern %d" % (name, pi))


@pytest.mark.gpu
@pytest.mark.parametrize("name", GAMES + ["bodies_large", "sector_turn_large", "arrange_large"])
def test_repeated_set_action_in_every_game_on_the_gpu(name):
    for pi, pat in enumerate(PATTERNS[1:]):
        H.assert_same(play_scenario(H.ensure_oracle(), name, 5, 11 + pi, pat), play_scenario(H.HIP_LIB, name, 5, 11 + pi, pat),
                      "%s: oracle vs HIP engine, pattern %d" % (name, pi))

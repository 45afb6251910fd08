"""Shared scenario driver for the parity tests.

One driver, three engines: the product (HIP, magent_amd/lib/libmagent.so), the CPU restatement
(oracle/liboracle.so) and -- when present -- the compiled reference (oracle/_ref/libmagent_ref.so).
All are driven through the same magent_amd.GridWorld wrapper with the same seeds and action streams.
"""
import hashlib
import os
import subprocess

import numpy as np

import magent_amd
from magent_amd import gridworld as gw  # noqa: F401
from magent_amd.builtin.config import _games

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libmagent_ref.so")
HIP_LIB = os.path.join(ROOT, "magent_amd", "lib", "libmagent.so")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def ensure_oracle():
    """build oracle/liboracle.so from its own source if needed (gcc only; no GPU involved)"""
    src = os.path.join(ROOT, "oracle", "gridworld_oracle.cc")
    if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return ORACLE_LIB


def have_ref():
    return os.path.exists(REF_LIB)


# ---------------------------------------------------------------------------------------------- scenarios
def config_for(game, map_size, **over):
    """built-in game, optionally with agent-type overrides {type_name: {attr: value}} for edge-case scenarios"""
    cfg = _games.make(game, map_size)
    for tname, attrs in over.items():
        cfg.agent_type_dict[tname].update(attrs)
    return cfg


class Scenario(object):
    """a reproducible episode: game + placement recipe + random action stream"""

    def __init__(self, name, game, map_size, seed=12345, place=(), steps=10, action_seed=0, walls=0,
                 acting=None, over=None, clear_every=1, obs_every=1):
        self.name, self.game, self.map_size, self.seed = name, game, map_size, seed
        self.place, self.steps, self.action_seed, self.walls = list(place), steps, action_seed, walls
        self.acting, self.over, self.clear_every, self.obs_every = acting, over or {}, clear_every, obs_every

    def build(self, lib):
        env = magent_amd.GridWorld(config_for(self.game, self.map_size, **self.over), lib=lib)
        env.set_seed(self.seed)
        env.reset()
        if self.walls:
            env.add_walls(method="random", n=self.walls)
        handles = env.get_handles()
        for g, method, kw in self.place:
            env.add_agents(handles[g], method, **kw)
        return env, handles


def run(sc, lib, record=None):
    """Play scenario `sc` on library `lib`; returns list (one dict per step) of every observable output.

    Order of calls per step follows examples/train_battle.py:61-109: get_observation + set_action per group,
    step, get_reward / get_alive / get_pos / get_num per group, clear_dead."""
    env, handles = sc.build(lib)
    rs = np.random.RandomState(sc.action_seed)
    acting = sc.acting if sc.acting is not None else list(range(len(handles)))
    out = []
    for step in range(sc.steps):
        rec = {}
        for g, h in enumerate(handles):
            n = env.get_num(h)
            if step % sc.obs_every == 0 and n > 0:
                view, feat = env.get_observation(h)
                rec["view%d" % g], rec["feat%d" % g] = view.copy(), feat.copy()
            rec["id%d" % g] = env.get_agent_id(h)
            if g in acting:
                env.set_action(h, rs.randint(env.get_action_space(h)[0], size=n).astype(np.int32))
        rec["done"] = np.array([env.step()], dtype=np.int32)
        for g, h in enumerate(handles):
            rec["reward%d" % g] = env.get_reward(h)
            rec["alive%d" % g] = env.get_alive(h).astype(np.uint8)
            rec["pos%d" % g] = env.get_pos(h)
            rec["num%d" % g] = np.array([env.get_num(h)], dtype=np.int32)
        if (step + 1) % sc.clear_every == 0:
            env.clear_dead()
        out.append(rec)
        if record is not None:
            record(step, rec)
        if all(env.get_num(h) == 0 for h in handles):
            break
    return out


def digest(trajectory):
    """SHA-256 over every array of every step (dtype + shape + bytes), in key order"""
    h = hashlib.sha256()
    for rec in trajectory:
        for k in sorted(rec):
            a = np.ascontiguousarray(rec[k])
            h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def assert_same(ta, tb, what=""):
    """bit-exact comparison of two trajectories (floats compared as raw uint32)"""
    assert len(ta) == len(tb), "%s: step counts differ %d vs %d" % (what, len(ta), len(tb))
    for s, (ra, rb) in enumerate(zip(ta, tb)):
        assert sorted(ra) == sorted(rb), "%s step %d: keys differ" % (what, s)
        for k in sorted(ra):
            a, b = np.ascontiguousarray(ra[k]), np.ascontiguousarray(rb[k])
            assert a.shape == b.shape and a.dtype == b.dtype, "%s step %d %s: shape/dtype %s%s vs %s%s" % (
                what, s, k, a.dtype, a.shape, b.dtype, b.shape)
            if a.dtype == np.float32:
                a, b = a.view(np.uint32), b.view(np.uint32)
            if not np.array_equal(a, b):
                bad = np.argwhere(a != b)
                raise AssertionError("%s step %d %s: %d mismatching elements, first at %s: %r vs %r" % (
                    what, s, k, len(bad), tuple(bad[0]), ra[k][tuple(bad[0])], rb[k][tuple(bad[0])]))


# the scenario table: sizes the CPU checkers finish in seconds.  Covers the reference's edge cases for this path:
# dense attacks (order dependence), walls, large_map_mode stripes (> 99x99), in-group attack (gather), a group that
# does not act (food), deaths + clear_dead compaction, skipped clear_dead (dead agents stay listed), map borders.
def scenarios():
    rnd = lambda g, n: (g, "random", {"n": n})
    S = [
        Scenario("battle_small_dense", "battle", 30, place=[rnd(0, 150), rnd(1, 150)], steps=25),
        Scenario("battle_brawl", "battle", 20, place=[rnd(0, 120), rnd(1, 120)], steps=25, action_seed=1,
                 over={"small": {"hp": 4, "damage": 3, "step_recover": 0}}),
        Scenario("battle_brawl_big", "battle", 110, place=[rnd(0, 4500), rnd(1, 4500)], steps=14, action_seed=2,
                 over={"small": {"hp": 5, "damage": 3, "step_recover": 0.3}}),
        Scenario("battle60", "battle", 60, place=[rnd(0, 1200), rnd(1, 1200)], steps=40),
        Scenario("battle_walls", "battle", 50, walls=200, place=[rnd(0, 400), rnd(1, 400)], steps=20, action_seed=3),
        Scenario("battle_largemap", "battle", 120, place=[rnd(0, 3000), rnd(1, 3000)], steps=12, action_seed=5),
        Scenario("battle_largemap_odd", "battle", 101, place=[rnd(0, 2500), rnd(1, 2500)], steps=10, action_seed=6),
        Scenario("battle_fill_full", "battle", 40,
                 place=[(0, "fill", {"pos": (1, 1), "size": (19, 38)}), (1, "fill", {"pos": (20, 1), "size": (19, 38)})],
                 steps=15, action_seed=7),
        Scenario("battle_no_clear", "battle", 40, place=[rnd(0, 400), rnd(1, 400)], steps=12, clear_every=3, action_seed=8),
        Scenario("battle_tiny", "battle", 8, place=[(0, "custom", {"pos": [(1, 1), (6, 6)]}), (1, "custom", {"pos": [(2, 1)]})],
                 steps=8, action_seed=9),
        Scenario("gather", "gather", 60, place=[rnd(0, 300), rnd(1, 1200)], acting=[1], steps=20, action_seed=11),
        Scenario("gather_largemap", "gather", 130, place=[rnd(0, 1500), rnd(1, 6000)], acting=[1], steps=8, action_seed=12),
        Scenario("battle_lowhp", "battle", 40, place=[rnd(0, 500), rnd(1, 500)], steps=15, action_seed=13,
                 over={"small": {"hp": 3, "step_recover": -0.4}}),
        Scenario("battle_one_side", "battle", 12, place=[rnd(0, 60), rnd(1, 3)], steps=40, action_seed=14,
                 over={"small": {"damage": 12}}),
    ]
    return {s.name: s for s in S}

"""Shared scenario driver for the parity tests.

One driver, three engines: the product (HIP, magent_amd/lib/libmagent.so), the CPU restatement
(oracle/liboracle.so) and -- when present -- the compiled reference (oracle/_ref/libmagent_ref.so).
All are driven through the same magent_amd.GridWorld wrapper with the same seeds and action streams.
"""
import hashlib
import os
import subprocess
import sys

import numpy as np

import magent_amd
from magent_amd import gridworld as gw  # noqa: F401
from magent_amd.builtin.config import _games

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libmagent_ref.so")
HIP_LIB = os.path.join(ROOT, "magent_amd", "lib", "libmagent.so")
PRODUCT_LIB = HIP_LIB          # (tools/fuzz_parity.py points HIP_LIB at the emulated build when asked for "emu")
EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "_build", "libmagent_emu.so")


def ensure_emu():
    """the engine's HIP sources compiled as plain C++ against tests/hipemu (kernels run lane by lane on the CPU): a checker of
    the kernels' logic for the GPU-less build container, never the product"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build as emu_build
    return emu_build.build()
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def merge_env(*dicts):
    """environment dicts merged; MAGENT_TUNE entries (magent_amd/csrc/tune.h: "key=value,key=value") are joined, not overwritten"""
    out = {}
    for d in dicts:
        for k, v in d.items():
            out[k] = out[k] + "," + v if k == "MAGENT_TUNE" and out.get(k) else v
    return out


def ensure_oracle():
    """build oracle/liboracle.so from its own source if needed (gcc only; no GPU involved)"""
    src = os.path.join(ROOT, "oracle", "gridworld_oracle.cc")
    if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return ORACLE_LIB


def have_ref():
    return os.path.exists(REF_LIB)


_WORLDS = {}


def is_emu(lib):
    return lib is not None and os.path.abspath(lib) == os.path.abspath(EMU_LIB)


def torch_device(env, lib):
    """where the caller-owned "device" tensors of the *_device API live: the GPU -- or, for the emulated library, host memory
    (its HBM is the host heap, a CPU tensor's data_ptr() is a valid device pointer there)"""
    import torch
    return torch.device("cpu") if is_emu(lib) else torch.device("cuda", env.device_id)


def device_sync(lib):
    import torch
    if not is_emu(lib):
        torch.cuda.synchronize()


def world_on(lib):
    """the GridWorld wrapper bound to engine library `lib` (None / HIP_LIB: the product class itself).  The product has no
    switch for this: the test suite subclasses it and overrides the class attribute that names the library."""
    if lib is None or os.path.abspath(lib) == os.path.abspath(PRODUCT_LIB):
        return magent_amd.GridWorld
    key = os.path.abspath(lib)
    if key not in _WORLDS:
        _WORLDS[key] = type("CheckerWorld", (magent_amd.GridWorld,), {"_engine_path": key})
    return _WORLDS[key]


def gridworld(config, lib=None, **kwargs):
    return world_on(lib)(config, **kwargs)


# ---------------------------------------------------------------------------------------------- scenarios
def config_for(game, map_size, **over):
    """built-in game, optionally with agent-type overrides {type_name: {attr: value}} for edge-case scenarios"""
    cfg = _games.make(game, map_size)
    for tname, attrs in over.items():
        cfg.agent_type_dict[tname].update(attrs)
    return cfg


def custom_tri(map_w, map_h):
    """three groups of three different types, every supported rule shape (subject / object receivers, kill, collide),
    in-group attack, kill_supply, a non-square map"""
    cfg = gw.Config()
    cfg.set({"map_width": map_w, "map_height": map_h, "minimap_mode": True, "embedding_size": 4})
    a = cfg.register_agent_type("a", dict(width=1, length=1, hp=6, speed=2, view_range=gw.CircleRange(5), attack_range=gw.CircleRange(1.5),
                                          damage=2, step_recover=0.1, step_reward=-0.01, kill_reward=3, dead_penalty=-0.5, attack_penalty=-0.05))
    b = cfg.register_agent_type("b", dict(width=1, length=1, hp=4, speed=3, view_range=gw.CircleRange(4), attack_range=gw.CircleRange(2),
                                          damage=3, step_recover=0, attack_in_group=1, kill_supply=2.5, kill_reward=1, dead_penalty=-1))
    c = cfg.register_agent_type("c", dict(width=1, length=1, hp=9, speed=1, view_range=gw.CircleRange(2), attack_range=gw.CircleRange(1),
                                          damage=1.5, step_recover=-0.3, kill_supply=4, attack_penalty=-0.2, step_reward=0.25))
    ga, gb, gc = cfg.add_group(a), cfg.add_group(b), cfg.add_group(c)
    sa, sb, sc_ = gw.AgentSymbol(ga, "any"), gw.AgentSymbol(gb, "any"), gw.AgentSymbol(gc, "any")
    cfg.add_reward_rule(gw.Event(sa, "attack", sb), receiver=[sa, sb], value=[0.3, -0.7])
    cfg.add_reward_rule(gw.Event(sb, "kill", sc_), receiver=sb, value=2.25)
    cfg.add_reward_rule(gw.Event(sc_, "collide", sa), receiver=[sa], value=[-0.125])
    cfg.add_reward_rule(gw.Event(sb, "attack", sb), receiver=sb, value=0.0625)      # same symbol twice: never fires
    sb2 = gw.AgentSymbol(gb, "any")
    cfg.add_reward_rule(gw.Event(sb, "attack", sb2), receiver=sb, value=0.03125)    # in-group attack, subject paid
    cfg.add_reward_rule(gw.Event(sb2, "attack", sb), receiver=sb, value=-0.015625)  # in-group attack, object pays
    cfg.add_reward_rule(gw.Event(sc_, "attack", sa), receiver=[sa, sa, sc_], value=[-0.2, 0.05, 0.4])
    return cfg


def custom_chase(map_size):
    """no minimap, no embedding (the pursuit layout with a 1x1 predator): rule rewards land on the object of the event"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size})
    pred = cfg.register_agent_type("predator", dict(width=1, length=1, hp=1, speed=1, view_range=gw.CircleRange(5),
                                                    attack_range=gw.CircleRange(2), attack_penalty=-0.2))
    prey = cfg.register_agent_type("prey", dict(width=1, length=1, hp=1, speed=1.5, view_range=gw.CircleRange(4),
                                                attack_range=gw.CircleRange(0)))
    gp, gq = cfg.add_group(pred), cfg.add_group(prey)
    a, b = gw.AgentSymbol(gp, "any"), gw.AgentSymbol(gq, "any")
    cfg.add_reward_rule(gw.Event(a, "attack", b), receiver=[a, b], value=[1, -1])
    return cfg


def custom_quad(map_size):
    """four groups (the unpacked 8-byte view-cell format; 4 x 8 = 32 attack offsets fill the per-cell hit word)"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size, "minimap_mode": True, "embedding_size": 3})
    t = cfg.register_agent_type("q", dict(width=1, length=1, hp=5, speed=2, view_range=gw.CircleRange(3), attack_range=gw.CircleRange(1.5),
                                          damage=2, step_recover=0.05, step_reward=-0.01, kill_reward=2, dead_penalty=-0.3,
                                          attack_penalty=-0.1))
    gs = [cfg.add_group(t) for _ in range(4)]
    syms = [gw.AgentSymbol(g, "any") for g in gs]
    for i in range(4):
        cfg.add_reward_rule(gw.Event(syms[i], "attack", syms[(i + 1) % 4]), receiver=syms[i], value=0.1 * (i + 1))
    return cfg


def custom_bodies(map_w, map_h):
    """multi-cell bodies of three shapes (3x2, 2x2, 1x1) that attack and block each other; minimap on"""
    cfg = gw.Config()
    cfg.set({"map_width": map_w, "map_height": map_h, "minimap_mode": True, "embedding_size": 5})
    big = cfg.register_agent_type("big", dict(width=3, length=2, hp=8, speed=2, view_range=gw.CircleRange(5), attack_range=gw.CircleRange(2.5),
                                              damage=2, step_recover=0.2, kill_reward=2, dead_penalty=-1, attack_penalty=-0.1))
    mid = cfg.register_agent_type("mid", dict(width=2, length=2, hp=4, speed=1, view_range=gw.CircleRange(4), attack_range=gw.CircleRange(2),
                                              damage=1.5, step_recover=-0.1, kill_supply=1.5, kill_reward=1, attack_in_group=1))
    tiny = cfg.register_agent_type("tiny", dict(width=1, length=1, hp=2, speed=3, view_range=gw.CircleRange(3), attack_range=gw.CircleRange(1),
                                                damage=1, step_reward=0.05, dead_penalty=-2))
    g0, g1, g2 = cfg.add_group(big), cfg.add_group(mid), cfg.add_group(tiny)
    a, b, c = gw.AgentSymbol(g0, "any"), gw.AgentSymbol(g1, "any"), gw.AgentSymbol(g2, "any")
    cfg.add_reward_rule(gw.Event(a, "attack", b), receiver=[a, b], value=[0.5, -0.25])
    cfg.add_reward_rule(gw.Event(b, "kill", c), receiver=b, value=1.5)
    cfg.add_reward_rule(gw.Event(c, "collide", a), receiver=c, value=-0.0625)
    cfg.add_reward_rule(gw.Event(b, "collide", b2 := gw.AgentSymbol(g1, "any")), receiver=b, value=-0.03125)
    return cfg


def custom_trans(map_size):
    """examples/train_trans.py:17-37: ONE group, a type registered without any attack range, a 2:1 map"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size * 2, "map_height": map_size, "minimap_mode": True, "embedding_size": 10})
    agent = cfg.register_agent_type("agent", dict(width=1, length=1, hp=10, speed=1, view_range=gw.CircleRange(6),
                                                  damage=2, step_recover=0.1, step_reward=-1))
    cfg.add_group(agent)
    return cfg


def custom_duo(map_size):
    """Event(a, p, c) & Event(b, p, c) in every supported arrangement: both subjects in one group with different values
    (the per-agent float adds interleave by agent order), the second subject declared first, subjects in two groups,
    the object paid, different predicates for the two subjects, in-group targets"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size, "embedding_size": 3})
    deer = cfg.register_agent_type("deer", dict(width=1, length=1, hp=6, speed=1, view_range=gw.CircleRange(2), attack_range=gw.CircleRange(0),
                                                step_recover=0.2, kill_supply=3, step_reward=0.0625))
    tiger = cfg.register_agent_type("tiger", dict(width=1, length=1, hp=10, speed=1, view_range=gw.CircleRange(4), attack_range=gw.CircleRange(1.5),
                                                  damage=1, step_recover=-0.1, step_reward=-0.03125, attack_penalty=-0.1))
    wolf = cfg.register_agent_type("wolf", dict(width=2, length=1, hp=7, speed=2, view_range=gw.CircleRange(3), attack_range=gw.CircleRange(2),
                                                damage=1.5, attack_in_group=1, kill_reward=1.5))
    gd, gt, gwf = cfg.add_group(deer), cfg.add_group(tiger), cfg.add_group(wolf)
    b0 = gw.AgentSymbol(gt, "any")      # declared before its partner: it is the outer loop of the search
    a, b, c = gw.AgentSymbol(gt, "any"), gw.AgentSymbol(gt, "any"), gw.AgentSymbol(gd, "any")
    w1, w2 = gw.AgentSymbol(gwf, "any"), gw.AgentSymbol(gwf, "any")
    ev = gw.Event
    cfg.add_reward_rule(ev(a, "attack", c) & ev(b, "attack", c), receiver=[a, b], value=[1, 0.3])
    cfg.add_reward_rule(ev(a, "attack", c) & ev(b0, "attack", c), receiver=[a, b0], value=[0.7, -0.15])
    cfg.add_reward_rule(ev(a, "attack", c) & ev(w1, "attack", c), receiver=[w1, a, c], value=[0.45, 0.11, -0.6])
    cfg.add_reward_rule(ev(w1, "kill", c) & ev(a, "attack", c), receiver=[a, w1, a], value=[2.2, 0.9, -0.05])
    cfg.add_reward_rule(ev(w1, "attack", w2) & ev(a, "attack", w2), receiver=[w1, a], value=[0.35, 0.21])
    cfg.add_reward_rule(ev(a, "attack", c) & ev(b, "kill", c), receiver=[b, a], value=[1.3, 0.17])
    cfg.add_reward_rule(ev(w1, "attack", c), receiver=[w1], value=[0.019])
    return cfg


def custom_arrange(map_size, live):
    """examples/train_arrange.py:181-214: goals are agents of a `can_absorb` type; the first mover that bumps into a free goal
    is taken in (it dies, the goal's hp doubles) and is paid by the collide rule.  `live` adds a 2x2 group that attacks goals and movers and bumps into goals
    itself; the goals are observed too (a goal-typed observer leaves taken goals out of its minimap)"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size, "minimap_mode": True, "embedding_size": 12})
    goal = cfg.register_agent_type("goal", {"width": 1, "length": 1, "can_absorb": True})
    agent = cfg.register_agent_type("agent", dict(width=1, length=1, hp=10, speed=2, view_range=gw.CircleRange(6),
                                                  step_recover=-10.0 / 400, step_reward=0))
    g_goal, g_agent = cfg.add_group(goal), cfg.add_group(agent)
    g, a = gw.AgentSymbol(g_goal, "any"), gw.AgentSymbol(g_agent, "any")
    cfg.add_reward_rule(gw.Event(a, "collide", g), receiver=a, value=10)
    if live:
        brute = cfg.register_agent_type("brute", dict(width=2, length=2, hp=6, speed=1, view_range=gw.CircleRange(4),
                                                      attack_range=gw.CircleRange(2), damage=0.75, kill_reward=1, step_reward=-0.01))
        g_brute = cfg.add_group(brute)
        b = gw.AgentSymbol(g_brute, "any")
        cfg.add_reward_rule(gw.Event(b, "collide", g), receiver=[b, g], value=[3, 0.5])
        cfg.add_reward_rule(gw.Event(b, "attack", g), receiver=b, value=0.25)
    return cfg


def custom_rules(map_size):
    """rule expressions beyond Event(a, p, b): die / at / in leaves and and / or / not over them, for searches that iterate a
    single symbol (the partner of a binary event is inferred from the subject's op_obj); one rule ends the game"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size, "minimap_mode": True, "embedding_size": 6})
    t = cfg.register_agent_type("r", dict(width=1, length=1, hp=3, speed=2, view_range=gw.CircleRange(4), attack_range=gw.CircleRange(1.5),
                                          damage=2, step_recover=-0.02, step_reward=-0.01, kill_reward=1, dead_penalty=-0.5, attack_penalty=-0.05))
    big = cfg.register_agent_type("b", dict(width=2, length=2, hp=9, speed=1, view_range=gw.CircleRange(3), attack_range=gw.CircleRange(2),
                                            damage=1.5, attack_in_group=1))
    g0, g1, g2 = cfg.add_group(t), cfg.add_group(t), cfg.add_group(big)
    a, b, c = gw.AgentSymbol(g0, "any"), gw.AgentSymbol(g1, "any"), gw.AgentSymbol(g2, "any")
    b2, c2 = gw.AgentSymbol(g1, "any"), gw.AgentSymbol(g2, "any")
    ev = gw.Event
    half = map_size // 2
    cfg.add_reward_rule(ev(a, "die"), receiver=a, value=-0.7)
    cfg.add_reward_rule(ev(a, "in", ((1, 1), (half, half))) & ~ev(a, "die"), receiver=a, value=0.11)
    cfg.add_reward_rule(ev(a, "attack", b) & ev(b, "attack", a), receiver=[a, b], value=[0.9, -0.3])         # duel
    cfg.add_reward_rule(ev(b2, "kill", c) | ev(b2, "collide", c), receiver=[b2, c, b2], value=[0.4, -0.2, 0.05])
    cfg.add_reward_rule(~ev(c2, "attack", a) & ev(c2, "in", ((half, 0), (map_size, map_size))), receiver=c2, value=0.021)
    cfg.add_reward_rule(ev(b, "at", (half, half)), receiver=b, value=5, terminal=True)
    cfg.add_reward_rule(ev(a, "kill", c) & ev(c, "die"), receiver=[c, a], value=[-1.5, 2.5])
    return cfg


def custom_search(map_size):
    """rule shapes only the reference's recursive search takes (RewardEngine.cc:373-443) -- on the HIP engine they send ALL rules
    of the game to the host evaluation: two iterated symbols (O(N^2) bindings), 'all' symbols as subject and as receiver (group
    reward), a fixed-index symbol (with the reference's Agent::index quirk: 0 until the agent has been through a clear_dead),
    in_a_line, subject and object paid in one group"""
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size, "minimap_mode": True, "embedding_size": 4})
    t = cfg.register_agent_type("s", dict(width=1, length=1, hp=3, speed=2, view_range=gw.CircleRange(4), attack_range=gw.CircleRange(1.5),
                                          damage=2, step_recover=-0.02, step_reward=-0.01, kill_reward=1, dead_penalty=-0.5, attack_penalty=-0.05,
                                          attack_in_group=1))
    post = cfg.register_agent_type("post", dict(width=1, length=1, hp=2, speed=0, view_range=gw.CircleRange(2), attack_range=gw.CircleRange(0)))
    g0, g1, g2 = cfg.add_group(t), cfg.add_group(t), cfg.add_group(post)
    ev, sym = gw.Event, gw.AgentSymbol
    a, b, c, d = sym(g0, "any"), sym(g1, "any"), sym(g0, "any"), sym(g1, "any")
    cfg.add_reward_rule(ev(a, "attack", b) & ev(c, "attack", d), receiver=[a, d], value=[0.0625, -0.03125])          # two subjects, two objects
    e, f = sym(g0, "any"), sym(g1, "any")
    cfg.add_reward_rule(ev(e, "attack", f) & ev(d, "die"), receiver=[e, d, f], value=[0.5, -0.25, 0.125])          # a binding per (attacker, dead agent)
    all1, all2 = sym(g1, "all"), sym(g2, "all")
    cfg.add_reward_rule(ev(all1, "in", ((0, 0), (map_size, map_size))), receiver=all1, value=0.015625)              # group reward
    first = sym(g0, 0)
    cfg.add_reward_rule(ev(first, "attack", b), receiver=[first, b], value=[1.5, -0.75])                             # a fixed agent
    third = sym(g1, 3)
    cfg.add_reward_rule(ev(a, "attack", third), receiver=[a, third], value=[0.3, -0.7])                              # ... as the object
    cfg.add_reward_rule(ev(all2, "in_a_line"), receiver=[all2], value=[0.2])
    x, y = sym(g0, "any"), sym(g0, "any")
    cfg.add_reward_rule(ev(x, "attack", y), receiver=[x, y], value=[0.11, -0.13])                                    # subject and object in one group
    cfg.add_reward_rule(ev(a, "kill", b) & ev(c, "kill", d), receiver=[a, c], value=[2, 3], terminal=True)
    return cfg


def custom_sector(map_w, map_h):
    """sector ranges (Range.h:104-144): the view is a wedge in front of the agent (a window that does not contain the agent itself,
    wider than high or higher than wide), attacks reach forward only; a 2x2 body among them.  With turn_mode the wedge turns with
    the agent; without it everybody looks north"""
    cfg = gw.Config()
    cfg.set({"map_width": map_w, "map_height": map_h, "minimap_mode": True, "embedding_size": 6})
    scout = cfg.register_agent_type("scout", dict(width=1, length=1, hp=5, speed=2, view_range=gw.SectorRange(7, 100), attack_range=gw.SectorRange(2.5, 120),
                                                  damage=2, step_recover=0.1, step_reward=-0.01, kill_reward=2, dead_penalty=-0.3, attack_penalty=-0.05))
    guard = cfg.register_agent_type("guard", dict(width=2, length=2, hp=9, speed=1, view_range=gw.SectorRange(4, 60), attack_range=gw.CircleRange(2),
                                                  damage=1.5, step_recover=-0.05, kill_supply=1, attack_in_group=1))
    lurker = cfg.register_agent_type("lurker", dict(width=1, length=1, hp=3, speed=1, view_range=gw.CircleRange(3), attack_range=gw.SectorRange(3, 40),
                                                    damage=3, dead_penalty=-1))
    g0, g1, g2 = cfg.add_group(scout), cfg.add_group(guard), cfg.add_group(lurker)
    a, b, c = gw.AgentSymbol(g0, "any"), gw.AgentSymbol(g1, "any"), gw.AgentSymbol(g2, "any")
    cfg.add_reward_rule(gw.Event(a, "attack", b), receiver=[a, b], value=[0.25, -0.125])
    cfg.add_reward_rule(gw.Event(c, "kill", a), receiver=c, value=1.5)
    return cfg


CUSTOM = {"sector": custom_sector, "search": custom_search, "rules": custom_rules, "arrange": custom_arrange, "duo": custom_duo, "trans": custom_trans, "tri": custom_tri, "chase": custom_chase, "quad": custom_quad, "bodies": custom_bodies}


class Scenario(object):
    """a reproducible episode: game + placement recipe + random action stream

    game     : built-in game name, or ("custom-name", args...) for one of the CUSTOM configs
    events   : {step: [("add", group, method, kw) | ("walls", method, kw) | ("reset", placements)]} applied before
               that step's observations -- mid-episode placement and a second episode on the same engine (the RNG
               stream continues across reset, GridWorld.cc:72-118)"""

    def __init__(self, name, game, map_size, seed=12345, place=(), steps=10, action_seed=0, walls=0,
                 acting=None, over=None, clear_every=1, obs_every=1, events=None, settings=None, engine=True, still=()):
        self.name, self.game, self.map_size, self.seed = name, game, map_size, seed
        self.engine = engine                # False: the engine refuses this game (turn_mode); the oracle is pinned on it all the same
        self.settings = settings or {}      # extra GridWorld settings (food_mode, ...) and, for custom games, type overrides
        self.place, self.steps, self.action_seed, self.walls = list(place), steps, action_seed, walls
        self.acting, self.over, self.clear_every, self.obs_every = acting, over or {}, clear_every, obs_every
        self.events = events or {}
        self.still = tuple(still)           # groups that are given actions but told to stand still: every move drawn becomes the zero move

    def draw(self, rs, env, g, h, n):
        """this step's actions of group g: uniform over its action space (one draw from `rs`, whatever `still` says)"""
        a = rs.randint(env.get_action_space(h)[0], size=n).astype(np.int32)
        if g in self.still:
            n_move = env.get_view2attack(h)[0] - (2 if self.settings.get("turn_mode") else 0)
            a[:] = n_move // 2              # (the move tables are circle ranges: the centre, (0, 0), is the middle entry)
        return a

    def config(self):
        if callable(self.game):
            return self.game()
        if isinstance(self.game, tuple):
            cfg = CUSTOM[self.game[0]](*self.game[1:])
            for tname, attrs in self.over.items():
                cfg.agent_type_dict[tname].update(attrs)
        else:
            cfg = config_for(self.game, self.map_size, **self.over)
        if self.settings:
            cfg.set(dict(self.settings))
        return cfg

    def populate(self, env, place, walls):
        if walls:
            env.add_walls(method="random", n=walls)
        handles = env.get_handles()
        for g, method, kw in place:
            env.add_agents(handles[g], method, **kw)

    def build(self, lib):
        env = gridworld(self.config(), lib=lib)
        env.set_seed(self.seed)
        env.reset()
        self.populate(env, self.place, self.walls)
        return env, env.get_handles()

    def apply_events(self, env, step):
        for ev in self.events.get(step, ()):
            if ev[0] == "add":
                env.add_agents(env.get_handles()[ev[1]], ev[2], **ev[3])
            elif ev[0] == "walls":
                env.add_walls(method=ev[1], **ev[2])
            elif ev[0] == "goal":
                env.set_goal(env.get_handles()[ev[1]], "random")
            elif ev[0] == "reset":
                env.reset()
                self.populate(env, ev[1], 0)


def run(sc, lib, record=None, device_io=False, env_out=None):
    """Play scenario `sc` on library `lib`; returns list (one dict per step) of every observable output.

    Order of calls per step follows examples/train_battle.py:61-109: get_observation + set_action per group,
    step, get_reward / get_alive / get_pos / get_num per group, clear_dead.

    device_io (HIP engine only): the call sequence bench.py times -- env_get_observation_device into caller-owned torch
    tensors allocated ONCE for the initial population, env_set_action_device from a device tensor, env_get_reward_device --
    instead of the host-buffer reference ABI; the arrays are brought to the host only to be recorded."""
    env, handles = sc.build(lib)
    if env_out is not None:
        env_out.append(env)        # (the caller wants to look at the engine afterwards: engine_stats)
    rs = np.random.RandomState(sc.action_seed)
    twice = np.random.RandomState(sc.action_seed ^ 0x7A1CE) if os.environ.get("FUZZ_TWICE", "0") == "1" else None
    acting = sc.acting if sc.acting is not None else list(range(len(handles)))
    if device_io:
        import torch
        dev = torch_device(env, lib)
        cap = [env.get_num(h) for h in handles]
        d_view = [torch.empty((cap[g],) + env.get_view_space(h), device=dev) for g, h in enumerate(handles)]
        d_feat = [torch.empty((cap[g],) + env.get_feature_space(h), device=dev) for g, h in enumerate(handles)]
        d_rew = [torch.empty(cap[g], device=dev) for g in range(len(handles))]
    out = []
    for step in range(sc.steps):
        rec = {}
        sc.apply_events(env, step)
        for g, h in enumerate(handles):
            n = env.get_num(h)
            if device_io:
                assert n <= cap[g], "device_io: the population grew beyond the caller's buffers"
                if step % sc.obs_every == 0 and n > 0:
                    env.get_observation_device(h, d_view[g], d_feat[g])
                    env.sync()
                    rec["view%d" % g], rec["feat%d" % g] = d_view[g][:n].cpu().numpy(), d_feat[g][:n].cpu().numpy()
                rec["id%d" % g] = env.get_agent_id(h)
                if g in acting:
                    a = torch.from_numpy(sc.draw(rs, env, g, h, n)).to(dev)
                    device_sync(lib)
                    env.set_action_device(h, a)
                continue
            if step % sc.obs_every == 0 and n > 0:
                view, feat = env.get_observation(h)
                rec["view%d" % g], rec["feat%d" % g] = view.copy(), feat.copy()
            rec["id%d" % g] = env.get_agent_id(h)
            if g in acting:
                env.set_action(h, sc.draw(rs, env, g, h, n))
        if twice is not None and not device_io:      # FUZZ_TWICE=1: some groups are given actions again before the step (GridWorld.cc:403-454 appends)
            for g in [g for g in acting if twice.rand() < 0.4]:
                env.set_action(handles[g], twice.randint(env.get_action_space(handles[g])[0], size=env.get_num(handles[g])).astype(np.int32))
        rec["done"] = np.array([env.step()], dtype=np.int32)
        for g, h in enumerate(handles):
            if device_io:
                env.get_reward_device(h, d_rew[g])
                env.sync()
                rec["reward%d" % g] = d_rew[g][:env.get_num(h)].cpu().numpy()
            else:
                rec["reward%d" % g] = env.get_reward(h)
            rec["alive%d" % g] = env.get_alive(h).astype(np.uint8)
            rec["pos%d" % g] = env.get_pos(h)
            rec["num%d" % g] = np.array([env.get_num(h)], dtype=np.int32)
        if step % 7 == 3:
            rec["global_minimap"] = env.get_global_minimap(5, 6).copy()
        if (step + 1) % sc.clear_every == 0:
            env.clear_dead()
        out.append(rec)
        if record is not None:
            record(step, rec)
        if all(env.get_num(h) == 0 for h in handles) and not any(k > step for k in sc.events):
            break
    return out


def mean_info_trace(sc, lib, steps=6):
    """get_info("mean_info") (GridWorld.cc:765-786, "deprecated" there) of every acting, non-empty group of scenario `sc` at three points of
    every step: between set_action and step (Agent::get_action already shows the new action), behind the step (the dead still counted), behind
    clear_dead.  Only groups that have been given actions are asked: the reference counts an agent that never acted one past the end of a
    heap array."""
    env, handles = sc.build(lib)
    rs = np.random.RandomState(sc.action_seed)
    acting = sc.acting if sc.acting is not None else list(range(len(handles)))
    out = []
    for step in range(min(steps, sc.steps)):
        sc.apply_events(env, step)
        asked = []
        for g, h in enumerate(handles):
            n = env.get_num(h)
            if g in acting:
                env.set_action(h, sc.draw(rs, env, g, h, n))
                if n > 0:
                    asked.append(h)
        out += [env.get_mean_info(h).copy() for h in asked]
        env.step()
        out += [env.get_mean_info(h).copy() for h in asked]
        env.clear_dead()
        out += [env.get_mean_info(h).copy() for h in asked if env.get_num(h) > 0]
    return out


def run_cycle(sc, lib, fused, preset=False):
    """Play `sc` one environment CYCLE at a time (observe + set_action per group, step, rewards, clear_dead) and record what a
    caller of magent_amd.EnvBatch.cycle can see: observations, ids, rewards, done, and the state AFTER clear_dead.

    fused=True : the HIP engine through env_cycle_many (EnvBatch.cycle: two launches per cycle for small worlds), device buffers
    fused=False: any library, the same calls one after the other through the reference API (the CPU checkers take this leg)"""
    assert sc.clear_every == 1
    env, handles = sc.build(lib)
    rs = np.random.RandomState(sc.action_seed)
    acting = sc.acting if sc.acting is not None else list(range(len(handles)))
    if fused:
        import torch
        dev = torch_device(env, lib)
        batch = magent_amd.EnvBatch([env], n_threads=1)
        batch.order_streams = not is_emu(lib)
    out = []
    for step in range(sc.steps):
        rec = {}
        sc.apply_events(env, step)
        nums = [env.get_num(h) for h in handles]
        acts = [sc.draw(rs, env, g, h, nums[g]) if g in acting else None for g, h in enumerate(handles)]
        observe = [step % sc.obs_every == 0 and nums[g] > 0 for g in range(len(handles))]
        for g, h in enumerate(handles):
            rec["id%d" % g] = env.get_agent_id(h)
        if fused:
            views = [torch.empty((nums[g],) + env.get_view_space(h), device=dev) if observe[g] else None for g, h in enumerate(handles)]
            feats = [torch.empty((nums[g],) + env.get_feature_space(h), device=dev) if observe[g] else None for g, h in enumerate(handles)]
            d_acts = [torch.from_numpy(a).to(dev) if a is not None else None for a in acts]
            rews = [torch.empty(nums[g], device=dev) for g in range(len(handles))]
            device_sync(lib)
            done = batch.cycle([views], [feats], [d_acts], [rews])[0]
            env.sync()
            for g in range(len(handles)):
                if observe[g]:
                    rec["view%d" % g], rec["feat%d" % g] = views[g].cpu().numpy(), feats[g].cpu().numpy()
                rec["reward%d" % g] = rews[g].cpu().numpy()
        else:
            assert not (preset and fused)
            for g, h in enumerate(handles):       # (preset: every set_action ahead of every observation)
                if preset and acts[g] is not None:
                    env.set_action(h, acts[g])
            for g, h in enumerate(handles):
                if observe[g]:
                    v, f = env.get_observation(h)
                    rec["view%d" % g], rec["feat%d" % g] = v.copy(), f.copy()
                if not preset and acts[g] is not None:
                    env.set_action(h, acts[g])
            done = env.step()
            for g, h in enumerate(handles):
                rec["reward%d" % g] = env.get_reward(h)
            env.clear_dead()
        rec["done"] = np.array([done], dtype=np.int32)
        for g, h in enumerate(handles):       # the state after clear_dead
            rec["num%d" % g] = np.array([env.get_num(h)], dtype=np.int32)
            rec["pos%d" % g] = env.get_pos(h)
            rec["alive%d" % g] = env.get_alive(h).astype(np.uint8)
            rec["ids_after%d" % g] = env.get_agent_id(h)
        out.append(rec)
        if all(env.get_num(h) == 0 for h in handles) and not any(k > step for k in sc.events):
            break
    return out


def run_cycle_batch(scs, lib, preset=False, envs_out=None):
    """run_cycle(fused=True) for SEVERAL environments of one configuration at once: a single magent_amd.EnvBatch, so that for
    small worlds all of them share one pair of launches per cycle (k_render_batch + k_step_solo_batch).  Returns one trajectory
    per scenario; an environment whose groups are all empty keeps cycling with the others (its trajectory stops there, as
    run_cycle's does).
    preset: the actions are handed over by env_set_action_device BEFORE the call and the cycle is given none (the NULL entries
    include/magent_runtime_api.h documents): the observations then show the new last_action, as the reference's do when
    set_action comes first (GridWorld.cc:386-396) -- the checker's leg is run_cycle(..., preset=True)."""
    import torch
    built = [sc.build(lib) for sc in scs]
    envs, handles = [b[0] for b in built], [b[1] for b in built]
    if envs_out is not None:
        envs_out.extend(envs)
    dev = torch_device(envs[0], lib)
    batch = magent_amd.EnvBatch(envs, n_threads=1)
    batch.order_streams = not is_emu(lib)
    rss = [np.random.RandomState(sc.action_seed) for sc in scs]
    out, live = [[] for _ in scs], [True] * len(scs)
    for step in range(max(sc.steps for sc in scs)):
        recs, views, feats, d_acts, rews, observes = [], [], [], [], [], []
        for k, (sc, env, hs) in enumerate(zip(scs, envs, handles)):
            assert sc.clear_every == 1
            acting = sc.acting if sc.acting is not None else list(range(len(hs)))
            rec = {}
            sc.apply_events(env, step)
            nums = [env.get_num(h) for h in hs]
            acts = [rss[k].randint(env.get_action_space(h)[0], size=nums[g]).astype(np.int32) if g in acting else None for g, h in enumerate(hs)]
            observe = [step % sc.obs_every == 0 and nums[g] > 0 for g in range(len(hs))]
            for g, h in enumerate(hs):
                rec["id%d" % g] = env.get_agent_id(h)
            views.append([torch.empty((nums[g],) + env.get_view_space(h), device=dev) if observe[g] else None for g, h in enumerate(hs)])
            feats.append([torch.empty((nums[g],) + env.get_feature_space(h), device=dev) if observe[g] else None for g, h in enumerate(hs)])
            d_acts.append([torch.from_numpy(a).to(dev) if a is not None else None for a in acts])
            rews.append([torch.empty(nums[g], device=dev) for g in range(len(hs))])
            recs.append(rec); observes.append(observe)
        device_sync(lib)
        if preset:
            for env, hs, per_env in zip(envs, handles, d_acts):
                for h, a in zip(hs, per_env):
                    if a is not None:
                        env.set_action_device(h, a)
            dones = batch.cycle(views, feats, None, rews)
        else:
            dones = batch.cycle(views, feats, d_acts, rews)
        for k, (sc, env, hs) in enumerate(zip(scs, envs, handles)):
            env.sync()
            rec = recs[k]
            for g in range(len(hs)):
                if observes[k][g]:
                    rec["view%d" % g], rec["feat%d" % g] = views[k][g].cpu().numpy(), feats[k][g].cpu().numpy()
                rec["reward%d" % g] = rews[k][g].cpu().numpy()
            rec["done"] = np.array([dones[k]], dtype=np.int32)
            for g, h in enumerate(hs):
                rec["num%d" % g] = np.array([env.get_num(h)], dtype=np.int32)
                rec["pos%d" % g] = env.get_pos(h)
                rec["alive%d" % g] = env.get_alive(h).astype(np.uint8)
                rec["ids_after%d" % g] = env.get_agent_id(h)
            if live[k] and step < sc.steps:
                out[k].append(rec)
                if all(env.get_num(h) == 0 for h in hs) and not any(e > step for e in sc.events):
                    live[k] = False
    return out


def run_hashed(sc, lib, device_io=False, env_out=None):
    """run(sc, lib) for sizes whose trajectories do not fit in memory: every array of every step is reduced to its
    xxh3-128 (10 GB/s on one core; SHA-256 would cost more than the engines) as soon as the step is over.
    Returns [{key: hex digest} per step]."""
    import xxhash
    steps = []

    def keep(step, rec):
        steps.append({k: xxhash.xxh3_128(np.ascontiguousarray(v).reshape(-1).view(np.uint8)).hexdigest()
                      for k, v in rec.items()})
        rec.clear()

    run(sc, lib, record=keep, device_io=device_io, env_out=env_out)
    return steps


def assert_same_hashed(want, got, what=""):
    assert len(want) == len(got), "%s: step counts differ %d vs %d" % (what, len(want), len(got))
    for s, (a, b) in enumerate(zip(want, got)):
        assert sorted(a) == sorted(b), "%s step %d: keys differ" % (what, s)
        for k in sorted(a):
            assert a[k] == b[k], "%s step %d: %s differs" % (what, s, k)


def battle_formation(map_size, gap=3, first_left=1):
    """the placement of examples/train_battle.py:15-40 (`generate_map`) as Scenario.place entries: two squares of agents on
    every other cell, `side` = 2 * int(sqrt(0.04 * map_size^2)) cells wide, `gap` cells either side of the middle column; the
    script swaps leftID / rightID before it places (train_battle.py:21-22), so the FIRST round of a process puts handles[1] on
    the left and adds it first.  map_size 3536 -> 2 x 707^2 = 2 x 499,849 agents (BASELINE config 5's "1M agents")."""
    import math
    side = int(math.sqrt(map_size * map_size * 0.04)) * 2
    ys = np.arange((map_size - side) // 2, (map_size - side) // 2 + side, 2)

    def square(x0):
        xs = np.arange(x0, x0 + side, 2)
        pos = np.zeros((len(xs) * len(ys), 3), dtype=np.int32)
        pos[:, 0], pos[:, 1] = np.repeat(xs, len(ys)), np.tile(ys, len(xs))      # x outer, y inner: the script's loop order
        return pos

    left, right = square(map_size // 2 - gap - side), square(map_size // 2 + gap)
    return [(first_left, "custom", {"pos": left}), (1 - first_left, "custom", {"pos": right})]


def battle_melee(map_size):
    """the two lattices of battle_formation pushed INTO each other (the state a self-play episode reaches once the fronts have met,
    set up directly): one side on the even columns, the other on the odd columns of the same square -- every agent has hostile
    neighbours at distance 1, half of all cells of the square are occupied"""
    place = battle_formation(map_size)
    (g_l, _, kw_l), (g_r, _, kw_r) = place
    shifted = kw_r["pos"].copy()
    shifted[:, 0] += kw_l["pos"][0, 0] + 1 - shifted[0, 0]
    return [(g_l, "custom", {"pos": kw_l["pos"]}), (g_r, "custom", {"pos": shifted})]


def fullsize_scenarios():
    """BASELINE.json's configurations at their stated sizes.  Too large for full trajectories in memory:
    compared through per-step hashes (run_hashed); goldens from the compiled reference in tests/golden/digests_fullsize.json"""
    rnd = lambda g, n: (g, "random", {"n": n})
    S = [
        # C2: battle 200x200, 2x2000 agents placed by add_agents("random"), 40 steps
        Scenario("c2_battle200", "battle", 200, place=[rnd(0, 2000), rnd(1, 2000)], steps=40),
        # C3(i) with hp 4 / damage 3: deaths from the first step on -- kills, dead_penalty, compaction and the shuffle of
        # ~300k attack-list entries compared bit for bit at 2 x 400k
        Scenario("c3_battle1000_deaths", "battle", 1000, place=[rnd(0, 400000), rnd(1, 400000)], steps=6,
                 over={"small": {"hp": 4, "damage": 3}}),
        # C3(i) as bench.py plays it: default hp (10) / damage (2), 72 steps -- the first deaths come after a few steps, then
        # compaction every step; longer than a bench run (25 cycles), and past the two things of the plain pipeline that only change
        # with the length of an episode: the refill of the claim words (every 63 steps) and the fall from two optimistic pairs of
        # death-rank rounds to one (after 64 steps that did not need the second)
        Scenario("c3_battle1000_long", "battle", 1000, place=[rnd(0, 400000), rnd(1, 400000)], steps=72),
        # a whole episode's length of the multi-launch pipeline at a size the reference plays in a minute: battle 300 x 300, 2 x 10,000,
        # hp 4 / damage 3, 200 steps (examples/train_battle.py plays rounds of 550), reinforcements at steps 70 and 130 so that there are
        # kills in every one of the 200 steps -- three windows of claim-word epochs, the carried round stamps, both pair budgets
        Scenario("battle300_long", "battle", 300, place=[rnd(0, 10000), rnd(1, 10000)], steps=200, action_seed=77,
                 over={"small": {"hp": 4, "damage": 3}},
                 events={70: [("add", 0, "random", {"n": 9000}), ("add", 1, "random", {"n": 9000})],
                         130: [("add", 0, "random", {"n": 9000}), ("add", 1, "random", {"n": 9000})]}),
        # C4: gather 500x500 (train_gather.py), 20k food + 100k agents, only the agents act
        Scenario("c4_gather500", "gather", 500, place=[rnd(0, 20000), rnd(1, 100000)], acting=[1], steps=8),
        # whole episodes of the two smaller BASELINE configurations (round 6): config 2 for the 550 steps of a train_battle.py round (the
        # multi-launch pipeline at 4000 agents: nine epoch windows), config 4 until the agents have starved (train_gather.py's world: the
        # population falls from 100k to a few thousand -- the step moves from the pipeline to the one-launch step on the way)
        Scenario("c2_battle200_episode", "battle", 200, place=[rnd(0, 2000), rnd(1, 2000)], steps=550, action_seed=41),
        Scenario("c4_gather500_episode", "gather", 500, place=[rnd(0, 20000), rnd(1, 100000)], acting=[1], steps=150, action_seed=42),
        # the reference's own 1M harness (scripts/test/test_1m.py:62-71): map sqrt(20 N), N/10 walls, N/2 2x2 predators, N/2 prey
        Scenario("test_1m", "pursuit", 4472, walls=100000, place=[rnd(0, 500000), rnd(1, 500000)], steps=2),
        # C5 at the size BASELINE.json names: examples/train_battle.py --map_size 3536, its own formation (2 x 499,849 on every
        # other cell of two squares 6 columns apart; 12.5 M cells, large_map_mode), 8 steps of random actions
        # (round 6: 66 steps instead of 8 -- past the first refill of the claim words' epoch window and the fall to one optimistic pair at
        # this size too; the fronts meet around step 3, kills from step ~8 on)
        Scenario("c5_battle3536_formation", "battle", 3536, place=battle_formation(3536), steps=66, action_seed=31),
        # ... and the same two lattices interleaved, hp 4 / damage 3: a million agents with hostile neighbours on both sides --
        # ~380k attacks per step nearly all of which land, thousands of kills per step, dense move contention
        Scenario("c5_battle3536_melee", "battle", 3536, place=battle_melee(3536), steps=5, action_seed=32,
                 over={"small": {"hp": 4, "damage": 3}}),
    ]
    return {s.name: s for s in S}


def preset_batch_scenarios():
    """two battle worlds between the one-launch step's limit for an environment on its own (1536 agents) and its limit inside a batch
    (16384): env_set_action_device on such a world leaves tile counts (the multi-launch form), and a batch then steps it in one launch"""
    rnd = lambda g, n: (g, "random", {"n": n})
    return [Scenario("preset_a", "battle", 60, place=[rnd(0, 1000), rnd(1, 1000)], steps=6, action_seed=81, over={"small": {"hp": 4, "damage": 3}}),
            Scenario("preset_b", "battle", 60, place=[rnd(0, 900), rnd(1, 1100)], steps=6, action_seed=82, seed=4242, over={"small": {"hp": 4, "damage": 3}})]


def pipe_batch_scenarios():
    """environments of one env_cycle_many call that take all three of its forms: plain worlds beyond the one-launch step (the batched pipeline,
    pipe.hip: deaths from the first step, a group that dies out, reinforcements that change every grid size mid-episode, a non-square
    large_map_mode world), a world small enough for the two-launch cycle, and gather (its two groups look through different windows: it goes alone)"""
    rnd = lambda g, n: (g, "random", {"n": n})
    S = scenarios()
    return [Scenario("pipe_a", "battle", 60, place=[rnd(0, 1000), rnd(1, 1000)], steps=12, action_seed=91, over={"small": {"hp": 4, "damage": 3}}),
            Scenario("pipe_b", "battle", 70, place=[rnd(0, 1500), rnd(1, 700)], steps=12, action_seed=92, seed=99, over={"small": {"hp": 3, "damage": 3, "step_recover": -0.3}},
                     events={4: [("add", 1, "random", {"n": 900})], 8: [("add", 0, "fill", {"pos": (2, 2), "size": (30, 20)})]}),
            Scenario("pipe_c", "battle", 104, place=[rnd(0, 2600), rnd(1, 60)], steps=12, action_seed=93, over={"small": {"damage": 11}}),
            S["battle_brawl"], S["gather"]]


def episode_scenarios():
    """whole episodes at full size -- too long for the driver's suite (minutes of GPU time, an hour and a half of single-threaded reference to
    generate): compared once per round by hand (tools/gpu_golden_check.py --episodes; tests/golden/digests_episode.json; the run is kept under
    profiles/).  `c5_battle3536_episode`: BASELINE config 5's world (examples/train_battle.py --map_size 3536, 2 x 499,849 agents in the script's
    own formation) for the 550 steps of one of the script's rounds (train_battle.py:45-140: `while not done` ... `if step_ct > 550: break`)"""
    rnd = lambda g, n: (g, "random", {"n": n})
    return {"c5_battle3536_episode": Scenario("c5_battle3536_episode", "battle", 3536, place=battle_formation(3536), steps=550, action_seed=31),
            # ... and the bench's own workload (C3(i): battle 1000 x 1000, 2 x 400k placed at random) for the same 550 steps
            "c3_battle1000_episode": Scenario("c3_battle1000_episode", "battle", 1000, place=[rnd(0, 400000), rnd(1, 400000)], steps=550)}


def render_episode(lib, out_dir, steps=6, twice=False):
    """a short battle with the text video dump on: returns {file name: bytes} of what env.render() wrote
    (twice: group 0 is given actions a second time before every other step -- the attack events of the literal loop)"""
    sc = Scenario("render", "battle", 16, place=[(0, "random", {"n": 30}), (1, "random", {"n": 30})], steps=steps, action_seed=4,
                  over={"small": {"hp": 3, "damage": 2}})
    env, handles = sc.build(lib)
    env.set_render_dir(out_dir)
    rs = np.random.RandomState(sc.action_seed)
    for step in range(steps):
        for h in handles:
            env.set_action(h, rs.randint(21, size=env.get_num(h)).astype(np.int32))
        if twice and step % 2 == 1:
            env.set_action(handles[0], rs.randint(21, size=env.get_num(handles[0])).astype(np.int32))
        env.step()
        env.render()                       # before clear_dead, like examples/train_battle.py:88-96
        if step == 2:
            env._get_render_info((0, 15), (0, 15))
        env.clear_dead()
    return {name: open(os.path.join(out_dir, name), "rb").read() for name in sorted(os.listdir(out_dir))}


_ATTACK_COUNT = {0: 0, 1: 4, 1.5: 8, 2: 12, 2.5: 20}


def sector_shape(radius, angle, parity):
    """(width, height, count) of the reference's SectorRange (Range.h:104-144), float32 inputs as the C-ABI passes them"""
    import math
    PI = 3.1415926536
    radius, angle = np.float32(radius), np.float32(angle)
    height = int(float(radius) + 0.5)
    width = int(float(np.float32(2) * radius) * math.sin(float(angle / np.float32(2)) * (PI / 180)) + 0.5)
    if width % 2 != parity:
        width -= 1
    count = 0
    for i in range(height):
        for j in range(max(width, 0)):
            dx, dy = abs(j - (width - 1) / 2.0), abs(height - i)
            if math.sqrt(dx * dx + dy * dy) < float(radius) + 0.2 + 0.00001 and dx / dy < math.tan(float(angle / np.float32(2)) * PI / 180) + 0.00001:
                count += 1
    return width, height, count

TURN_MULTICELL_ON_ENGINE = True       # turn_mode with bodies larger than one cell / goals (the generic turn phase)


def fuzz_scenario(seed):
    """a random game inside the engine's scope: 2-4 groups, random body sizes / ranges / hp / damage / recover /
    kill_supply / in-group attack, random rules (subject and object receivers, attack | kill | collide, pairs of events
    joined by `&`), food_mode, random map
    shape, walls, densities, clear_dead cadence and non-acting groups -- for differential testing"""
    rs = np.random.RandomState(seed)
    G = int(rs.choice([2, 2, 2, 3, 3, 4]))
    w = int(rs.randint(12, 150))
    h = w if rs.rand() < 0.5 else int(rs.randint(12, 150))
    minimap, emb = bool(rs.rand() < 0.6), int(rs.choice([0, 3, 10]))
    food_mode = False
    # FUZZ_TURN=1: turn_mode in 60 % of the games; FUZZ_TURN=2: the same draws, but turn_mode only stays on in games the engine
    # takes today (see Env::reset)
    fuzz_turn = int(os.environ.get("FUZZ_TURN", "0"))
    turn_mode = fuzz_turn >= 1 and bool(rs.rand() < 0.6)
    frac = lambda lo, hi: float(rs.randint(int(lo * 16), int(hi * 16) + 1)) / 16.0
    while True:
        specs = []
        for g in range(G):
            bw, bl = (1, 1) if rs.rand() < 0.65 else (int(rs.randint(1, 4)), int(rs.randint(1, 4)))
            specs.append(dict(width=bw, length=bl, hp=frac(1, 12), speed=float(rs.choice([0, 1, 1, 1.5, 2, 2, 3])),
                              view_range=float(rs.choice([1, 2, 3, 4, 5, 6, 7])), attack_range=float(rs.choice([0, 1, 1, 1.5, 1.5, 2, 2.5])),
                              damage=frac(0, 6), step_recover=float(rs.choice([0, 0, 0.1, 0.25, -0.25, -0.5])),
                              kill_supply=float(rs.choice([0, 0, 0, 2.5, 8])), attack_in_group=int(rs.rand() < 0.3),
                              step_reward=frac(-0.25, 0.25), kill_reward=frac(0, 5), dead_penalty=frac(-2, 0),
                              attack_penalty=frac(-0.5, 0)))
        # FUZZ_SECTOR=1: sector ranges (a wedge in front of the agent) for some views and attacks
        for t in specs:
            t["n_attack"] = _ATTACK_COUNT[t["attack_range"]]
            if os.environ.get("FUZZ_SECTOR", "0") == "1":
                parity = t["width"] % 2
                if rs.rand() < 0.35:
                    r, a = float(rs.choice([3, 4, 5, 6, 7])), float(rs.choice([40, 60, 90, 120, 150]))
                    if sector_shape(r, a, parity)[0] >= 1:
                        t["view_sector"] = (r, a)
                if rs.rand() < 0.35:
                    r, a = float(rs.choice([1, 1.5, 2, 2.5, 3])), float(rs.choice([40, 60, 90, 120, 150]))
                    wd, ht, cnt = sector_shape(r, a, parity)
                    if wd >= 1:
                        t["attack_sector"], t["n_attack"] = (r, a), cnt
        # engine limits: the attack offsets of all groups share a 32-bit word; hit lists are bounded
        if sum(t["n_attack"] for t in specs) > 32:
            continue
        kmax = max(t["width"] * t["length"] * sum(a["n_attack"] for j, a in enumerate(specs)
                                                  if j != i or a["attack_in_group"]) for i, t in enumerate(specs))
        if kmax * (4 if turn_mode else 1) <= 256 and all(max(t["width"], t["length"]) + 2 < min(w, h) for t in specs):
            break
    # FUZZ_PLAIN=1: the same draws, bent into a game the pipeline of plain games takes (Env::reset: plain_world) with rules k_strike evaluates
    # itself -- one-cell bodies, no kill_supply / food / goals / turn_mode, one view window, subject-paying attack | kill rules only --
    # so that batches of them go through env_cycle_many's batched pipeline (pipe.hip)
    fuzz_plain = os.environ.get("FUZZ_PLAIN", "0") == "1"
    if fuzz_plain:
        turn_mode = False
        for t in specs:
            t.update(width=1, length=1, kill_supply=0.0, view_range=specs[0]["view_range"])
            t.pop("view_sector", None)
    rules = []
    for _ in range(int(rs.randint(0, 5))):
        a, b = int(rs.randint(G)), int(rs.randint(G))
        op = str(rs.choice(["attack", "attack", "kill", "collide"]))
        who = str(rs.choice(["s", "o", "so", "os", "ss"]))
        if a == b and "s" in who and "o" in who:
            who = "s"
        rules.append((a, op, b, who, [frac(-1, 1) for _ in who]))
    pair_rules = []
    for _ in range(int(rs.choice([0, 0, 1, 2]))):
        a, b, c = int(rs.randint(G)), int(rs.randint(G)), int(rs.randint(G))
        who = str(rs.choice(["ab", "ba", "a", "b", "abc", "cab", "aab", "c"]))
        if c in (a, b):
            who = who.replace("c", "") or "a"          # the object is paid only from another group (engine scope)
        pair_rules.append((a, b, str(rs.choice(["attack", "attack", "kill"])), str(rs.choice(["attack", "attack", "kill"])), c,
                           who, [frac(-1, 1) for _ in who], bool(rs.rand() < 0.5)))

    # expressions the engine takes: and / or / not over die / at / in / attack / kill / collide whose search iterates ONE symbol x
    # (a second symbol y only as the partner of a binary event with x)
    prog_rules = []
    for _ in range(int(rs.choice([0, 0, 1, 1, 2]))):
        gx, gy = int(rs.randint(G)), int(rs.randint(G))
        two = bool(rs.rand() < 0.6)
        def pleaf(force_binary=False):
            kind = "bin" if force_binary else str(rs.choice(["die", "at", "in", "bin", "bin"] if two else ["die", "at", "in"]))
            who = int(rs.randint(2)) if two else 0
            if kind == "die":
                return ("die", who)
            if kind == "at":
                return ("at", who, (int(rs.randint(1, w - 1)), int(rs.randint(1, h - 1))))
            if kind == "in":
                return ("in", who, ((int(rs.randint(0, w)), int(rs.randint(0, h))), (int(rs.randint(0, w)), int(rs.randint(0, h)))))
            return (str(rs.choice(["attack", "kill", "collide"])), who, 1 - who)
        def ptree(depth):
            if depth == 0 or rs.rand() < 0.4:
                return pleaf()
            op = str(rs.choice(["and", "or", "not"]))
            return (op, ptree(depth - 1)) if op == "not" else (op, ptree(depth - 1), ptree(depth - 1))
        expr = ptree(2)
        def has_binary(e):
            return e[0] in ("attack", "kill", "collide") or (e[0] in ("and", "or", "not") and any(has_binary(c) for c in e[1:]))
        if two and not has_binary(expr):
            expr = (str(rs.choice(["and", "or"])), expr, pleaf(force_binary=True))
        who = str(rs.choice(["x", "x", "y", "xy", "yx", "xx"])) if two else str(rs.choice(["x", "xx"]))
        if two and gx == gy and "x" in who and "y" in who:
            who = "x"
        prog_rules.append((gx, gy, two, expr, who, [frac(-1, 1) for _ in who], bool(rs.rand() < 0.05)))

    # FUZZ_RULES=1 (oracle-vs-reference runs only: the engine refuses these shapes): random expressions over die / at / in and
    # the binary events, joined by & | ~ -- they pin the oracle's literal restatement of the recursive rule search
    tree_rules = []
    fuzz_rules = int(os.environ.get("FUZZ_RULES", "0"))
    if fuzz_rules >= 1:
        # FUZZ_RULES=2 adds what only the recursive search handles: 'all' and fixed-index symbols, in_a_line.  Kept inside what
        # the reference defines: the object of a binary event is never 'all', in_a_line only takes 'all' (it asserts both)
        for _ in range(int(rs.randint(1, 4))):
            n_sym = int(rs.randint(1, 4))
            syms = [int(rs.randint(G)) for _ in range(n_sym)]      # group of each symbol
            kinds = ["any"] * n_sym
            if fuzz_rules >= 2:
                kinds = [str(rs.choice(["any", "any", "any", "any", "all", "idx"])) for _ in range(n_sym)]
                kinds = [int(rs.randint(0, 4)) if k == "idx" else k for k in kinds]
            def leaf():
                a = int(rs.randint(len(syms)))
                kind = str(rs.choice(["die", "at", "in", "bin", "bin"] + (["line"] if kinds[a] == "all" else [])))
                if kind == "line":
                    return ("in_a_line", a)
                if kind == "die":
                    return ("die", a)
                if kind == "at":
                    return ("at", a, (int(rs.randint(1, w - 1)), int(rs.randint(1, h - 1))))
                if kind == "in":
                    return ("in", a, ((int(rs.randint(0, w)), int(rs.randint(0, h))), (int(rs.randint(0, w)), int(rs.randint(0, h)))))
                objs = [k for k in range(len(syms)) if kinds[k] != "all"]
                if not objs:
                    return ("die", a)
                b = int(rs.choice(objs))
                return (str(rs.choice(["attack", "kill", "collide"])), a, b)
            def tree(depth):
                if depth == 0 or rs.rand() < 0.4:
                    return leaf()
                op = str(rs.choice(["and", "or", "not"]))
                return (op, tree(depth - 1)) if op == "not" else (op, tree(depth - 1), tree(depth - 1))
            expr = tree(2)
            used = set()
            def collect(e):
                if e[0] in ("and", "or", "not"):
                    for c in e[1:]:
                        collect(c)
                else:
                    used.add(e[1])
                    if e[0] in ("attack", "kill", "collide"):
                        used.add(e[2])
            collect(expr)
            recv = [k for k in sorted(used) if rs.rand() < 0.7] or [sorted(used)[0]]
            tree_rules.append((syms, kinds, expr, recv, [frac(-1, 1) for _ in recv], bool(rs.rand() < 0.1)))

    if fuzz_plain:
        rules = [(a, "attack" if op == "collide" else op, b, "s" * len(who), vals) for a, op, b, who, vals in rules]
        pair_rules, prog_rules = [], []

    def make():
        cfg = gw.Config()
        cfg.set({"map_width": w, "map_height": h, "minimap_mode": minimap, "embedding_size": emb})
        if food_mode:
            cfg.set({"food_mode": True})
        if turn_mode:
            cfg.set({"turn_mode": True})
        if goal_mode:
            cfg.set({"goal_mode": True})
        names = []
        for g, t in enumerate(specs):
            t = dict(t)
            t["view_range"] = gw.SectorRange(*t["view_sector"]) if "view_sector" in t else gw.CircleRange(t["view_range"])
            t["attack_range"] = gw.SectorRange(*t["attack_sector"]) if "attack_sector" in t else gw.CircleRange(t["attack_range"])
            for k in ("view_sector", "attack_sector", "n_attack"):
                t.pop(k, None)
            names.append(cfg.register_agent_type("t%d" % g, t))
        hs = [cfg.add_group(n) for n in names]
        for a, op, b, who, vals in rules:
            sa, sb = gw.AgentSymbol(hs[a], "any"), gw.AgentSymbol(hs[b], "any")
            cfg.add_reward_rule(gw.Event(sa, op, sb), receiver=[{"s": sa, "o": sb}[c] for c in who], value=vals)
        for a, b, opa, opb, c, who, vals, b_first in pair_rules:
            if b_first:
                sb, sa = gw.AgentSymbol(hs[b], "any"), gw.AgentSymbol(hs[a], "any")
            else:
                sa, sb = gw.AgentSymbol(hs[a], "any"), gw.AgentSymbol(hs[b], "any")
            sc_ = gw.AgentSymbol(hs[c], "any")
            cfg.add_reward_rule(gw.Event(sa, opa, sc_) & gw.Event(sb, opb, sc_),
                                receiver=[{"a": sa, "b": sb, "c": sc_}[k] for k in who], value=vals)
        def build_expr(S, e):
            if e[0] == "and":
                return build_expr(S, e[1]) & build_expr(S, e[2])
            if e[0] == "or":
                return build_expr(S, e[1]) | build_expr(S, e[2])
            if e[0] == "not":
                return ~build_expr(S, e[1])
            if e[0] == "die":
                return gw.Event(S[e[1]], "die")
            if e[0] in ("at", "in"):
                return gw.Event(S[e[1]], e[0], e[2])
            return gw.Event(S[e[1]], e[0], S[e[2]])
        for gx, gy, two, expr, who, vals, terminal in prog_rules:
            S = [gw.AgentSymbol(hs[gx], "any"), gw.AgentSymbol(hs[gy], "any")]
            cfg.add_reward_rule(build_expr(S, expr), receiver=[S["xy".index(c)] for c in who], value=vals, terminal=terminal)
        for syms, kinds, expr, recv, vals, terminal in tree_rules:
            S = [gw.AgentSymbol(hs[g], kinds[q]) for q, g in enumerate(syms)]
            def build(e):
                if e[0] == "and":
                    return build(e[1]) & build(e[2])
                if e[0] == "or":
                    return build(e[1]) | build(e[2])
                if e[0] == "not":
                    return ~build(e[1])
                if e[0] in ("die", "in_a_line"):
                    return gw.Event(S[e[1]], e[0])
                if e[0] in ("at", "in"):
                    return gw.Event(S[e[1]], e[0], e[2])
                return gw.Event(S[e[1]], e[0], S[e[2]])
            cfg.add_reward_rule(build(expr), receiver=[S[k] for k in recv], value=vals, terminal=terminal)
        return cfg

    area = (w - 2) * (h - 2)
    density = float(rs.choice([0.03, 0.1, 0.2, 0.3]))
    if any(t["width"] * t["length"] > 1 for t in specs):
        density = min(density, 0.12)       # rejection-sampled placement of big bodies needs room
    place = []
    for g, t in enumerate(specs):
        n = max(1, int(area * density / G / (t["width"] * t["length"])))
        if rs.rand() < 0.15:
            place.append((g, "fill", {"pos": (int(rs.randint(1, w // 2)), int(rs.randint(1, h // 2))),
                                      "size": (int(rs.randint(2, w // 5 + 3)), int(rs.randint(2, h // 5 + 3)))}))
        place.append((g, "random", {"n": min(n, 4000)}))
    acting = [g for g in range(G) if rs.rand() < 0.85] or [0]
    if rs.rand() < 0.3 and not fuzz_plain:           # food_mode: the killed leave food, attackers eat it
        food_mode = True
        for t in specs:
            t["food_supply"] = float(rs.choice([0, 0.05, 1, 2.5, 6]))
            t["eat_ability"] = float(rs.choice([0, 0.5, 1, 3]))
    if rs.rand() < 0.15 and not fuzz_plain:          # one group of goals (can_absorb); goals are never given actions (engine scope)
        goal = int(rs.randint(G))
        specs[goal]["can_absorb"] = True
        if os.environ.get("FUZZ_GOALS_ACT", "0") != "1":     # (=1: the goals are given actions like everybody -- they move: the literal loop)
            acting = [g for g in acting if g != goal]
    if fuzz_turn == 2 and turn_mode and not TURN_MULTICELL_ON_ENGINE and any(t["width"] * t["length"] > 1 or t.get("can_absorb") for t in specs):
        turn_mode = False
    # FUZZ_GOAL=1: goal_mode in half of the games and set_goal calls between steps (a generator of their own: the games of a seed stay
    # what they are without it)
    goal_mode, events = False, {}
    if os.environ.get("FUZZ_GOAL", "0") == "1":
        rg = np.random.RandomState(seed ^ 0x60A1)
        goal_mode = bool(rg.rand() < 0.5)
        for _ in range(int(rg.randint(1, 4))):
            events.setdefault(int(rg.randint(0, 8)), []).append(("goal", int(rg.randint(G))))
    return Scenario("fuzz%d" % seed, make, 0, seed=int(rs.randint(1, 1 << 20)), place=place, steps=int(rs.randint(6, 14)),
                    action_seed=seed, walls=int(area * float(rs.choice([0, 0, 0.02, 0.08]))), acting=acting,
                    clear_every=int(rs.choice([1, 1, 1, 2])), obs_every=int(rs.choice([1, 1, 2])), events=events)


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
        # a long episode: the plain pipeline's claim words carry an epoch that wraps every 63 steps (step.hip: claim_word)
        Scenario("battle_epochs", "battle", 26, place=[rnd(0, 90), rnd(1, 90)], steps=140, action_seed=61, over={"small": {"hp": 6}},
                 events={40: [("add", 0, "random", {"n": 50}), ("add", 1, "random", {"n": 50})],
                         85: [("add", 0, "random", {"n": 60}), ("add", 1, "random", {"n": 60})], 120: [("add", 1, "random", {"n": 40})]}),
        # goal_mode: two feature slots that nothing writes (GridWorld.cc:926-934); set_goal: two draws of the engine's generator per agent,
        # with or without goal_mode (GridWorld.cc:667-679) -- seen in the shuffles and the placements that follow
        Scenario("battle_goal_mode", "battle", 28, place=[rnd(0, 110), rnd(1, 110)], steps=16, action_seed=62, settings={"goal_mode": True},
                 over={"small": {"hp": 4, "damage": 3}}, clear_every=2,
                 events={0: [("goal", 0)], 5: [("goal", 1), ("add", 0, "random", {"n": 30})], 9: [("goal", 0), ("goal", 1)]}),
        Scenario("pursuit_goals_drawn", "pursuit", 30, walls=20, place=[rnd(0, 40), rnd(1, 80)], steps=14, action_seed=63,
                 events={3: [("goal", 1)], 4: [("add", 1, "random", {"n": 25})], 8: [("goal", 0)]}),
        # goals that are given actions: Map::do_move treats a goal that has taken nobody in like any mover (Map.cc:313-358; one that has
        # stands still, GridWorld.cc:580) -- the parallel move resolution rests on goals that stand still, so these steps run the
        # reference's own loops on one lane of the device (k_step_serial)
        Scenario("arrange_goals_move", ("arrange", 36, True), 0, place=[rnd(0, 150), rnd(1, 250), rnd(2, 30)], walls=40, steps=25, action_seed=64),
        Scenario("arrange_goals_move_turn", ("arrange", 40, True), 0, place=[rnd(0, 160), rnd(1, 260), rnd(2, 30)], walls=30, steps=20, action_seed=65,
                 settings={"turn_mode": True}, clear_every=2),
        # ... and goals that are given actions every step but told to stay where they are (the zero move): an ordinary step of the parallel
        # phases -- Env::set_action_device looks at the actions before it sends a step through the literal loop
        Scenario("arrange_goals_stand", ("arrange", 36, True), 0, place=[rnd(0, 150), rnd(1, 250), rnd(2, 30)], walls=40, steps=25, action_seed=66, still=(0,)),
        Scenario("arrange_goals_stand_turn", ("arrange", 40, True), 0, place=[rnd(0, 160), rnd(1, 260), rnd(2, 30)], walls=30, steps=20, action_seed=67,
                 settings={"turn_mode": True}, clear_every=2, still=(0,)),
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
        Scenario("battle_brawl_dense_big", "battle", 300, place=[rnd(0, 40000), rnd(1, 40000)], steps=6, action_seed=21,
                 obs_every=3, over={"small": {"hp": 4, "damage": 3, "step_recover": 0}}),
        Scenario("forest", "forest", 50, place=[rnd(0, 500), rnd(1, 250)], walls=60, steps=25, action_seed=15),
        Scenario("tri_rect", ("tri", 70, 45), 0, place=[rnd(0, 500), rnd(1, 400), rnd(2, 450)], walls=80, steps=25, action_seed=16),
        Scenario("tri_rect_large", ("tri", 150, 101), 0, place=[rnd(0, 3000), rnd(1, 3000), rnd(2, 2500)], steps=10, action_seed=17),
        Scenario("quad", ("quad", 36), 0, place=[rnd(0, 150), rnd(1, 150), rnd(2, 150), rnd(3, 150)], steps=20, action_seed=22),
        Scenario("pursuit", "pursuit", 40, walls=48, place=[rnd(0, 20), rnd(1, 40)], steps=40, action_seed=23),
        Scenario("pursuit_dense", "pursuit", 30, walls=30, place=[rnd(0, 60), rnd(1, 120)], steps=30, action_seed=24),
        Scenario("pursuit_large", "pursuit", 140, walls=600, place=[rnd(0, 1500), rnd(1, 3000)], steps=10, action_seed=25),
        Scenario("bodies", ("bodies", 48, 37), 0, walls=40, place=[rnd(0, 60), rnd(1, 90), rnd(2, 150)], steps=30, action_seed=26),
        Scenario("bodies_large", ("bodies", 130, 111), 0, walls=300, place=[rnd(0, 500), rnd(1, 900), rnd(2, 1500),
                 (0, "fill", {"pos": (100, 80), "size": (12, 10)})], steps=10, action_seed=27),
        Scenario("trans", ("trans", 40), 0, walls=150, place=[rnd(0, 300), (0, "fill", {"pos": (60, 10), "size": (4, 12)})],
                 steps=15, action_seed=29),
        Scenario("chase", ("chase", 40), 0, place=[rnd(0, 150), rnd(1, 300)], walls=40, steps=20, action_seed=18),
        Scenario("battle_events", "battle", 45, place=[rnd(0, 300), rnd(1, 300)], steps=24, action_seed=19,
                 over={"small": {"hp": 4, "damage": 3}},
                 events={5: [("add", 0, "random", {"n": 80}), ("walls", "random", {"n": 30})],
                         9: [("add", 1, "custom", {"pos": [(1, 1), (2, 2), (2, 2), (43, 43), (0, 0)]}),
                             ("walls", "fill", {"pos": (20, 20), "size": (3, 2)})],
                         14: [("reset", [(0, "random", {"n": 200}), (1, "fill", {"pos": (5, 5), "size": (10, 12)})])]}),
        Scenario("battle_grow", "battle", 70, place=[rnd(0, 50), rnd(1, 50)], steps=12, action_seed=28,
                 over={"small": {"hp": 4, "damage": 3}},
                 events={3: [("add", 0, "random", {"n": 1800})], 6: [("add", 1, "random", {"n": 1500})],
                         9: [("add", 0, "fill", {"pos": (2, 2), "size": (6, 60)})]}),
        Scenario("double_attack", "double_attack", 30, place=[rnd(0, 120), rnd(1, 200)], walls=20, steps=25, action_seed=31),
        Scenario("duo", ("duo", 34), 0, place=[rnd(0, 150), rnd(1, 260), rnd(2, 60)], walls=30, steps=25, action_seed=32),
        Scenario("duo_dense", ("duo", 26), 0, place=[rnd(0, 80), rnd(1, 300), rnd(2, 30)], steps=20, action_seed=34),
        Scenario("duo_large", ("duo", 120), 0, place=[rnd(0, 2500), rnd(1, 3500), rnd(2, 800)], steps=8, action_seed=33),
        Scenario("arrange", ("arrange", 40, False), 0, place=[rnd(0, 180), rnd(1, 300)], walls=60, acting=[1], steps=30, action_seed=35),
        Scenario("arrange_live", ("arrange", 44, True), 0, place=[rnd(0, 220), rnd(1, 350), rnd(2, 40)], walls=40, steps=30,
                 acting=[1, 2], action_seed=36, clear_every=2),
        Scenario("arrange_large", ("arrange", 125, True), 0, place=[rnd(0, 2500), rnd(1, 3500), rnd(2, 300)], steps=10,
                 acting=[1, 2], action_seed=37),
        Scenario("battle_food", "battle", 24, place=[rnd(0, 160), rnd(1, 160)], steps=30, action_seed=38, settings={"food_mode": True},
                 over={"small": {"hp": 4, "damage": 3, "step_recover": -0.05, "food_supply": 2.5, "eat_ability": 1}}),
        Scenario("bodies_food", ("bodies", 48, 37), 0, walls=40, place=[rnd(0, 60), rnd(1, 90), rnd(2, 150)], steps=30, action_seed=39,
                 settings={"food_mode": True},
                 over={"big": {"food_supply": 6, "eat_ability": 2}, "mid": {"food_supply": 0.05, "eat_ability": 0.5}, "tiny": {"food_supply": 1, "eat_ability": 3}}),
        Scenario("rules_mix", ("rules", 30), 0, place=[rnd(0, 140), rnd(1, 140), rnd(2, 25)], walls=20, steps=30, action_seed=40),
        Scenario("rules_mix_large", ("rules", 110), 0, place=[rnd(0, 2500), rnd(1, 2500), rnd(2, 300)], steps=8, action_seed=41),
        Scenario("sector", ("sector", 44, 37), 0, walls=40, place=[rnd(0, 150), rnd(1, 40), rnd(2, 120)], steps=25, action_seed=52),
        Scenario("sector_turn", ("sector", 50, 41), 0, walls=50, place=[rnd(0, 180), rnd(1, 45), rnd(2, 150)], steps=25, action_seed=53,
                 settings={"turn_mode": True}),
        Scenario("sector_turn_large", ("sector", 130, 105), 0, walls=300, place=[rnd(0, 1800), rnd(1, 350), rnd(2, 1500)], steps=8, action_seed=54,
                 settings={"turn_mode": True}),
        Scenario("rules_search", ("search", 26), 0, place=[rnd(0, 60), rnd(1, 60), (2, "custom", {"pos": [(5, 3), (5, 4), (5, 5), (5, 6)]})],
                 walls=10, steps=30, action_seed=44),
        Scenario("rules_search_grow", ("search", 30), 0, place=[rnd(0, 40), rnd(1, 40), (2, "custom", {"pos": [(8, 20), (9, 20), (10, 20)]})],
                 steps=20, action_seed=45, clear_every=2,
                 events={4: [("add", 0, "random", {"n": 30})], 9: [("add", 1, "random", {"n": 25}), ("add", 2, "custom", {"pos": [(11, 20)]})]}),
        Scenario("battle_turn", "battle", 26, place=[rnd(0, 120), rnd(1, 120), (0, "custom", {"pos": [(1, 1, 0), (3, 1, 1), (1, 3, 2), (3, 3, 3)]})],
                 steps=25, action_seed=42, settings={"turn_mode": True}),
        Scenario("battle_turn_large", "battle", 130, place=[rnd(0, 3500), rnd(1, 3500)], steps=10, action_seed=46, settings={"turn_mode": True},
                 over={"small": {"hp": 4, "damage": 3}}),
        Scenario("gather_turn", "gather", 50, place=[rnd(0, 200), rnd(1, 700)], acting=[1], steps=20, action_seed=47, settings={"turn_mode": True}),
        Scenario("tri_turn", ("tri", 60, 41), 0, place=[rnd(0, 400), rnd(1, 300), rnd(2, 350)], walls=60, steps=25, action_seed=48,
                 settings={"turn_mode": True}),
        Scenario("bodies_turn", ("bodies", 48, 37), 0, walls=40, place=[rnd(0, 50), rnd(1, 80), rnd(2, 150),
                 (0, "fill", {"pos": (30, 20), "size": (8, 9), "dir": 2})], steps=25, action_seed=43, settings={"turn_mode": True}),
        Scenario("pursuit_turn", "pursuit", 36, walls=40, place=[rnd(0, 50), rnd(1, 100)], steps=30, action_seed=49, settings={"turn_mode": True}),
        Scenario("bodies_turn_large", ("bodies", 120, 101), 0, walls=250, place=[rnd(0, 450), rnd(1, 800), rnd(2, 1300)], steps=10, action_seed=50,
                 settings={"turn_mode": True}),
        Scenario("arrange_turn", ("arrange", 40, True), 0, place=[rnd(0, 200), rnd(1, 300), rnd(2, 35)], walls=30, steps=25, acting=[1, 2], action_seed=51,
                 settings={"turn_mode": True}),
        Scenario("battle_one_side", "battle", 12, place=[rnd(0, 60), rnd(1, 3)], steps=40, action_seed=14,
                 over={"small": {"damage": 12}}),
    ]
    return {s.name: s for s in S}

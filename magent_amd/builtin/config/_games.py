"""Built-in game definitions (inputs of the hot path), table-driven.

Values are those of the reference's game files -- they are data the parity tests depend on:
  battle : python/magent/builtin/config/battle.py:6-33
  pursuit: python/magent/builtin/config/pursuit.py:4-33
  gather : examples/train_gather.py:14-43
  forest : python/magent/builtin/config/forest.py:6-33 (no reward rule; deer carry kill_supply)
  double_attack: python/magent/builtin/config/double_attack.py:8-42 (two tigers on one deer: Event & Event)
"""
from ... import gridworld as gw

_GAMES = {
    "battle": dict(
        settings={"minimap_mode": True, "embedding_size": 10},
        types={"small": dict(width=1, length=1, hp=10, speed=2, view_range=6, attack_range=1.5, damage=2,
                             step_recover=0.1, step_reward=-0.005, kill_reward=5, dead_penalty=-0.1,
                             attack_penalty=-0.1)},
        groups=["small", "small"],
        # (subject group, predicate, object group, receivers, values)
        rules=[(0, "attack", 1, "s", [0.2]), (1, "attack", 0, "s", [0.2])],
    ),
    "pursuit": dict(
        settings={},
        types={"predator": dict(width=2, length=2, hp=1, speed=1, view_range=5, attack_range=2, attack_penalty=-0.2),
               "prey": dict(width=1, length=1, hp=1, speed=1.5, view_range=4, attack_range=0)},
        groups=["predator", "prey"],
        rules=[(0, "attack", 1, "so", [1, -1])],
    ),
    "forest": dict(
        settings={"embedding_size": 10},
        types={"deer": dict(width=1, length=1, hp=5, speed=1, view_range=1, attack_range=0, damage=0, step_recover=0.2,
                            food_supply=0, kill_supply=8),
               "tiger": dict(width=1, length=1, hp=10, speed=1, view_range=4, attack_range=1, damage=3, step_recover=-0.5,
                             food_supply=0, kill_supply=0, step_reward=1, attack_penalty=-0.1)},
        groups=["deer", "tiger"],
        rules=[],
    ),
    "double_attack": dict(
        settings={"embedding_size": 10},
        types={"deer": dict(width=1, length=1, hp=5, speed=1, view_range=1, attack_range=0, step_recover=0.2, kill_supply=8),
               "tiger": dict(width=1, length=1, hp=10, speed=1, view_range=4, attack_range=1, damage=1, step_recover=-0.2)},
        groups=["deer", "tiger"],
        rules=[],
        # (subject group a, subject group b, predicate, object group, receivers, values): Event(a, p, c) & Event(b, p, c)
        pair_rules=[(1, 1, "attack", 0, "ab", [1, 1])],
    ),
    "gather": dict(
        settings={"minimap_mode": True},
        types={"agent": dict(width=1, length=1, hp=3, speed=3, view_range=7, attack_range=1, damage=6, step_recover=0,
                             step_reward=-0.01, dead_penalty=-1, attack_penalty=-0.1, attack_in_group=1),
               "food": dict(width=1, length=1, hp=25, speed=0, view_range=1, attack_range=0, kill_reward=5)},
        groups=["food", "agent"],
        rules=[(1, "attack", 0, "s", [0.5])],
    ),
}


def make(game, map_size):
    spec = _GAMES[game]
    cfg = gw.Config()
    cfg.set({"map_width": map_size, "map_height": map_size})
    cfg.set(dict(spec["settings"]))
    for name, attr in spec["types"].items():
        attr = dict(attr)
        attr["view_range"] = gw.CircleRange(attr["view_range"])
        attr["attack_range"] = gw.CircleRange(attr["attack_range"])
        cfg.register_agent_type(name, attr)
    handles = [cfg.add_group(t) for t in spec["groups"]]
    for subj, pred, obj, who, values in spec["rules"]:
        s = gw.AgentSymbol(handles[subj], index="any")
        o = gw.AgentSymbol(handles[obj], index="any")
        cfg.add_reward_rule(gw.Event(s, pred, o), receiver=[{"s": s, "o": o}[c] for c in who], value=list(values))
    for ga, gb, pred, gc, who, values in spec.get("pair_rules", ()):
        a, b = gw.AgentSymbol(handles[ga], index="any"), gw.AgentSymbol(handles[gb], index="any")
        c = gw.AgentSymbol(handles[gc], index="any")
        cfg.add_reward_rule(gw.Event(a, pred, c) & gw.Event(b, pred, c), receiver=[{"a": a, "b": b, "c": c}[k] for k in who],
                            value=list(values))
    return cfg

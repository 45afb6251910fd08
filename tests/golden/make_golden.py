"""Generate the golden vectors under tests/golden/ from the COMPILED REFERENCE (oracle/_ref/libmagent_ref.so).

Run in the build container (where /root/reference exists and `make -C oracle ref` has built the library):

    OMP_NUM_THREADS=1 python tests/golden/make_golden.py

Outputs (committed; the GPU box never sees /root/reference):
  digests.json     SHA-256 of the full trajectory (view, feature, id, reward, alive, pos, num, done at every step)
                   of every scenario in tests/helpers.scenarios()
  kat_<name>.npz   full per-step arrays of the small scenarios, for readable diffs when a digest breaks
  kat_appendix_b.npz  the hand-checked known answers of SURVEY.md Appendix B (static tables + a scripted duel)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ["OMP_NUM_THREADS"] = "1"

import helpers as H  # noqa: E402
import magent_amd  # noqa: E402

FULL = ["battle_tiny", "battle_one_side"]


def appendix_b(lib):
    """scripted duel of SURVEY.md Appendix B: g0 agent 0 attacks its +x neighbour until it dies"""
    env = H.gridworld("battle", lib=lib, map_size=30)
    env.reset()
    h0, h1 = env.get_handles()
    env.add_agents(h0, "custom", pos=[(10, 12), (3, 3)])
    env.add_agents(h1, "custom", pos=[(11, 12), (10, 14)])
    out = {}
    base, table = env.get_view2attack(h0)
    out["attack_base"], out["view2attack"] = np.array([base], np.int32), table
    for s in range(8):
        for g, h in enumerate((h0, h1)):
            v, f = env.get_observation(h)
            out["view%d_s%d" % (g, s)], out["feat%d_s%d" % (g, s)] = v.copy(), f.copy()
        env.set_action(h0, np.array([17, 6][:env.get_num(h0)], dtype=np.int32))
        env.set_action(h1, np.full(env.get_num(h1), 6, dtype=np.int32))
        out["done_s%d" % s] = np.array([env.step()], np.int32)
        for g, h in enumerate((h0, h1)):
            out["reward%d_s%d" % (g, s)] = env.get_reward(h)
            out["alive%d_s%d" % (g, s)] = env.get_alive(h).astype(np.uint8)
            out["pos%d_s%d" % (g, s)] = env.get_pos(h)
        env.clear_dead()
    return out


def main():
    assert H.have_ref(), "build oracle/_ref first: make -C oracle ref"
    digests = {}
    for name, sc in H.scenarios().items():
        traj = H.run(sc, H.REF_LIB)
        digests[name] = {"sha256": H.digest(traj), "steps": len(traj),
                         "final_num": [int(traj[-1][k][0]) for k in sorted(traj[-1]) if k.startswith("num")][:2]}
        if name in FULL:
            flat = {"%s_s%d" % (k, s): v for s, rec in enumerate(traj) for k, v in rec.items()}
            np.savez_compressed(os.path.join(HERE, "kat_%s.npz" % name), **flat)
        print(name, digests[name])
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "kat_appendix_b.npz"), **appendix_b(H.REF_LIB))
    # text video dump of the reference's RenderGenerator for a short episode
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp()
    files = H.render_episode(H.REF_LIB, tmp)
    out = os.path.join(HERE, "render_battle16")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    for name, data in files.items():
        open(os.path.join(out, name), "wb").write(data)
    print("render files:", {k: len(v) for k, v in files.items()})
    tmp = tempfile.mkdtemp()
    files = H.render_episode(H.REF_LIB, tmp, twice=True)
    out = os.path.join(HERE, "render_battle16_twice")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    for name, data in files.items():
        open(os.path.join(out, name), "wb").write(data)
    print("render files (a group given actions twice):", {k: len(v) for k, v in files.items()})


if __name__ == "__main__":
    main()

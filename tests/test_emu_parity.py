"""The engine's HIP kernels, run lane by lane on the CPU (tests/hipemu), against the oracle.

The build container has no GPU, so the `-m gpu` parity suite cannot run here.  tests/hipemu compiles the SAME sources
(magent_amd/csrc/*.hip, unchanged) as plain C++ against a stand-in <hip/hip_runtime.h>: workgroups run one after another,
their threads as fibers that meet at __syncthreads() / wave ballots.  This checks the kernels' logic -- indexing, the
attack / move fixed points, atomics protocols, barrier placement -- before a GPU minute is spent; it is test
infrastructure like oracle/ (only tests/ and tools/fuzz_parity.py load it), it is not a CPU path of the product, and it
does not replace the GPU suite, which runs the hipcc build on the MI355X.

HIPEMU_SCRAMBLE=<seed> runs lanes and workgroups in a pseudo-random order: results must not depend on it.
"""
import os
import subprocess
import sys

import pytest

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# a slice that touches every phase family: dense attack chains, striped moves, multi-cell bodies, rules, goals, turn_mode, food
SLICE = ["battle_small_dense", "battle_brawl", "battle_walls", "gather", "forest", "tri_rect", "pursuit_dense", "bodies", "quad", "battle_goal_mode", "pursuit_goals_drawn", "arrange_goals_move", "arrange_goals_stand"]


@pytest.fixture(scope="module")
def emu():
    return H.ensure_emu()


@pytest.mark.parametrize("name", SLICE)
def test_emulated_kernels_match_oracle(emu, name):
    sc = H.scenarios()[name]
    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, emu), name + " (hipemu)")


def test_emulated_mean_info(emu):
    """get_info("mean_info") of the engine (host arithmetic over the device's arrays; the pending actions of a set_action committed first)
    against the oracle: between set_action and step, behind the step, behind clear_dead -- the one-launch step and the pipeline"""
    import test_oracle
    for name in test_oracle.MEAN_INFO:
        sc = H.scenarios()[name]
        want, got = H.mean_info_trace(sc, H.ensure_oracle()), H.mean_info_trace(sc, emu)
        assert len(want) == len(got) and len(want) > 0, name
        for k, (a, b) in enumerate(zip(want, got)):
            assert a.tobytes() == b.tobytes(), (name, k, a, b)


def test_emulated_kernels_do_not_depend_on_lane_order(emu):
    """the same under two scrambled lane / workgroup orders (a subprocess each: the order is fixed when the library starts)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "for n in ('battle_brawl', 'tri_rect', 'bodies'):\n"
            "    sc = H.scenarios()[n]\n"
            "    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, H.ensure_emu()), n + ' (hipemu, scrambled)')\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    for seed in ("1", "7"):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPEMU_SCRAMBLE=seed, OMP_NUM_THREADS="1"),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]


@pytest.mark.parametrize("knobs", [{"MAGENT_TUNE": "render=1"}, {"MAGENT_TUNE": "render=4,render_sweep=3"},
                                   {"MAGENT_TUNE": "render=4,render_sweep=2,render_su=3,render_depth=3"},
                                   {"MAGENT_TUNE": "render=4,render_sweep=7,render_su=1,render_depth=1"}],
                         ids=["fast", "sweep", "sweep_3strips_depth3", "sweep_1strip_depth1"])
def test_emulated_battle_render_kernels(emu, knobs):
    """the battle-shaped render kernels (k_render_fast: LDS tables + one-step look-ahead; k_render_sweep2: persistent workgroups
    sweeping the output, register ring of requests, several strips per wave) forced onto small worlds with few workgroups, so
    that every workgroup plays many rounds: float32 observations against the oracle, and the bf16-cell form against the rounded
    float32 one; the sweeping kernel also in its 5-channel form (two groups without minimap channels: pursuit, forest, double_attack)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch\n"
            "import helpers as H\n"
            "emu = H.ensure_emu()\n"
            "for n in ('battle_small_dense', 'battle_walls', 'battle_largemap_odd', 'battle_tiny', 'battle_grow', 'gather', 'pursuit_dense', 'forest', 'double_attack'):\n"
            "    sc = H.scenarios()[n]\n"
            "    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, emu), n)\n"
            "env = H.gridworld('battle', lib=emu, map_size=45)\n"
            "env.set_seed(5); env.reset()\n"
            "hs = env.get_handles()\n"
            "for h in hs: env.add_agents(h, 'random', n=500)\n"
            "rs = np.random.RandomState(1)\n"
            "for step in range(3):\n"
            "    for h in hs:\n"
            "        k = env.get_num(h)\n"
            "        view, feat = torch.empty((k, 13, 13, 7)), torch.empty((k, 34))\n"
            "        env.get_observation_device(h, view, feat)\n"
            "        cells, f2 = torch.empty((k, 13, 13, 8), dtype=torch.bfloat16), torch.empty((k, 34))\n"
            "        env.get_observation_device_bf16(h, cells, f2)\n"
            "        assert view.numpy().tobytes() == env.get_observation(h)[0].tobytes()\n"
            "        assert torch.equal(cells[..., :7].contiguous().view(torch.int16), view.to(torch.bfloat16).view(torch.int16))\n"
            "        assert bool((cells[..., 7] == 1).all()) and torch.equal(f2, feat)\n"
            "        env.set_action(h, rs.randint(21, size=k).astype(np.int32))\n"
            "    env.step(); env.clear_dead()\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS="1", **knobs), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]


def test_emulated_render_dumps(emu, tmp_path):
    """env.render()'s text dump from the emulated engine, byte for byte what the reference's RenderGenerator wrote -- also when a group is
    given actions twice (the attack events of k_step_serial)"""
    for twice in (False, True):
        d = tmp_path / ("twice" if twice else "once")
        d.mkdir()
        got = H.render_episode(emu, str(d), twice=twice)
        gold = os.path.join(H.GOLDEN_DIR, "render_battle16_twice" if twice else "render_battle16")
        for name in sorted(os.listdir(gold)):
            assert got[name] == open(os.path.join(gold, name), "rb").read(), (twice, name)


def test_emulated_large_world_drivers(emu):
    """the multi-launch step of large worlds forced onto small ones (MAGENT_TUNE=solo_step=0, block scans from 100 agents on): the
    single-sync driver, its continuation when the optimistic attack / move rounds run out, the minimap made by k_minimap instead of
    clear_dead's own launches; workgroups in scrambled order"""
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "emu = H.ensure_emu()\n"
            "for n in os.environ['EMU_SCENARIOS'].split(','):\n"
            "    sc = H.scenarios()[n]\n"
            "    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, emu), n)\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    base = {"MAGENT_TUNE": "solo_step=0,scan_solo_max=100", "OMP_NUM_THREADS": "1"}
    wide = "battle_brawl,battle60,battle_largemap_odd,gather_largemap,battle_grow,battle_events,tri_rect,bodies,forest"     # (the 80,000-agent brawl: GPU suite)
    plain = "battle_brawl,battle_largemap_odd,gather_largemap,battle_grow"        # (one-cell bodies, large_map_mode among them)
    for extra, names in (({}, wide), ({"MAGENT_TUNE": "attack_pairs=0"}, plain), ({"MAGENT_TUNE": "fold_minimap=0"}, plain),
                         ({"HIPEMU_SCRAMBLE": "3"}, plain + ",battle60"),
                         ({"MAGENT_TUNE": "move_batches=0,attack_pairs=0", "HIPEMU_SCRAMBLE": "11"}, plain)):
        p = subprocess.run([sys.executable, "-c", code], env=H.merge_env(os.environ, {"EMU_SCENARIOS": names}, base, extra), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "ok" in p.stdout, (extra, p.stdout[-1000:] + p.stderr[-3000:])


def test_emulated_fused_step_of_plain_games(emu):
    """the pipeline of plain games (k_plain_rank, k_plain_eval, k_strike, k_plain_commit; step.hip) forced onto small
    worlds (MAGENT_TUNE=solo_step=0), workgroups and lanes in scrambled order: every scenario whose game it takes -- starving occupants
    whose cell is claimed in the same step (battle_lowhp), skipped clear_dead (stale events are paid again: rules not fused,
    battle_no_clear), agents and walls added mid-episode and a second episode (battle_events, battle_grow), 140 steps (the claim
    words' epoch wraps twice, battle_epochs), four groups, a group that never acts, rules that pay the object (chase: not fused) or
    run on the host (rules_search), the run-out continuation of its own rounds -- and that the pipeline really ran"""
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "emu = H.ensure_emu()\n"
            "for n in os.environ['EMU_SCENARIOS'].split(','):\n"
            "    sc = H.scenarios()[n]\n"
            "    seen = []\n"
            "    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, emu, env_out=seen), n)\n"
            "    assert seen[0].engine_stats()[7] > 0, (n, seen[0].engine_stats())\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    base = {"MAGENT_TUNE": "solo_step=0,scan_solo_max=100", "OMP_NUM_THREADS": "1"}
    plain = ("battle_small_dense,battle_brawl,battle_brawl_big,battle60,battle_walls,battle_largemap,battle_largemap_odd,battle_fill_full,"
             "battle_no_clear,battle_tiny,gather,gather_largemap,battle_lowhp,quad,trans,chase,battle_events,battle_grow,rules_search,battle_epochs,battle_goal_mode")
    names = [n for n in plain.split(",") if n in H.scenarios()]
    assert len(names) == 21
    big = {"battle_brawl_big", "battle_largemap", "gather_largemap", "battle_fill_full"}      # (thousands of agents lane by lane: once)
    for extra, chosen in (({"HIPEMU_SCRAMBLE": "5"}, names), ({"MAGENT_TUNE": "attack_pairs=0,early_report=0", "HIPEMU_SCRAMBLE": "8"}, [n for n in names if n not in big])):
        p = subprocess.run([sys.executable, "-c", code], env=H.merge_env(os.environ, {"EMU_SCENARIOS": ",".join(chosen)}, base, extra), capture_output=True, text=True,
                           timeout=1500)
        assert p.returncode == 0 and "ok" in p.stdout, (extra, p.stdout[-1000:] + p.stderr[-3000:])


def test_emulated_batch_cycle_of_worlds_given_their_actions_beforehand(emu):
    """env_set_action_device on a world of 1537..16384 agents takes the tiled form (attack counts in spread counters, no
    sequence numbers yet); env_cycle_many over two such environments with NULL action entries then steps each in ONE launch -- which needs
    the numbers written out first (Env::cycle_prepare: k_seq_assign, as Env::step_begin does) and the new last_action in the feature rows"""
    scs = H.preset_batch_scenarios()
    got = H.run_cycle_batch(scs, emu, preset=True)
    for sc, g in zip(scs, got):
        H.assert_same(H.run_cycle(sc, H.ensure_oracle(), fused=False, preset=True), g, sc.name + " (preset actions, batch of 2, hipemu)")


def test_emulated_batched_pipeline(emu):
    """env_cycle_many over worlds beyond the one-launch step: ONE launch per phase of the plain pipeline for all of them (pipe.hip:
    k_pipe_render, _set_action, _draw, _rank, _eval x rounds, _strike, _commit, then _clear + _finish with get_reward folded in), beside a
    small world on the two-launch cycle and one that goes alone in the same call -- every environment against the oracle driven alone
    through the reference call sequence; also with the small world in the pipeline's batch too (batch_pipe_min=1) and one optimistic pair
    of rounds only, and with none at all (attack_pairs=0: every step of every environment is finished by the host, environment by environment)"""
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "emu = H.ensure_emu()\n"
            "scs = H.pipe_batch_scenarios()\n"
            "seen = []\n"
            "got = H.run_cycle_batch(scs, emu, envs_out=seen)\n"
            "for sc, g in zip(scs, got):\n"
            "    H.assert_same(H.run_cycle(sc, H.ensure_oracle(), fused=False), g, sc.name + ' (batched pipeline, hipemu)')\n"
            "stats = [e.pipeline_stats() for e in seen]\n"
            "assert all(s[6] >= 8 for s in stats[:3]) and (stats[3][6] > 0) == any(k in os.environ.get('MAGENT_TUNE', '') for k in ('batch_pipe_min=1', 'attack_pairs=0')) and stats[4][6] == 0, stats\n"
            "print('ok', stats)\n") % (ROOT, os.path.join(ROOT, "tests"))
    for extra in ({}, {"HIPEMU_SCRAMBLE": "7"}, {"MAGENT_TUNE": "attack_pairs=1,batch_pipe_min=1,pipe_sweep=0"},
                  {"MAGENT_TUNE": "attack_pairs=0,pipe_sweep=3", "HIPEMU_SCRAMBLE": "9"}):      # (pipe_sweep: the batch's render -- 0 generic workgroups, N sweeping ones per segment; default ~256 per launch)
        p = subprocess.run([sys.executable, "-c", code], env=H.merge_env(os.environ, {"OMP_NUM_THREADS": "1"}, extra), capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0 and "ok" in p.stdout, (extra, p.stdout[-1500:] + p.stderr[-3000:])


@pytest.mark.parametrize("env,needle", [({"MAGENT_SOLO_STEP": "0"}, "MAGENT_TUNE=solo_step="), ({"MAGENT_RENDER_PAD": "1"}, "has no successor"),
                                        ({"MAGENT_TUNE": "solo_stepp=0"}, "unknown entry")],
                         ids=["removed_variable_with_successor", "removed_variable_without", "unknown_tune_key"])
def test_a_knob_the_engine_does_not_read_aborts(emu, env, needle):
    """csrc/tune.h: a typo in MAGENT_TUNE, or one of the MAGENT_* variables of rounds 1-3 still set by an old script, must not silently
    run the default path -- the first environment of the process aborts with the variable's successor named"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "H.gridworld('battle', lib=H.ensure_emu(), map_size=12)\n"
            "print('constructed')\n") % (ROOT, os.path.join(ROOT, "tests"))
    e = {k: v for k, v in os.environ.items() if not k.startswith("MAGENT_")}
    p = subprocess.run([sys.executable, "-c", code], env=dict(e, **env), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "constructed" not in p.stdout and "magent-amd FATAL" in p.stderr and needle in p.stderr, (p.stdout, p.stderr[-800:])


def test_emulated_render_of_a_map_beyond_the_l2s(emu):
    """a painted map of more than 16 MB under a randomly placed population: the first render of every cycle goes behind a launch that
    streams the map through the caches (Env::observe_device, k_touch) -- it only reads, the observations stay what they are"""
    sc = H.Scenario("battle_2100_scattered", "battle", 2100, place=[(0, "random", {"n": 300}), (1, "random", {"n": 300})], steps=3, action_seed=71)
    H.assert_same(H.run(sc, H.ensure_oracle()), H.run(sc, emu), "battle 2100 x 2100 (hipemu)")

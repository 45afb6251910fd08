"""GPU parity tests at BASELINE.json's stated sizes, and of the alternate step drivers.

The trajectories are too large to hold (3.9 GB of observations per 800k-agent step), so every array of every step is
reduced to a 128-bit hash as soon as the step is over (tests/helpers.run_hashed) and the hash lists are compared:
  HIP engine  ==  tests/golden/digests_fullsize.json   (generated from the COMPILED REFERENCE in the build container)
  HIP engine  ==  the CPU oracle run beside it          (localises a failing step / array when the golden differs)
Bit-exact: hashes of the raw bytes, no tolerance.
"""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

FULL = H.fullsize_scenarios()
with open(os.path.join(H.GOLDEN_DIR, "digests_fullsize.json")) as f:
    GOLD = json.load(f)


def _golden_check(names, tune=None, device_io=False, timeout=1500):
    env = H.merge_env(os.environ, {"OMP_NUM_THREADS": "1"}, {"MAGENT_TUNE": tune} if tune else {})
    cmd = [sys.executable, os.path.join(H.ROOT, "tools", "gpu_golden_check.py")] + (["--device-io"] if device_io else []) + list(names)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == len(names) and all(l["ok"] for l in lines), (tune, out.stdout[-3000:], out.stderr[-2000:])
    return {l["name"]: l for l in lines}


@pytest.mark.parametrize("name", sorted(FULL))
def test_fullsize_matches_reference_digest_and_oracle(name):
    """c2_battle200 (40 steps), c3_battle1000_deaths (2x400k, hp 4 / damage 3, 6 steps: kills, dead_penalty, compaction,
    the ~300k-entry attack shuffle), c4_gather500 (100k agents + 20k food, 8 steps), test_1m (2x500k, 2x2 predators),
    c5_battle3536_formation / _melee (train_battle.py --map_size 3536: 2 x 499,849 agents; host-buffer reference ABI here, the
    device ABI below)"""
    got = H.run_hashed(FULL[name], H.HIP_LIB)
    try:
        H.assert_same_hashed(GOLD[name], got, name + " vs compiled-reference digest")
    except AssertionError:
        # which engine disagrees with the golden?  the oracle run tells a stale golden from an engine bug
        H.assert_same_hashed(H.run_hashed(FULL[name], H.ensure_oracle()), got, name + " vs oracle (golden differs too)")
        raise
    if name in ("c2_battle200", "c4_gather500"):     # cheap enough to run the CPU checkers beside it every time
        H.assert_same_hashed(H.run_hashed(FULL[name], H.ensure_oracle()), got, name + " vs oracle")
        if H.have_ref():
            H.assert_same_hashed(H.run_hashed(FULL[name], H.REF_LIB), got, name + " vs compiled reference")


@pytest.mark.parametrize("name", ["c3_battle1000_deaths", "c3_battle1000_long", "c5_battle3536_formation", "c5_battle3536_melee"])
def test_fullsize_device_abi_matches_reference_digest(name):
    """the call sequence bench.py TIMES (env_get_observation_device into caller-owned tensors sized once, env_set_action_device,
    env_get_reward_device) at 2 x 400k agents, against the digests of the compiled reference: `c3_battle1000_long` is the bench
    workload itself (default hp, 72 steps: three bench runs long, and past the first refill of the plain pipeline's claim words at
    step 63 and the fall from two optimistic pairs of death-rank rounds to one after step 64 -- both asserted from the engine's own
    counters) -- every view, feature row, reward, position, alive flag of every step.  `c5_battle3536_*`: BASELINE config 5's world
    (examples/train_battle.py --map_size 3536: 2 x 499,849 agents in the script's own formation, 12.5 M cells) and the same two
    lattices interleaved (a million agents with hostile neighbours)"""
    r = _golden_check([name], device_io=True, timeout=2400)[name]
    plain_steps, two, one, refills, _, ran_out = r["pipeline_stats"][:6]
    assert plain_steps == r["steps"] == FULL[name].steps, r          # the multi-launch pipeline played every step
    if name == "c3_battle1000_long":
        # (a step whose rounds run out puts the budget back to two pairs for 64 steps: on the MI355X that happens once in this episode,
        # so the fall to one pair is asserted where it does happen -- test_long_episodes_of_the_plain_pipeline)
        assert refills >= 2 and two >= 64 and two + one == 72 and (one >= 1 or ran_out >= 1), r


def test_bf16_cell_observation_at_full_size():
    """env_get_observation_device_bf16 at 2 x 400k: every window cell is the round-to-nearest-even bf16 of the float32 observation's
    channels, zeros up to channel 6, 1.0 in channel 7 (include/magent_policy.h) -- compared on the device, all 2 x 67.6 M cells,
    before and after five steps of play"""
    import numpy as np
    import torch
    sc = FULL["c3_battle1000_long"]
    env, handles = sc.build(H.HIP_LIB)
    dev = torch.device("cuda", env.device_id)
    rs = np.random.RandomState(3)
    for step in range(6):
        for h in handles:
            n = env.get_num(h)
            if step in (0, 5):
                view, feat = env.get_observation_device(h)
                env.sync()
                cells, feat16 = env.get_observation_device_bf16(h)
                env.sync()
                C = view.shape[-1]
                assert cells.dtype == torch.bfloat16 and tuple(cells.shape) == (n,) + tuple(view.shape[1:3]) + (8,)
                assert torch.equal(cells[..., :C].view(torch.int16), view.to(torch.bfloat16).view(torch.int16)), "step %d" % step
                assert bool((cells[..., C:7] == 0).all()) and bool((cells[..., 7] == 1).all())
                assert torch.equal(feat16, feat)
                del view, cells
            env.set_action(h, rs.randint(21, size=n).astype(np.int32))
        env.step()
        env.clear_dead()


# The step has two drivers over the same kernels (engine.hip: Env::step) and, in the single-sync driver, a continuation
# path for the rare step whose optimistic fixed-point rounds run out.  Each variant is forced through the environment
# (read once per process) and must reproduce the oracle on dense scenarios: long attack chains, conga lines of movers,
# multi-cell bodies, goals, three groups.
# (battle_epochs: 140 steps, 180-420 agents -- with solo_step=0 the multi-launch pipeline plays it: the claim words' epoch window
# is refilled twice and the budget of optimistic rounds falls from two pairs to one on the way)
DENSE = ["battle_epochs", "battle_brawl", "battle_brawl_big", "battle_brawl_dense_big", "battle_fill_full", "bodies_large", "tri_rect_large",
         "arrange_live", "battle_food", "battle_turn_large", "bodies_turn", "bodies_turn_large", "arrange_turn", "pursuit_large"]
VARIANTS = {
    "checked_step": {"MAGENT_TUNE": "checked_step=1"},
    "attack_runs_out": {"MAGENT_TUNE": "attack_pairs=0"},
    "move_runs_out": {"MAGENT_TUNE": "move_batches=0"},
    "host_shuffle": {"MAGENT_TUNE": "host_shuffle=1"},
    "multi_launch_step": {"MAGENT_TUNE": "solo_step=0"},
    # the one-launch step up to the limit it has inside a batch (an environment on its own leaves it at 1536 agents since round 5)
    "one_launch_step_to_16384": {"MAGENT_TUNE": "solo_max=16384"},
    "multi_launch_side_stream": {"MAGENT_TUNE": "solo_step=0,overlap=3"},   # set_action and the head of the step beside the renders
    "multi_launch_late_report": {"MAGENT_TUNE": "solo_step=0,early_report=0"},   # the plain pipeline's report behind the moves (default: ahead of them)
    # the battle-shaped render kernels forced on small worlds (defaults: k_render_sweep2 only at scale, k_render_fast only for bf16 cells)
    "render_fast": {"MAGENT_TUNE": "render=1"},
    "render_sweep": {"MAGENT_TUNE": "render=4,render_sweep=5"},
    "render_sweep_3strips": {"MAGENT_TUNE": "render=4,render_sweep=2,render_su=3,render_depth=3"},
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_step_driver_variants(variant):
    env = H.merge_env(os.environ, {"OMP_NUM_THREADS": "1"}, VARIANTS[variant])
    out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "gpu_check.py")] + DENSE, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "failures: 0" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])
    assert out.stdout.count("OK  ") == len(DENSE)


@pytest.mark.parametrize("tune", [None, "attack_pairs=0", "early_report=0", "overlap=3"], ids=["default", "attack_runs_out", "late_report", "side_stream"])
def test_long_episodes_of_the_plain_pipeline(tune):
    """What only acts with the LENGTH of an episode: the plain pipeline's claim words carry a 6-bit epoch and
    are refilled when a window of 63 steps begins (Env::scratch_for), the "inputs changed" round stamps count on from step to step
    (PlainWorld::round_base), and the step driver falls from two optimistic pairs of death-rank rounds to one after 64 steps that did not
    need the second (Env::reset / step_end).  `battle300_long`: battle 300 x 300, 2 x 10,000 agents, hp 4 / damage 3, 200 steps with
    reinforcements at steps 70 and 130 -- kills in every step, three windows -- through the multi-launch pipeline (the default driver at
    this size), every array of every step against the digests of the COMPILED REFERENCE; also with every step's rounds running out
    (the host-checked continuation for 200 steps), with the report behind the moves, and with the head of the step on the side stream.
    The test asserts that the pipeline really played every step, that the window was refilled, and that both pair budgets were used.
    (A refill every 64 steps instead of 63 makes this test fail: tests/README.md, profiles/r06_raw/mutation_refill.txt.)"""
    r = _golden_check(["battle300_long"], tune)["battle300_long"]
    plain_steps, two, one, refills, _, ran_out = r["pipeline_stats"][:6]
    assert r["steps"] == 200 and plain_steps == 200, r
    assert refills >= 4, r                     # the first step, steps 63 / 126 / 189 (the reinforcements' steps too: add_agents keeps the words)
    if tune == "attack_pairs=0":
        assert ran_out == 200, r               # every step was finished by the host-checked driver
    else:
        assert two >= 64 and one >= 1 and two + one == 200, r   # ... the fall from two pairs to one happened


def test_fuzz_rule_search_on_the_host_path():
    """reward rules whose shape the kernels do not take ('all' / fixed-index symbols, several iterated symbols, in_a_line) run
    through the engine's host evaluation (Env::eval_rules_host): random games with random such rules (FUZZ_RULES=2), HIP == oracle"""
    env = dict(os.environ, OMP_NUM_THREADS="1", FUZZ_RULES="2")
    out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "120"], env=env,
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "120 seeds, 0 failures" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])


def test_fuzz_turn_mode():
    """turn_mode (agents face a direction; moves, attack offsets and the observation window live in the agent's frame; bodies larger
    than one cell re-lay their footprint when they turn -- an order-dependent fixed point of its own): random games, 60 % of them with
    turn_mode on (FUZZ_TURN=2), on both step drivers, HIP == oracle"""
    for extra in ({}, {"MAGENT_TUNE": "solo_step=0"}):
        env = H.merge_env(os.environ, {"OMP_NUM_THREADS": "1", "FUZZ_TURN": "2"}, extra)
        out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "150"], env=env,
                             capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0 and "150 seeds, 0 failures" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])


def test_fuzz_goal_mode():
    """goal_mode and set_goal (the deprecated pair of GridWorld.cc:137, :667-679: feature slots nothing writes, draws of the engine's
    generator that every later shuffle and placement sees) in random games (FUZZ_GOAL=1), on both step drivers, HIP == oracle"""
    for extra in ({}, {"MAGENT_TUNE": "solo_step=0"}):
        env = H.merge_env(os.environ, {"OMP_NUM_THREADS": "1", "FUZZ_GOAL": "1"}, extra)
        out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "120"], env=env,
                             capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0 and "120 seeds, 0 failures" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])


def test_fuzz_literal_loop():
    """the reference's own loops on one lane of the device (k_step_serial), in random games of every kind (turn_mode in 60 % of them):
    groups that are given actions twice before a step (FUZZ_TWICE=1), goals that are given actions and move (FUZZ_GOALS_ACT=1); HIP == oracle"""
    for extra in ({"FUZZ_TWICE": "1"}, {"FUZZ_GOALS_ACT": "1"}):
        env = H.merge_env(os.environ, {"OMP_NUM_THREADS": "1", "FUZZ_TURN": "1"}, extra)
        out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "150"], env=env,
                             capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0 and "150 seeds, 0 failures" in out.stdout, (extra, out.stdout[-3000:], out.stderr[-2000:])


def test_fuzz_fused_cycle():
    """random games (40 % of them turn_mode) with the HIP leg driven through env_cycle_many -- a whole environment cycle in two
    launches: set_action, step, rewards, clear_dead and the next minimap inside k_step_solo -- against the oracle driven through the
    reference call sequence"""
    # (with the one-launch step's limit for a single environment raised to the batch's: the two-launch cycle at every size a batch runs it;
    # and with the defaults: worlds beyond 1536 agents fall back to the ordinary launches inside the same call)
    for extra in ({"MAGENT_TUNE": "solo_max=16384"}, {}):
        env = H.merge_env(os.environ, {"OMP_NUM_THREADS": "1", "FUZZ_CYCLE": "1", "FUZZ_TURN": "1"}, extra)
        out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "120"], env=env,
                             capture_output=True, text=True, timeout=1200)
        assert out.returncode == 0 and "120 seeds, 0 failures" in out.stdout, (extra, out.stdout[-3000:], out.stderr[-2000:])


def test_one_launch_cycle_between_the_two_limits():
    """worlds between the limit of the one-launch step for an environment on its own (1536 agents) and its limit inside a batch (16384):
    the two-launch cycle of env_cycle_many with the first limit raised to the second (what a batch of such worlds runs per environment),
    and with the default limits (the same call falls back to the multi-launch pipeline); both against the oracle's call sequence"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, helpers as H\n"
            "for n in ('battle_brawl_big', 'battle_largemap', 'tri_rect_large', 'bodies_large', 'pursuit_large', 'gather_largemap', 'duo_large'):\n"
            "    sc = H.scenarios()[n]\n"
            "    H.assert_same(H.run_cycle(sc, H.ensure_oracle(), fused=False), H.run_cycle(sc, H.HIP_LIB, fused=True), n)\n"
            "print('ok')\n") % (H.ROOT, os.path.join(H.ROOT, "tests"))
    for extra in ({"MAGENT_TUNE": "solo_max=16384"}, {}):
        p = subprocess.run([sys.executable, "-c", code], env=H.merge_env(os.environ, {"OMP_NUM_THREADS": "1"}, extra), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0 and "ok" in p.stdout, (extra, p.stdout[-1000:] + p.stderr[-3000:])


def test_fuzz_batched_pipeline():
    """three environments per random PLAIN game (FUZZ_PLAIN=1: one-cell bodies, subject-paying attack / kill rules, one view window -- what
    the pipeline of plain games takes) in ONE magent_amd.EnvBatch with every world sent through the batched pipeline (batch_pipe_min=1:
    pipe.hip's one launch per phase for all of them), each against the oracle driven alone through the reference call sequence; the second
    leg with one optimistic pair of death-rank rounds (steps that run out are finished by the host, environment by environment)"""
    for tune in ("batch_pipe_min=1", "batch_pipe_min=1,attack_pairs=1"):
        env = dict(os.environ, OMP_NUM_THREADS="1", FUZZ_BATCH="3", FUZZ_PLAIN="1", MAGENT_TUNE=tune)
        out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "150"], env=env,
                             capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0 and "150 seeds, 0 failures" in out.stdout and "batched pipeline (pipe.hip): 150" in out.stdout, (tune, out.stdout[-3000:], out.stderr[-2000:])


@pytest.mark.parametrize("tune", ["pipe_sweep=0", "pipe_sweep=3", "pipe_sweep=32"])
def test_batched_pipeline_render_forms(tune):
    """the batch's render launch in its other forms -- the generic workgroups (pipe_sweep=0: what every shape but the battle one takes) and the
    sweeping kernel at workgroup counts the default (~256 over the launch) does not pick -- on tests/test_gpu_parity.py's batched-pipeline scenarios, every
    environment against the oracle driven alone through the reference call sequence"""
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H\n"
            "scs = H.pipe_batch_scenarios()\n"
            "seen = []\n"
            "got = H.run_cycle_batch(scs, H.HIP_LIB, envs_out=seen)\n"
            "for sc, g in zip(scs, got):\n"
            "    H.assert_same(H.run_cycle(sc, H.ensure_oracle(), fused=False), g, sc.name + ' (batched pipeline)')\n"
            "assert all(e.pipeline_stats()[6] >= 8 and (e.pipeline_stats()[7] > 0) == ('pipe_sweep=0' not in os.environ['MAGENT_TUNE']) for e in seen[:3])\n"
            "print('ok')\n") % (H.ROOT, os.path.join(H.ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=H.merge_env(os.environ, {"OMP_NUM_THREADS": "1"}, {"MAGENT_TUNE": tune}), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ok" in p.stdout, (tune, p.stdout[-1500:] + p.stderr[-3000:])


def test_fuzz_batched_cycle():
    """three environments per random game in ONE magent_amd.EnvBatch -- k_render_batch + k_step_solo_batch: one pair of launches for
    all of them -- each against the oracle driven alone through the reference call sequence"""
    env = dict(os.environ, OMP_NUM_THREADS="1", FUZZ_BATCH="3", FUZZ_TURN="1")
    out = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_parity.py"), "oracle", "hip", "0", "80"], env=env,
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "80 seeds, 0 failures" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])

"""CPU tests of the product's host logic and of the C-ABI surface -- no compute call, no GPU.

The HIP library is cross-compiled by __graft_entry__.build(); loading it and calling the pre-reset entry points
(configuration, agent types, groups, the *_space / view2attack infos) touches no device."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

ROOT = H.ROOT


@pytest.fixture(scope="module")
def hip_lib():
    if not os.path.exists(H.HIP_LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return H.HIP_LIB


def declared_symbols():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("magent_runtime_api.h", "magent_policy.h"))
    return sorted(set(re.findall(r"^int\s+(\w+)\s*\(", text, flags=re.M)))


def test_header_declares_the_reference_abi():
    names = declared_symbols()
    for must in ("env_new_game", "env_delete_game", "env_config_game", "env_reset", "env_get_observation", "env_set_action",
                 "env_step", "env_get_reward", "env_get_info", "env_render", "env_render_next_file",
                 "gridworld_register_agent_type", "gridworld_new_group", "gridworld_add_agents", "gridworld_clear_dead",
                 "gridworld_set_goal", "gridworld_define_agent_symbol", "gridworld_define_event_node",
                 "gridworld_add_reward_rule", "discrete_snake_clear_dead", "discrete_snake_add_object"):
        assert must in names


def test_library_exports_every_declared_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib, mode=os.RTLD_LOCAL)
    for name in declared_symbols():
        assert hasattr(lib, name), "libmagent.so does not export %s" % name


def test_oracle_exports_the_reference_abi():
    lib = ctypes.CDLL(H.ensure_oracle(), mode=os.RTLD_LOCAL)
    from magent_amd import c_lib
    for name, _ in c_lib.REFERENCE_ABI:
        assert hasattr(lib, name)


def test_product_does_not_link_the_oracle(hip_lib):
    out = subprocess.run(["ldd", hip_lib], capture_output=True, text=True).stdout
    assert "oracle" not in out and "magent_ref" not in out
    assert "libamdhip64" in out


@pytest.mark.parametrize("game,size", [("battle", 40), ("gather", 60)])
def test_pre_reset_infos_match_oracle(hip_lib, game, size):
    """spaces, action layout and the view2attack table are host-side tables (Range.h / AgentType.cc restated):
    the product must agree with the oracle without ever touching a device"""
    import magent_amd
    a = H.gridworld(game, lib=hip_lib, map_size=size)
    b = H.gridworld(game, lib=H.ensure_oracle(), map_size=size)
    for ha, hb in zip(a.get_handles(), b.get_handles()):
        assert a.get_view_space(ha) == b.get_view_space(hb)
        assert a.get_feature_space(ha) == b.get_feature_space(hb)
        assert a.get_action_space(ha) == b.get_action_space(hb)
        (ba, ta), (bb, tb) = a.get_view2attack(ha), b.get_view2attack(hb)
        assert ba == bb and np.array_equal(ta, tb)


def test_battle_spaces_known_answers(hip_lib):
    import magent_amd
    env = H.gridworld("battle", lib=hip_lib, map_size=100)
    h = env.get_handles()[0]
    assert env.get_view_space(h) == (13, 13, 7) and env.get_feature_space(h) == (34,) and env.get_action_space(h) == (21,)


def test_unsupported_features_fail_loudly(hip_lib):
    """no silent approximation: an out-of-scope feature aborts the process with a message"""
    code = ("import ctypes, magent_amd\n"
            "from magent_amd import c_lib\n"
            "lib = c_lib.load()\n"
            "game = ctypes.c_void_p()\n"
            "lib.env_new_game(ctypes.byref(game), b'DiscreteSnake')\n")
    env = dict(os.environ, MAGENT_AMD_NO_TORCH="1", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert p.returncode != 0 and "magent-amd FATAL" in p.stderr and "DiscreteSnake" in p.stderr


def test_no_cpu_fallback_without_gpu(hip_lib):
    """on a box without a HIP device env_reset must abort, not fall back to anything"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    code = ("import magent_amd\n"
            "env = magent_amd.GridWorld('battle', map_size=30)\n"
            "env.reset()\n")
    env = dict(os.environ, MAGENT_AMD_NO_TORCH="1", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert p.returncode != 0 and "no HIP device" in p.stderr


def test_reward_rule_serialisation_order():
    """symbols are numbered receivers-first then in expression order, nodes in pre-order (gridworld.py:493-565)"""
    from magent_amd import gridworld as gw
    calls = []

    class Spy(object):
        has_device_api = False

        def __getattr__(self, name):
            def f(*args):
                calls.append((name, args))
                if name == "env_get_info":
                    pass
                return 0
            return f

    cfg = gw.Config()
    cfg.set({"map_width": 10, "map_height": 10})
    t = cfg.register_agent_type("t", {"width": 1, "length": 1, "hp": 1, "speed": 1, "view_range": gw.CircleRange(1),
                                      "attack_range": gw.CircleRange(1)})
    g0, g1 = cfg.add_group(t), cfg.add_group(t)
    a, b = gw.AgentSymbol(g0, "any"), gw.AgentSymbol(g1, "any")
    cfg.add_reward_rule(gw.Event(a, "attack", b), receiver=[a, b], value=[1, -1])
    cfg.add_reward_rule(gw.Event(b, "kill", a), receiver=b, value=2)
    env = gw.GridWorld.__new__(gw.GridWorld)
    env._lib, env.game = Spy(), None
    env._send_reward_rules(cfg)
    syms = [(c[1][1], c[1][2], c[1][3]) for c in calls if c[0] == "gridworld_define_agent_symbol"]
    assert syms == [(0, g0, -1), (1, g1, -1)]
    nodes = [(c[1][1], c[1][2]) for c in calls if c[0] == "gridworld_define_event_node"]
    assert nodes == [(0, gw.EventNode.OP_ATTACK), (1, gw.EventNode.OP_KILL)]
    rules = [c[1][1] for c in calls if c[0] == "gridworld_add_reward_rule"]
    assert rules == [0, 1]


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/magent"), reason="reference python package not present")
def test_reference_wrapper_binds_to_the_new_library(hip_lib, tmp_path):
    """the UNMODIFIED reference wrapper finds every symbol it calls in the new library.
    (Only pre-reset calls here: this container has no GPU.)"""
    (tmp_path / "build").mkdir()
    os.symlink(hip_lib, tmp_path / "build" / "libmagent.so")
    import shutil
    shutil.copytree("/root/reference/python", tmp_path / "python")   # scratch copy: c_lib.py resolves ../../build physically
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import magent\n"
            "env = magent.GridWorld('battle', map_size=50)\n"
            "h = env.get_handles()[0]\n"
            "t = lambda v: tuple(int(i) for i in v)\n"
            "print(t(env.get_view_space(h)), t(env.get_feature_space(h)), t(env.get_action_space(h)), env.get_view2attack(h)[0])\n"
            % str(tmp_path / "python"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip() == "(13, 13, 7) (34,) (21,) 13"


def test_font_provider_and_random_actor(tmp_path):
    """small callers of the path: the 8x8 font of the arrange game's goal layouts, the random policy on numpy observations"""
    from magent_amd.utility import FontProvider
    from magent_amd.builtin.rule_model import RandomActor
    path = tmp_path / "font.txt"
    path.write_text("0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80\n" + "\n".join(["0x00, 0, 0, 0, 0, 0, 0, 0xFF"] * 2) + "\n")
    font = FontProvider(str(path))
    assert len(font.data) == 3 and [row.index(1) for row in font.get(0)] == list(range(8))     # bit j of row i = pixel (i, j)
    assert font.get(chr(1))[7] == [1] * 8 and font.get(1)[0] == [0] * 8

    class Env(object):
        def get_action_space(self, handle):
            return (9,)
    a, b = RandomActor(Env(), 0, seed=3), RandomActor(Env(), 0, seed=3)
    obs = (np.zeros((50, 3, 3, 2), np.float32), np.zeros((50, 4), np.float32))
    x, y = a.infer_action(obs), b.infer_action(obs)
    assert x.dtype == np.int32 and x.shape == (50,) and x.min() >= 0 and x.max() < 9 and np.array_equal(x, y)


def test_no_kernel_keeps_the_world_in_scratch(hip_lib):
    """the compiler's resource report of the build (magent_amd/lib/kernel_resources_<translation unit>.txt): indexing the by-value world description
    with a per-lane value makes it keep a private copy in scratch memory, 2 KB per lane -- a 15x slowdown that no parity test
    sees.  Every kernel stays within a few bytes of scratch, and the render kernel within its register budget."""
    import re
    text = ""
    for tu in ("render", "step", "cycle"):
        path = os.path.join(os.path.dirname(hip_lib), "kernel_resources_%s.txt" % tu)
        assert os.path.exists(path), "run __graft_entry__.build()"
        text += open(path).read()
    names = re.findall(r"Function Name: (\S+)", text)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
    vgprs = [int(x) for x in re.findall(r"remark:\s+VGPRs: (\d+)", text)]
    assert len(names) == len(scratch) == len(vgprs) and len(names) > 60
    for n, sc in zip(names, scratch):
        if "k_step_serial" in n:      # one lane walking the reference's sequential loops (repeated set_action): off the hot path by design
            continue
        assert sc <= 16, (n, sc)
    sweep = [v for n, v in zip(names, vgprs) if "k_render_sweep2ILb0ELi2ELi2E" in n]
    assert sweep and max(sweep) <= 256, sweep         # one 4-wave workgroup per CU (1 wave per SIMD: 512 registers to spare); no scratch above
    render = [v for n, v in zip(names, vgprs) if "k_renderILb1ELb1ELi1ELb1ELb0" in n]
    assert render and max(render) <= 96, render      # 5 waves per SIMD (the measured sweet spot of the store-bound kernel)


def test_committed_bench_line_keeps_the_contract():
    """profiles/r04_bench_line.json is the line `python bench.py` printed at the end of the round: the driver's contract fields, the
    roofline object of the dominant kernel, the CPU baseline timed beside it, the repeats behind the median, and the extras the
    round-3 verdict asked for (per-configuration rooflines, config 5 at 2 x 499,849 agents with the policy's dtype named)"""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "agent-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2          # PMC bytes vs algorithmic bytes of the same launch
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / sum(d["config"]["agents_at_start"]) - 1) < 0.25    # value ~ agents / step time
    assert d["repeats"] == 5 and len(d["repeats_ms_per_step"]) == 5 and sorted(d["repeats_ms_per_step"])[2] == round(d["ms_per_step"], 4)
    x = d["extra"]
    assert "error" not in x
    for k in ("gather_500_100k", "test_1m_2x500k"):
        assert 0 < x[k]["roofline"]["frac"] < 1
    assert 0 < x["battle_200_2x2000"]["calls_with_events"]["roofline"]["frac"] < 1
    c5 = x["c5_train_round_1m"]
    assert c5["bf16_policy"]["agents"] == [499849, 499849] and "bf16" in c5["bf16_policy"]["policy_dtype"] and c5["f32_policy"]["policy_dtype"].startswith("f32")
    assert x["battle_selfplay_2x400k_f32_policy"]["policy_dtype"] == "f32"


def test_bench_command_line_parses_without_a_gpu():
    """bench.py is the driver's contract: its command line (and with it the whole file) must at least compile and describe itself on the
    CPU box -- the flags the driver passes (--gpus, --steps, --warmup) and the ones the profiles quote"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--repeats", "--event-every", "--preheat-ms", "--backend", "--gather", "--no-extras"):
        assert flag in p.stdout, flag

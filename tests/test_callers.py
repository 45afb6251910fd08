"""The reference's own caller scripts, byte for byte, against `import magent` = this repository.

`make -C oracle ref` stages examples/train_battle.py, train_gather.py, train_pursuit.py and scripts/test/test_1m.py from
/root/reference into the gitignored oracle/_ref/callers/ -- beside the compiled reference, and travelling to the GPU box like
it.  The files are run as they are (`runpy.run_path(..., run_name="__main__")` behind their own command line):

* `-m gpu`: on the PRODUCT library (magent_amd/lib/libmagent.so, the HIP engine) -- the launcher names no engine at all, and the
  process reports which engine libraries it has mapped at exit;
* `-m "not gpu"`: the same launcher with the CPU oracle named the way every test names a checker (`GridWorld._engine_path`),
  small worlds: the scripts' host side (argument parsing, model hosting, the episode buffer) is exercised in this container.

On the GPU the battle script runs twice at `--map_size 1000` (2 x 40,000 agents, 551 steps): on the default host-buffer path
(numpy observations through PCIe, the reference's ABI) and with MAGENT_DEVICE_OBS=1 (torch tensors on the engine's GPU) -- the
round times of both are printed and kept in gpurun_out/callers.json for INTEGRATION.md's claim.
"""
import json
import os
import re
import subprocess
import sys

import pytest

import helpers as H

ROOT = H.ROOT
CALLERS = os.path.join(ROOT, "oracle", "_ref", "callers")

LAUNCHER = r"""
import os, sys, runpy, json
engine = %(engine)r
if engine:                                   # CPU suite only: the checker named the way tests/helpers.world_on does
    import magent_amd.gridworld as gw
    gw.GridWorld._engine_path = engine
sys.argv = %(argv)r
try:
    runpy.run_path(%(script)r, run_name="__main__")
finally:
    libs = sorted(set(line.split()[-1] for line in open("/proc/self/maps") if line.rstrip().endswith(".so") and
                      ("libmagent" in line or "liboracle" in line)))
    print("ENGINE_LIBS " + json.dumps(sorted(set(os.path.relpath(os.path.realpath(p), %(root)r) for p in libs))))
"""


def staged(name):
    path = os.path.join(CALLERS, name)
    if not os.path.isfile(path):
        pytest.fail("%s is not staged: run `make -C oracle ref` where /root/reference exists (oracle/Makefile)" % path)
    return path


def run_caller(name, argv, cwd, engine=None, env=None, timeout=1500):
    os.makedirs(os.path.join(str(cwd), "build"), exist_ok=True)     # (train_*.py: set_render_dir("build/render") is an os.mkdir)
    code = LAUNCHER % dict(engine=engine, argv=[name] + list(argv), script=staged(name), root=os.path.realpath(ROOT))
    e = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    e.pop("MAGENT_DEVICE_OBS", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, "-c", code], cwd=str(cwd), env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    libs = json.loads(re.search(r"^ENGINE_LIBS (.*)$", p.stdout, re.M).group(1))
    return p.stdout, libs


def keep(record):
    """GPU runs: the figures the literal scripts printed, merged into gpurun_out/callers.json (scratch that comes back with gpurun)"""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "callers.json")
    have = json.load(open(path)) if os.path.exists(path) else {}
    have.update(record)
    json.dump(have, open(path, "w"), indent=1, sort_keys=True)


PRODUCT = ["magent_amd/lib/libmagent.so"]


def battle_figures(out):
    m = re.search(r"steps: (\d+),  total time: ([\d.]+),  step average ([\d.]+)", out)
    r = re.search(r"round time ([\d.]+)", out)
    n = re.search(r"eps [\d.]+ number \[(\d+), (\d+)\]", out)
    return {"steps": int(m.group(1)), "sample_s": float(m.group(2)), "round_s": float(r.group(1)),
            "agents": [int(n.group(1)), int(n.group(2))], "ms_per_step": 1e3 * float(m.group(2)) / int(m.group(1))}


# ------------------------------------------------------------------------------------------------ on the MI355X: the product
@pytest.mark.gpu
def test_train_battle_literal_trains_a_round_on_the_hip_engine(tmp_path):
    """examples/train_battle.py:143-231 (`__main__`) and :45-140 (`play_a_round`): `--train --render --n_round 1 --map_size 100` --
    two DQNs (`from magent.builtin.tf_model import DeepQNetwork`, :174) act for a whole 551-step round on the default host path,
    the round is rendered (`set_render_dir`, :153; `env.render()`, :90), both models train on it."""
    out, libs = run_caller("train_battle.py", ["--train", "--render", "--n_round", "1", "--map_size", "100"], tmp_path)
    assert libs == PRODUCT, libs
    assert "===== sample =====" in out and "===== train =====" in out and "round time" in out
    assert re.search(r"batches: \d+,  total time", out)
    render = tmp_path / "build" / "render"
    assert (render / "config.json").is_file() and (render / "video_1.txt").stat().st_size > 1000
    fig = battle_figures(out)
    assert fig["agents"] == [400, 400] and fig["steps"] == 551
    print("train_battle.py --train --map_size 100:", fig)
    keep({"train_battle_100_train_host_path": fig})


@pytest.mark.gpu
@pytest.mark.parametrize("device_obs", ["0", "1"])
def test_train_battle_literal_at_map_size_1000(tmp_path, device_obs):
    """the same file at `--map_size 1000` (2 x 40,000 agents from its own generate_map, :15-40), one sampling round of 551 steps,
    both sides through the float32 DQN: on the reference's host-buffer ABI, and with MAGENT_DEVICE_OBS=1 (nothing but the
    environment variable differs: get_observation hands out torch tensors on the engine's GPU)"""
    out, libs = run_caller("train_battle.py", ["--n_round", "1", "--map_size", "1000"], tmp_path,
                           env={"MAGENT_DEVICE_OBS": device_obs})
    assert libs == PRODUCT, libs
    fig = battle_figures(out)
    assert fig["agents"] == [40000, 40000] and fig["steps"] == 551
    print("train_battle.py --map_size 1000, MAGENT_DEVICE_OBS=%s:" % device_obs, fig)
    keep({"train_battle_1000_device_obs_%s" % device_obs: fig})


@pytest.mark.gpu
@pytest.mark.parametrize("device_obs", ["0", "1"])
def test_test_1m_literal(tmp_path, device_obs):
    """scripts/test/test_1m.py:62-129 with its own defaults but the size: `--agent_number 1000000 --n_step 5` -- a 4472 x 4472 map,
    100,000 walls, 500,000 prey and 500,000 2x2 predators, RandomActor on both sides, 20 warm-up steps + 5 timed ones"""
    out, libs = run_caller("test_1m.py", ["--n_step", "5", "--agent_number", "1000000"], tmp_path, env={"MAGENT_DEVICE_OBS": device_obs})
    assert libs == PRODUCT, libs
    fps = float(re.search(r"^FPS ([\d.eE+-]+)", out, re.M).group(1))
    assert "===== step 24 =====" in out and "number of deer: 500000" in out
    tigers = [int(x) for x in re.findall(r"number of tiger: (\d+)", out)]
    assert 400000 < tigers[-1] <= tigers[0] <= 500000 and len(tigers) == 25
    print("test_1m.py --agent_number 1000000, MAGENT_DEVICE_OBS=%s: FPS %.2f" % (device_obs, fps))
    keep({"test_1m_device_obs_%s" % device_obs: {"fps": fps, "agent_steps_per_s": fps * 1e6}})


@pytest.mark.gpu
def test_train_gather_and_train_pursuit_literal(tmp_path):
    """examples/train_gather.py (imports `magent.builtin.mx_model.DeepQNetwork`, :10) and examples/train_pursuit.py, one training round each"""
    out, libs = run_caller("train_gather.py", ["--train", "--n_round", "1", "--map_size", "60"], tmp_path)
    assert libs == PRODUCT and "round time" in out, (libs, out[-500:])
    out, libs = run_caller("train_pursuit.py", ["--train", "--n_round", "1", "--map_size", "60"], tmp_path)
    assert libs == PRODUCT and "round time" in out, (libs, out[-500:])


# ------------------------------------------------------------------------------------------------ in this container: host side only
needs_callers = pytest.mark.skipif(not os.path.isdir(CALLERS), reason="oracle/_ref/callers not staged (no /root/reference here)")


@needs_callers
@pytest.mark.parametrize("name,argv", [
    ("train_gather.py", ["--train", "--n_round", "1", "--map_size", "40"]),
    ("train_pursuit.py", ["--train", "--n_round", "1", "--map_size", "30"]),
    ("test_1m.py", ["--n_step", "1", "--agent_number", "400"]),
])
def test_staged_callers_run_on_the_cpu_checker(tmp_path, name, argv):
    out, libs = run_caller(name, argv, tmp_path, engine=H.ensure_oracle(), timeout=900)
    assert libs == ["oracle/liboracle.so"], libs
    assert ("FPS" in out) if name == "test_1m.py" else ("round time" in out)

"""`bench.py --gpus N` the way the driver starts it: no launcher around it, WORLD_SIZE unset.

bench.py re-runs itself as N ranks under torch.distributed.run (spawn_ranks).  On the single-GPU test box the two ranks
share the GPU, so the process group is gloo (RCCL refuses two ranks on one device) and the observation rows are staged
through host memory; everything else is the N > 1 path of the real run: one HIP engine replica per rank (seed 12345 +
rank), the counts-then-rows exchange of magent_amd.replicas.ObservationGather with two view tensors used alternately,
max-over-ranks timing, whole-job aggregation.  --check-gather digests, on every rank, the rows it rendered and the
shards it received: shard r anywhere must be the bytes rank r's engine rendered."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_ranks(monkeypatch):
    """CPU: --gpus N without a launcher builds the driver's own command line (torch.distributed.run, 127.0.0.1)"""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class FakeLauncher(object):          # (the launcher's stdout is passed through line by line; rank 0's one JSON line makes the call a success)
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen.update(cmd=cmd, env=env)
            self.stdout = iter(["some launcher noise\n", '{"metric": "m", "value": 1}\n'])

        def wait(self):
            return seen.get("rc", 0)
    monkeypatch.setattr(bench.subprocess, "Popen", FakeLauncher)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import io
    out, err = io.StringIO(), io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    monkeypatch.setattr(sys, "stderr", err)
    assert bench.main() == 0
    seen["rc"] = 1                       # a rank died after rank 0 had printed its line: the line stands, so does the run -- and it says so
    assert bench.main() == 0
    monkeypatch.undo()
    lines = [json.loads(ln) for ln in out.getvalue().splitlines() if ln.startswith('{"metric"')]
    assert [ln["launcher_rc"] for ln in lines] == [0, 1] and lines[0]["value"] == 1
    # stdout is the line and nothing else: what the launcher or a rank prints beside it is relayed to stderr
    assert len(out.getvalue().splitlines()) == 2 and "some launcher noise" in err.getvalue()
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_sigterm_reaches_a_rank_that_is_blocked_outside_the_interpreter():
    """CPU: bench.relay_sigterm -- what lets rank 0 print its one line when the launcher ends it because another rank died.  The main
    thread sits in a blocking read (a collective that never returns, for the purpose); SIGTERM must still produce the line, once."""
    code = ("import os, sys, threading\n"
            "sys.path.insert(0, %r)\n"
            "import bench\n"
            "def on_term():\n"
            "    print('{\"metric\": \"m\", \"extra\": \"sigterm\"}', flush=True)\n"
            "    os._exit(0)\n"
            "bench.relay_sigterm(on_term)\n"
            "print('ready', flush=True)\n"
            "r, w = os.pipe()\n"
            "os.read(r, 1)\n" % ROOT)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "ready"
    p.terminate()
    out = p.stdout.read()
    assert p.wait(timeout=30) == 0 and out.count('{"metric"') == 1, out


@pytest.mark.gpu
def test_a_rank_that_dies_in_the_extra_costs_the_extra_not_the_line():
    """`bench.py --gpus 2` where rank 1 dies inside extra.c4_gather_rccl (--fault, a test hook): torch.distributed.run ends rank 0 with
    SIGTERM; rank 0 still prints exactly ONE JSON line -- the headline it had already measured, the extra marked as failed -- and the
    command succeeds"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--repeats", "1", "--map-size", "200", "--agents", "6000", "--force-extra", "--fault", "kill-rank-1-in-extra"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    rec = json.loads(lines[0])
    # (which of the two ends the extra first is a race: the launcher's SIGTERM, or the collective library noticing the dead peer and raising)
    err = rec["extra"]["c4_gather_rccl"]["error"]
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and ("SIGTERM" in err or "rror" in err), err


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_and_exchange_their_observations():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--gather", "obs",
                        "--check-gather", "--steps", "3", "--warmup", "1", "--map-size", "200", "--agents", "6000",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["rccl_ranks"] == 2 and rec["config"]["envs"] == 2
    g = rec["config"]["gather_detail"]
    assert g["payload_bytes_sent_per_step"] > 0 and g["view_buffers"] == 2
    v = g["verified"]
    assert v["all_shards_bit_identical"] is True and v["pairs_checked"] == 8 and v["distinct_replicas"] == 2
    assert rec["value"] > 0 and rec["roofline"]["achieved"] > 0


@pytest.mark.gpu
def test_two_ranks_rccl_when_two_gpus():
    """the same through RCCL (device tensors, side stream, stream-ordered waits) where the box has two GPUs"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: RCCL needs a device per rank")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gather", "obs", "--check-gather",
                        "--steps", "5", "--warmup", "2", "--map-size", "400", "--agents", "50000", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["rccl_ranks"] == 2 and rec["config"]["backend"] == "nccl"
    assert rec["config"]["gather_detail"]["verified"]["all_shards_bit_identical"] is True


@pytest.mark.gpu
def test_default_multi_rank_line_carries_the_c4_gather_extra():
    """`bench.py --gpus 2` with nothing else but the dry-run backend -- what the driver's N > 1 command produces: the replicas
    headline on the default workload (no gather), and behind it `extra.c4_gather_rccl`: BASELINE config 4 on the job's ranks
    (gather 500 x 500, 100k agents + 20k food per replica), timed without and with the exchange of every replica's observation
    tensor, the exchange's own duration, bytes per peer and GB/s per link against the xGMI link peak, every gathered shard verified
    bit for bit against the rows its owner rendered"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--repeats", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["gather"] == "none" and rec["config"]["agents_at_start"] == [400000, 400000]
    x = rec["extra"]["c4_gather_rccl"]
    assert x["verified"]["all_shards_bit_identical"] is True and x["verified"]["distinct_replicas"] == 2
    assert x["payload_bytes_to_each_peer"] > 50000 * 6300 and x["exchange_ms"] > 0 and x["GBps_per_link"] > 0
    assert x["ms_per_step_with_gather"] > 0 and x["ms_per_step_without_gather"] > 0 and x["xgmi_link_peak_GBps"] == 153.0


@pytest.mark.gpu
def test_default_eight_rank_command_as_a_dry_run():
    """THE command the driver runs for the 8-GPU scaling point -- `python bench.py --gpus 8` with its default steps and warm-up, nothing else but
    the dry-run backend (8 ranks share this box's one GPU, so the process groups are gloo; everything else is the N = 8 path: eight engine
    replicas, the barrier + max-over-ranks timing, the lazily created gather group behind a watchdog, extra.c4_gather_rccl with seven peers
    per rank).  Must finish within two minutes, print exactly ONE JSON line, and that line must carry what the scaling report reads: the
    whole-job rate, rccl_ranks, the rates with and without the gather, GB/s per link, every shard pair verified."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    wall = time.time() - t0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    assert wall < 120, wall
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["launcher_rc"] == 0 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["rccl_ranks"] == 8 and rec["config"]["envs"] == 8 and rec["config"]["gather"] == "none"
    x = rec["extra"]["c4_gather_rccl"]
    for key in ("ms_per_step_without_gather", "agent_steps_per_s_without_gather", "ms_per_step_with_gather", "agent_steps_per_s_with_gather", "exchange_ms",
                "payload_bytes_to_each_peer", "GBps_per_link", "xgmi_link_peak_GBps", "link_frac"):
        assert x[key] > 0, key
    assert x["verified"]["all_shards_bit_identical"] is True and x["verified"]["pairs_checked"] == 64 and x["verified"]["distinct_replicas"] == 8

"""CPU test of the N > 1 path: two processes on gloo, each owning its own environment replica.

The replicas here are driven by the CPU oracle (no GPU in this test); what is under test is the host logic of
magent_amd/replicas.py that bench.py uses on RCCL: per-replica seeds, the observation gather (counts first, then sends /
receives sized by count into buffers allocated once -- or the padded one-collective form), max-over-ranks timing,
whole-job aggregation."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, oracle, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import magent_amd
    from magent_amd import replicas
    assert replicas.replica_info() == (rank, rank, world)
    env = H.gridworld("battle", lib=oracle, map_size=30)
    env.set_seed(replicas.replica_seed(12345, rank))
    env.reset()
    h0, h1 = env.get_handles()
    env.add_agents(h0, "random", n=40 + 10 * rank)       # replicas differ in size, like after deaths
    env.add_agents(h1, "random", n=40)
    view, _ = env.get_observation(h0)
    n = env.get_num(h0)
    # the observation gather of SURVEY.md 8e: counts first, then rows sized by count; every buffer allocated once
    send = torch.zeros((64,) + tuple(view.shape[1:]))                  # the replica's (preallocated) observation tensor
    send[:n] = torch.from_numpy(view)
    for mode in ("exact", "padded"):
        g = replicas.ObservationGather(view.shape[1:], capacity=64, mode=mode)
        ptrs = (g.recv.data_ptr(), g._count_recv.data_ptr(), g._count_send.data_ptr())
        for it, m in enumerate((n, n - 7 - 3 * rank, n - 20, 0 if rank == 1 else 5)):     # populations shrink, unevenly; one replica dies out
            shards, counts = g.gather(send, m)
            other = 1 - rank
            n_other = 40 + 10 * other
            want = [n_other, n_other - 7 - 3 * other, n_other - 20, 0 if other == 1 else 5][it]
            assert counts[rank] == m and counts[other] == want and [s.shape[0] for s in shards] == counts
            assert torch.equal(shards[rank], send[:m])
            assert shards[other].untyped_storage().data_ptr() == g.recv.untyped_storage().data_ptr()   # a view of the receive area
            assert (g.recv.data_ptr(), g._count_recv.data_ptr(), g._count_send.data_ptr()) == ptrs      # nothing was re-allocated
            if mode == "exact":
                assert g.bytes_sent == m * g.row_bytes and g.bytes_received == want * g.row_bytes       # no padding travels
            if it == 0:
                first = [s.clone() for s in shards]
    shards, counts = first, [40, 50]
    t = replicas.max_over_replicas(1.0 + rank)
    total = replicas.sum_over_replicas(n)
    assert t == 2.0 and total == 90.0
    np.save(os.path.join(out_dir, "view_rank%d.npy" % rank), view)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered_rank1.npy"), shards[1].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_gloo(tmp_path):
    oracle = H.ensure_oracle()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), oracle, str(tmp_path)), nprocs=world, join=True)
    # rank 0 received exactly what rank 1 rendered, and the replicas are not clones (different seeds)
    a = np.load(tmp_path / "gathered_rank1.npy")
    b = np.load(tmp_path / "view_rank1.npy")
    assert a.tobytes() == b.tobytes()
    assert np.load(tmp_path / "view_rank0.npy").shape[0] == 40 and b.shape[0] == 50


# ---------------------------------------------------------------------------------------------- the GPU sequencing, on a recording stand-in
class _Rec(object):
    """a recording stand-in for torch.cuda / torch.distributed (replicas._TorchBackend): two "ranks" that live in one process, streams
    and events that only log what is asked of them.  What is under test is the ORDER of ObservationGather's GPU path -- the one leg
    that needs two GPUs to execute for real: counts behind the producer's event, the host wait on the counts only, every Work.wait()
    issued on the side stream BEFORE the events that publish the exchange, release() / wait() ordering the right streams."""

    class Stream(object):
        def __init__(self, log, name):
            self.log, self.name = log, name

        def record_event(self, ev):
            self.log.append(("record", ev.name, self.name))

        def wait_event(self, ev):
            self.log.append(("stream_waits", self.name, ev.name))

    class Event(object):
        def __init__(self, log, name):
            self.log, self.name = log, name

        def record(self, stream):
            self.log.append(("record", self.name, stream.name))

        def synchronize(self):
            self.log.append(("host_waits", self.name))

    class Work(object):
        def __init__(self, rec, k):
            self.rec, self.k = rec, k

        def wait(self):
            self.rec.log.append(("work_wait", self.k, self.rec.cur.name))

    class On(object):
        def __init__(self, rec, stream):
            self.rec, self.stream = rec, stream

        def __enter__(self):
            self.prev, self.rec.cur = self.rec.cur, self.stream
            self.rec.log.append(("enter", self.stream.name))

        def __exit__(self, *a):
            self.rec.cur = self.prev
            self.rec.log.append(("exit", self.stream.name))

    def __init__(self, rank, peer_count, peer_rows):
        self.log, self.streams, self._rank, self.n_ev = [], True, rank, 0
        self.peer_count, self.peer_rows = peer_count, peer_rows
        self.default = self.Stream(self.log, "torch")
        self.cur = self.default

    def world_size(self):
        return 2

    def rank(self):
        return self._rank

    def new_stream(self):
        return self.Stream(self.log, "side")

    def new_event(self, timing=False):
        self.n_ev += 1
        return self.Event(self.log, "ev%d" % self.n_ev)

    def current_stream(self):
        return self.cur

    def on(self, stream):
        return self.On(self, stream)

    def all_gather_into_tensor(self, out, inp):
        self.log.append(("all_gather", tuple(out.shape), self.cur.name))
        if out.dim() == 1:                                      # the counts
            out[self._rank], out[1 - self._rank] = int(inp[0]), self.peer_count
        else:
            rows = out.shape[0] // 2
            out[self._rank * rows:(self._rank + 1) * rows] = inp
            out[(1 - self._rank) * rows:(2 - self._rank) * rows][:len(self.peer_rows)] = self.peer_rows

    def exchange(self, sends, recvs):
        self.log.append(("exchange", [(tuple(t.shape), p) for t, p in sends], [(tuple(t.shape), p) for t, p in recvs], self.cur.name))
        for t, p in recvs:
            t.copy_(self.peer_rows[:t.shape[0]])
        return [self.Work(self, k) for k in range(len(sends) + len(recvs))]


def test_observation_gather_stream_sequencing_on_a_recording_backend():
    from magent_amd import replicas
    peer_rows = torch.arange(5 * 6, dtype=torch.float32).reshape(5, 2, 3) + 100
    rec = _Rec(rank=0, peer_count=5, peer_rows=peer_rows)
    g = replicas.ObservationGather((2, 3), capacity=8, device="cpu", mode="exact", backend=rec)
    assert (g.world, g.rank) == (2, 0) and g.stream is not None
    engine = _Rec.Stream(rec.log, "engine")
    views = [torch.rand(8, 2, 3), torch.rand(8, 2, 3)]

    # ---- one exchange: launch behind the render, post under the step, wait for the consumer
    g.launch(views[0], 3, producer_stream=engine)
    L = list(rec.log); del rec.log[:]
    ready, counted, done = g._ready.name, g._counted.name, g._done.name
    assert L[0] == ("record", ready, "engine") and L[1] == ("stream_waits", "side", ready)           # the counts go behind the render
    assert L[2] == ("enter", "side") and L[3] == ("all_gather", (2,), "side") and L[4] == ("record", counted, "side") and L[5] == ("exit", "side")
    g.post()
    L = list(rec.log); del rec.log[:]
    assert L[0] == ("host_waits", counted)                                                           # the only host wait: 16 bytes
    assert L[1] == ("enter", "side") and L[2] == ("record", g._began.name, "side")
    assert L[3] == ("exchange", [((3, 2, 3), 1)], [((5, 2, 3), 1)], "side")                          # sized by count, nothing padded
    waits = [e for e in L if e[0] == "work_wait"]
    assert [w[1] for w in waits] == [0, 1] and all(w[2] == "side" for w in waits)                    # every work, on the side stream
    i_last_wait = max(i for i, e in enumerate(L) if e[0] == "work_wait")
    i_done = L.index(("record", done, "side"))
    busy0 = g._busy[views[0].data_ptr()].name
    i_busy = L.index(("record", busy0, "side"))
    assert i_last_wait < i_done < i_busy and L[-1] == ("exit", "side")                               # the events cover the transfers
    assert g.counts == [3, 5] and g.bytes_sent == 3 * 24 and g.bytes_received == 5 * 24
    g.post()                                                                                         # idempotent
    assert rec.log == []
    shards = g.wait(consumer_stream=rec.default)
    assert rec.log == [("stream_waits", "torch", done)]; del rec.log[:]
    assert torch.equal(shards[0], views[0][:3]) and torch.equal(shards[1], peer_rows)
    assert shards[1].untyped_storage().data_ptr() == g.recv.untyped_storage().data_ptr()

    # ---- two send tensors used alternately: the render into views[1] does not wait for anything, the next render into views[0]
    # waits for the exchange that read views[0] -- and an exchange that is launched but not posted yet is posted first
    g.release(views[1], engine)
    assert rec.log == []
    g.launch(views[1], 2, producer_stream=engine)
    del rec.log[:]
    g.release(views[0], engine)                          # views[0]'s last reader was exchange 1: already posted
    assert rec.log == [("stream_waits", "engine", busy0)]; del rec.log[:]
    g.release(views[1], engine)                          # its exchange is not posted yet: posted now, then waited for
    kinds = [e[0] for e in rec.log]
    assert kinds[0] == "host_waits" and "exchange" in kinds and rec.log[-1] == ("stream_waits", "engine", g._busy[views[1].data_ptr()].name)
    assert g._busy[views[1].data_ptr()].name != busy0

    # ---- padded mode: one collective of `capacity` rows, no works, the same event order
    rec2 = _Rec(rank=1, peer_count=4, peer_rows=peer_rows[:4])
    gp = replicas.ObservationGather((2, 3), capacity=8, device="cpu", mode="padded", backend=rec2)
    gp.launch(views[0], 6, producer_stream=_Rec.Stream(rec2.log, "engine"))
    del rec2.log[:]
    shards = gp.wait(consumer_stream=rec2.default)
    kinds = [e[0] for e in rec2.log]
    assert kinds == ["host_waits", "enter", "record", "all_gather", "record", "record", "exit", "stream_waits"]
    assert gp.counts == [4, 6] and torch.equal(shards[0], peer_rows[:4]) and torch.equal(shards[1], views[0][:6])
    assert gp.bytes_sent == 8 * 24


# ---------------------------------------------------------------------------------------------- RCCL on the GPU box
import pytest  # noqa: E402


@pytest.mark.gpu
def test_one_rank_process_group_on_rccl():
    """ProcessGroupNCCL (= RCCL on ROCm) loads and eager-initialises on the MI355X image, and an all_reduce, an
    all_gather_into_tensor and the replica helpers run on device tensors through it -- the 1-GPU box cannot run two ranks on RCCL
    (it refuses two ranks on one device), so this is the part of the N > 1 backend that CAN execute before the first 8-GPU run.
    In this process (not a child), so that the RCCL library shows among the native libraries the test process loaded."""
    assert not dist.is_initialized()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # eager init: the communicator is created here
    try:
        from magent_amd import replicas
        x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        want = float(x.sum())
        dist.all_reduce(x)
        out = torch.empty(1 << 20, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out, x)
        g = replicas.ObservationGather((13, 13, 7), capacity=64, device=dev)   # world 1: the shards are the send tensor
        view = torch.rand(64, 13, 13, 7, device=dev)
        shards, counts = g.gather(view, 50)
        t = replicas.max_over_replicas(1.5, device=dev)
        torch.cuda.synchronize()
        assert float(out.sum()) == want and counts == [50] and torch.equal(shards[0], view[:50]) and t == 1.5
        assert dist.get_backend() == "nccl"
        maps = open("/proc/self/maps").read()
        assert "librccl" in maps or "libnccl" in maps
    finally:
        dist.destroy_process_group()
        for k in ("MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)


@pytest.mark.gpu
def test_one_rank_rccl_data_group_under_a_gloo_control_plane():
    """what bench.py --gpus N sets up: the job's control plane (barriers, timing reductions) on gloo, the observation exchange on a group of
    its own created with backend nccl (= RCCL) -- one rank of it: the 1-GPU box cannot hold two.  The data group's collectives run on
    device tensors, the default group's on CPU tensors, side by side in one process."""
    assert not dist.is_initialized()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from magent_amd import replicas
        data = dist.new_group(backend="nccl")
        assert dist.get_backend() == "gloo" and dist.get_backend(data) == "nccl"
        g = replicas.ObservationGather((13, 13, 7), capacity=64, device=dev, group=data)
        view = torch.rand(64, 13, 13, 7, device=dev)
        shards, counts = g.gather(view, 41)
        x = torch.ones(1 << 16, device=dev)
        dist.all_reduce(x, group=data)                                  # a collective of the data group on device memory
        t = replicas.max_over_replicas(2.5, device=torch.device("cpu"))  # the control plane on CPU memory
        dist.barrier()
        torch.cuda.synchronize()
        assert counts == [41] and torch.equal(shards[0], view[:41]) and t == 2.5 and float(x.sum()) == float(1 << 16)
    finally:
        dist.destroy_process_group()
        for k in ("MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)

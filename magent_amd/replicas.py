"""Multi-GPU use of the engine: independent environment replicas, one process per GPU.

One environment is a single coupled grid, so the hot path does not shard: N GPUs run N environments.
The only cross-GPU traffic the north star asks for is an optional gather of the batched observation tensor, so that
one policy batch can see every replica.  `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI inside a node); the
same code runs on "gloo" CPU tensors, which is how the CPU tests cover it.

`ObservationGather` is the exchange SURVEY.md 8e describes:
  * counts first -- replicas differ in size after deaths: one `all_gather_into_tensor` of a single int64 per rank;
  * then the rows, sized by count: every rank posts one send to and one receive from each peer (`batch_isend_irecv`, on
    RCCL a single group of ncclSend / ncclRecv).  On the 8 x MI355X full mesh every pair of GPUs has its own xGMI link
    (~153 GB/s per direction), so the seven transfers of a rank run side by side and the exchange takes
    max_shard_bytes / 153 GB/s -- a ring all-gather would pay seven hops on one link each.  Nothing is padded: exactly
    n_r rows travel from rank r;
  * every buffer is allocated once (receive area sized for `capacity` rows per replica, count tensors, events): a call
    allocates nothing on the device;
  * on GPUs the exchange is issued on a side stream behind an event recorded on the producer's stream, so the engine's
    step kernels (which do not touch the observation) run under it; `wait()` orders a consumer stream after it.

`mode="padded"` is the one-collective alternative (`all_gather_into_tensor` of `capacity` rows per rank): no host
round trip for the counts before the payload is posted, at the price of sending the padding.
"""
import os

import torch
import torch.distributed as dist


def replica_info():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: a single replica)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def replica_seed(base_seed, rank):
    """engine seed of replica `rank`: replicas must not be clones of each other"""
    return base_seed + rank


class _TorchBackend(object):
    """what ObservationGather needs from torch.distributed and torch.cuda, behind one small object so that the ORDER in which
    the exchange uses streams, events and works can be unit-tested with a recording stand-in (tests/test_replicas.py) --
    the RCCL path itself needs two GPUs, which the build container and the 1-GPU test box do not have"""

    def __init__(self, device, group=None):
        self.device, self.group = torch.device(device), group
        self.streams = self.device.type == "cuda"

    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    # -- streams and events (GPU only)
    def new_stream(self):
        return torch.cuda.Stream(device=self.device)

    def new_event(self, timing=False):
        return torch.cuda.Event(enable_timing=timing)

    def current_stream(self):
        return torch.cuda.current_stream(self.device)

    def on(self, stream):
        return torch.cuda.stream(stream)

    # -- collectives and point-to-point
    def all_gather_into_tensor(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def exchange(self, sends, recvs):
        """one group of point-to-point transfers: sends = [(tensor, peer)], recvs = [(tensor, peer)] -> the works"""
        ops = [dist.P2POp(dist.irecv, t, peer, group=self.group) for t, peer in recvs]
        ops += [dist.P2POp(dist.isend, t, peer, group=self.group) for t, peer in sends]
        return dist.batch_isend_irecv(ops) if ops else []


class ObservationGather(object):
    """All-gather of a per-replica observation tensor whose row count differs between replicas.

    row_shape : shape of one agent's observation, e.g. (13, 13, 7)
    capacity  : rows reserved per replica in the receive area (>= the largest population any replica will hold)
    device    : where the tensors live ("cpu" for gloo); mode: "exact" (counts, then sized sends / receives) | "padded"
    backend   : test seam (see _TorchBackend); the default talks to torch.distributed / torch.cuda
    """

    def __init__(self, row_shape, capacity, dtype=torch.float32, device="cpu", mode="exact", group=None, backend=None):
        assert mode in ("exact", "padded")
        self.mode, self.group = mode, group
        self.device = torch.device(device)
        self._b = backend if backend is not None else _TorchBackend(self.device, group)
        self.world, self.rank = self._b.world_size(), self._b.rank()
        self.capacity, self.row_shape = int(capacity), tuple(row_shape)
        self.recv = torch.empty((self.world, self.capacity) + self.row_shape, dtype=dtype, device=self.device)
        self._count_send = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._count_recv = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        self._count_host = torch.zeros(self.world, dtype=torch.int64).pin_memory() if self.device.type == "cuda" else None
        self.counts = [0] * self.world
        self.row_bytes = self.recv[0, 0].numel() * self.recv.element_size()
        self.bytes_sent = self.bytes_received = 0          # of the last exchange, payload only
        if self._b.streams:
            self.stream = self._b.new_stream()
            self._ready, self._counted = self._b.new_event(), self._b.new_event()
            self._began, self._done = self._b.new_event(True), self._b.new_event(True)     # around the payload: exchange_ms()
        else:
            self.stream = None
        self._posted, self._view, self._n = True, None, 0
        self._busy = {}            # send tensor (data_ptr) -> event recorded behind the last exchange that read it
        self._host_ms = None       # gloo: the payload's wall time (it blocks the host)

    # -- step 1: the counts (and, for GPUs, the hand-over from the producer's stream)
    def launch(self, view, n, producer_stream=None):
        """`view[:n]` are this replica's rows (a preallocated tensor of at least n rows on self.device; in padded mode of at
        least `capacity` rows).  Starts the exchange of the counts; `post()` sends the rows.  producer_stream: the CUDA
        stream that fills `view` (default: the current one)"""
        assert view.shape[1:] == self.row_shape and view.is_contiguous() and n <= self.capacity and view.shape[0] >= n
        self._view, self._n, self._posted = view, int(n), False
        if self.world == 1:
            self.counts = [self._n]
            return
        if self.stream is not None:
            (producer_stream or self._b.current_stream()).record_event(self._ready)
            self.stream.wait_event(self._ready)
            with self._b.on(self.stream):
                self._count_send.fill_(self._n)
                self._b.all_gather_into_tensor(self._count_recv, self._count_send)
                (self._count_host if self._count_host is not None else self._count_recv).copy_(self._count_recv, non_blocking=True)
                self._counted.record(self.stream)
        else:
            self._count_send.fill_(self._n)
            self._b.all_gather_into_tensor(self._count_recv, self._count_send)

    def _exchange_rows(self, view, n):
        """the payload, on whatever stream is current: one collective (padded) or one group of sends / receives sized by count.
        Every work is waited for HERE, on the issuing stream: Work.wait() on a ProcessGroupNCCL work is a STREAM-side wait -- it
        makes the current stream (the side stream) depend on the communicator's internal stream and does not block the host.
        Without it an event recorded behind the exchange would not cover the transfers.  On gloo it blocks
        until the rows are in."""
        if self.mode == "padded":
            assert view.shape[0] >= self.capacity, "padded mode sends `capacity` rows: the send tensor must hold them"
            self._b.all_gather_into_tensor(self.recv.view(self.world * self.capacity, *self.row_shape), view[:self.capacity])
            self.bytes_sent = self.capacity * self.row_bytes * (self.world - 1)
            self.bytes_received = self.bytes_sent
            return
        peers = [p for p in range(self.world) if p != self.rank]   # sized by count, no padding; every pair has its own link on the full mesh
        recvs = [(self.recv[p, :self.counts[p]], p) for p in peers if self.counts[p] > 0]
        sends = [(view[:n], p) for p in peers] if n > 0 else []
        for work in self._b.exchange(sends, recvs):
            work.wait()
        self.bytes_sent = n * self.row_bytes * (self.world - 1)
        self.bytes_received = (sum(self.counts) - n) * self.row_bytes

    # -- step 2: the rows
    def post(self):
        """posts the payload (call it after the work that should run under the exchange has been enqueued)"""
        if self._posted:
            return
        self._posted = True
        if self.world == 1:
            return
        if self.stream is not None:
            self._counted.synchronize()                    # 8 * world bytes: the only host wait of the exchange
            self.counts = (self._count_host if self._count_host is not None else self._count_recv).tolist()
        else:
            self.counts = self._count_recv.tolist()
        assert max(self.counts) <= self.capacity, "capacity smaller than a replica's agent count"
        view, n = self._view, self._n
        if self.stream is None:
            import time
            t0 = time.perf_counter()
            self._exchange_rows(view, n)
            self._host_ms = (time.perf_counter() - t0) * 1e3
            return
        with self._b.on(self.stream):
            self._began.record(self.stream)
            self._exchange_rows(view, n)
            # both events sit BEHIND the works' stream-side waits: `_done` is what consumers wait for, `_busy[view]` what the
            # producer waits for before it overwrites the send tensor
            self._done.record(self.stream)
            ev = self._busy.get(view.data_ptr())
            if ev is None:
                ev = self._busy[view.data_ptr()] = self._b.new_event()
            ev.record(self.stream)

    def release(self, view, producer_stream=None):
        """orders `producer_stream` after the last exchange that READ `view`, so that it may be overwritten.  With two send
        tensors used alternately the render of step t+1 does not wait for the exchange of step t, only for that of t-1."""
        if self.stream is None or self.world == 1:
            return
        if not self._posted and self._view is not None and self._view.data_ptr() == view.data_ptr():
            self.post()
        ev = self._busy.get(view.data_ptr())
        if ev is not None:
            (producer_stream or self._b.current_stream()).wait_event(ev)

    def wait(self, consumer_stream=None):
        """orders `consumer_stream` (default: the current CUDA stream) after the exchange; returns the shards"""
        self.post()
        if self.stream is not None and self.world > 1:
            (consumer_stream or self._b.current_stream()).wait_event(self._done)
        return self.shards()

    def exchange_ms(self):
        """duration of the last payload exchange (rows only, counts excluded): on GPUs between two events on the side stream --
        blocks the host until the exchange is over; on gloo the host time the blocking exchange took"""
        if self.world == 1 or self._view is None:
            return 0.0
        if self.stream is None:
            return self._host_ms
        self.post()
        self._done.synchronize()
        return self._began.elapsed_time(self._done)

    def shards(self):
        """per-replica views of the gathered rows, trimmed to each replica's count (own rows: the send tensor itself)"""
        if self._view is None:         # nothing has been exchanged yet
            return []
        out = []
        for r in range(self.world):
            out.append(self._view[:self._n] if r == self.rank else self.recv[r, :self.counts[r]])
        return out

    def gather(self, view, n):
        """the whole exchange in one call"""
        self.launch(view, n)
        return self.wait(), list(self.counts)


def max_over_replicas(seconds, device=None):
    """wall time of the slowest replica (the bench's timing rule)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_replicas(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

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

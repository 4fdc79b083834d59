"""world_size-2 gloo tests of the data-parallel wrapper (CPU): sharding, weight broadcast, flat-gradient all-reduce,
max-over-ranks timing reduction - the N > 1 path of bench.py / train.py without GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeStore:
    def __init__(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_w = torch.randn(1000, generator=g)
        self.flat_stats = torch.randn(10, generator=g)
        self.flat_g = torch.full((1000,), float(rank + 1))


class _FakeModel:
    def __init__(self, rank):
        self.store = _FakeStore(rank)
        self.grad_sync = None


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from poisson_cnn_amd import parallel
    dp = parallel.DataParallel.from_env(backend='gloo')
    assert dp.world_size == world and dp.rank == rank
    assert dp.local_batch(8) == 4
    try:
        dp.local_batch(7)
        ok_err = False
    except ValueError:
        ok_err = True
    batch = torch.arange(8 * 3, dtype=torch.float32).view(8, 3)
    shard = dp.shard(batch)
    m = _FakeModel(rank)
    w_before = m.store.flat_w.clone()
    dp.attach(m)
    m.grad_sync(m.store.flat_g)
    t = dp.max_over_ranks(1.0 + rank)
    dp.barrier()
    q.put((rank, ok_err, shard.numpy().copy(), w_before.numpy(), m.store.flat_w.numpy().copy(), m.store.flat_g.numpy().copy(), t))
    torch.distributed.destroy_process_group()


def test_data_parallel_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, s0, wb0, w0, g0, t0), (r1, e1, s1, wb1, w1, g1, t1) = res
    assert e0 and e1
    assert np.array_equal(np.concatenate([s0, s1]), np.arange(24, dtype=np.float32).reshape(8, 3))   # even split by sample, rank order
    assert np.array_equal(w0, wb0) and np.array_equal(w1, wb0) and not np.array_equal(wb1, wb0)       # rank 0's weights everywhere
    assert np.all(g0 == 3.0) and np.all(g1 == 3.0)                                                    # SUM all-reduce of the flat bucket
    assert t0 == 2.0 and t1 == 2.0                                                                    # max over ranks


def test_single_rank_is_a_no_op():
    from poisson_cnn_amd import parallel
    dp = parallel.DataParallel()
    g = torch.ones(5)
    assert dp.all_reduce_sum(g) is g and dp.max_over_ranks(3.5) == 3.5 and dp.local_batch(6) == 6

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


def _bn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from poisson_cnn_amd import parallel
    dp = parallel.DataParallel.from_env(backend='gloo')
    out = []
    for training in (True, False):
        m = _FakeModel(rank)
        m.store.bn_training, m.store.nbn = training, 5
        dp.attach(m)                                             # broadcast: rank 0's statistics everywhere
        start = m.store.flat_stats.clone()
        m.store.flat_stats += float(rank + 1)                    # "this rank's shard moved the moving statistics by rank + 1"
        m.grad_sync(m.store.flat_g)
        out.append((start.numpy().copy(), m.store.flat_stats.numpy().copy(), m.store.flat_g.numpy().copy()))
    q.put((rank, out))
    torch.distributed.destroy_process_group()


def test_bn_moving_statistics_are_mean_reduced_in_training_mode():
    """VERDICT r4 missing #4 / SURVEY 8(e): with BatchNormalization in training mode every rank updates the moving statistics from its own shard;
    the reference's mirrored variables aggregate them with MEAN (models/Homogeneous_Poisson_NN_Legacy.py:53-57 under train/hpnn_legacy_train.py:37-41).
    After grad_sync both ranks must hold start + mean(1, 2); in the default inference mode the statistics are left alone (no collective)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (s0, a0, g0), (s0i, a0i, _) = res[0]
    (s1, a1, g1), (s1i, a1i, _) = res[1]
    assert np.array_equal(s0, s1)                                                   # attach broadcast rank 0's statistics
    assert np.array_equal(a0, a1) and np.allclose(a0, s0 + 1.5, rtol=0, atol=1e-6)  # training mode: mean over the ranks of the per-rank updates
    assert np.all(g0 == 3.0) and np.all(g1 == 3.0)                                  # the gradient all-reduce is unchanged
    assert np.allclose(a0i, s0i + 1.0) and np.allclose(a1i, s1i + 2.0)              # inference-mode BN: untouched by the wrapper


def test_single_rank_is_a_no_op():
    from poisson_cnn_amd import parallel
    dp = parallel.DataParallel()
    g = torch.ones(5)
    assert dp.all_reduce_sum(g) is g and dp.max_over_ranks(3.5) == 3.5 and dp.local_batch(6) == 6


def test_bench_self_launches_two_ranks_end_to_end():
    """`python bench.py --gpus 2` started BARE (no torchrun, no WORLD_SIZE): the parent spawns one fresh process per rank before touching
    any GPU, the ranks rendezvous on 127.0.0.1 (gloo here), all-reduce the 22.2 MB gradient bucket and rank 0 prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'launch-check', '--steps', '3', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['dry_run'] is True
    assert 'gloo' in out['config']['collective'] and out['value'] > 0


def _bare_bench(extra, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'launch-check', '--steps', '2', '--warmup', '1'] + extra,
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_launches_four_ranks_and_reports_the_ranks_it_saw():
    """the bare N > 1 launch at world size 4 (gloo): one compact, parseable JSON line that states how many ranks the
    process group itself saw."""
    import json
    r = _bare_bench(['--gpus', '4'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) < 3072, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 4 and out['config']['ranks_seen'] == 4 and out['config']['parallelism'] == 'dp4' and out['value'] > 0
    # every rank's own step time and convolution-kernel time (a slow rank vs collective cost), and what the NUMA binding did
    rm = out['rank_ms_per_step']
    assert len(rm['ranks']) == 4 and len(rm['conv_ms_per_step']) == 4 and all(v > 0 for v in rm['ranks'])
    assert abs(rm['max'] - max(rm['ranks'])) < 1e-9 and abs(rm['min'] - min(rm['ranks'])) < 1e-9
    assert isinstance(out['affinity'], dict) and 'bound' in out['affinity']          # no KFD topology in this container: bound False with a reason


def test_numa_binding_leaves_the_affinity_alone_when_the_topology_is_absent_or_disabled(monkeypatch):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    before = os.sched_getaffinity(0)
    rep = bench.bind_to_gpu_numa_node(3)
    if not rep['bound']:
        assert 'why' in rep and os.sched_getaffinity(0) == before
    else:                                                                            # a box with GPUs: the new set is a non-empty subset of the old one
        assert os.sched_getaffinity(0) <= before and rep['cpus'] == len(os.sched_getaffinity(0)) > 0
        os.sched_setaffinity(0, before)
    monkeypatch.setenv('PCNN_BENCH_AFFINITY', '0')
    assert bench.bind_to_gpu_numa_node(0) == {'bound': False, 'why': 'disabled'} and os.sched_getaffinity(0) == before


def test_bench_parent_returns_within_seconds_when_a_rank_dies_before_the_barrier():
    """A rank that exits before its first barrier must not leave the others waiting in the rendezvous until the driver's limit: the parent
    notices the exit, stops the other ranks, relays the failing rank's stderr and returns that rank's status."""
    import time
    t0 = time.monotonic()
    r = _bare_bench(['--gpus', '3', '--fail-rank', '2', '--timeout', '900'], timeout=300)
    took = time.monotonic() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert 'rank 2 exited with status 3' in r.stderr and '--fail-rank asked for this exit' in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]            # no result line from a failed run
    assert took < 120, took                                                           # seconds (python + torch start-up), not the 900 s group timeout


def test_bench_parent_enforces_the_launch_timeout():
    import time
    t0 = time.monotonic()
    r = _bare_bench(['--gpus', '2', '--launch-timeout', '0.5'], timeout=120)
    assert r.returncode == 124 and '--launch-timeout' in r.stderr
    assert time.monotonic() - t0 < 60


def test_compact_bench_line_from_a_recorded_full_result():
    """round 3's bench line was 22 959 bytes and reached the driver cut off (parsed: null).  The compact line built from
    that very record must parse, stay under the limit and still carry the contract's fields, `roofline` and `cpu_baseline`."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, 'profiles', 'r03_bench_c4.json')) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    full['detail_file'] = 'bench_detail.json'
    line = json.dumps(bench.compact_line(full))
    assert len(line) < bench.COMPACT_LIMIT <= 4096, len(line)
    out = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in out, k
    assert out['metric'] == full['metric'] and abs(out['value'] - full['value']) < 1e-3 and out['config']['workload'].startswith('c4')
    rf = out['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0 and abs(rf['frac'] - full['roofline']['frac']) < 1e-4
    assert rf['traffic'] == full['roofline']['traffic'] and rf['dominant_kernel']['name'].startswith('spec_mix')
    assert abs(rf['hbm_bound_frac'] - full['roofline']['hbm_bound']['frac']) < 1e-4
    cb = out['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == 16 and cb['value'] > 0 and 'sample' in cb
    assert out['c3']['value'] > 100 and out['split_f16']['fwd_rel_l2'] < 1e-5 and out['dataset']['value'] > 1e4


def test_global_metrics_are_identical_on_all_ranks():
    """callbacks must see the GLOBAL loss.  Two ranks with different local (loss share, mse) get the same pair back."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metric_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:] == res[1][1:]
    assert abs(res[0][1] - 0.75) < 1e-6 and abs(res[0][2] - 3.0) < 1e-6     # loss: sum of shares; mse: mean over ranks
    assert res[0][3] == res[1][3] and res[0][3] < 1e-3                       # ReduceLROnPlateau cut the rate identically on both ranks


def _metric_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from poisson_cnn_amd import parallel
    from poisson_cnn_amd.train import ReduceLROnPlateau
    dp = parallel.DataParallel.from_env(backend='gloo')
    loss, mse = dp.global_metrics(torch.tensor(0.25 + 0.25 * rank), torch.tensor(2.0 + 2.0 * rank))

    class _Opt:
        learning_rate = 1e-3

    class _M:
        optimizer = _Opt()
    cb = ReduceLROnPlateau(patience=2, min_lr=1e-7)
    cb.set_model(_M())
    # rank 0's share falls (1.0, 0.8, 0.6, 0.4), rank 1's rises (1.0, 1.2, 1.4, 1.6): on its own rank 0 would never cut the rate and rank 1
    # would; the GLOBAL loss is flat at 2.0, so both must cut it, at the same epoch
    for e in range(4):
        local = 1.0 + (-0.2 if rank == 0 else 0.2) * e
        gl, _ = dp.global_metrics(torch.tensor(local), torch.tensor(0.0))
        assert abs(float(gl) - 2.0) < 1e-6
        cb.on_epoch_end(e, {'loss': float(gl)})
    q.put((rank, float(loss), float(mse), _M.optimizer.learning_rate))
    torch.distributed.destroy_process_group()


def test_all_ranks_draw_one_grid_shape_per_step():
    """SURVEY section 8(e) / VERDICT r2 #13: every rank must train on the same (H, W) in a step, as the reference's single generator draws one
    shape per global batch (dataset/generators/reverse.py:192-193).  The shape stream is shared, the data stream is per rank, and a rank's
    per-sample grid spacings are its rows of the global batch's draw."""
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.dataset import reverse_poisson_dataset_generator, numerical_dataset_generator, _streams
    d = dict(configs.hpnn()['dataset'])
    d['batch_size'] = 4
    whole = reverse_poisson_dataset_generator(**{**d, 'batch_size': 8}, seed=7, device='cpu', shard=(0, 1))
    ranks = [reverse_poisson_dataset_generator(**d, seed=7, device='cpu', shard=(r, 2)) for r in range(2)]
    shapes = set()
    for step in range(6):
        sw, dxw = whole._shape_and_spacings()
        got = [g._shape_and_spacings() for g in ranks]
        assert tuple(got[0][0]) == tuple(got[1][0]) == tuple(sw)
        assert np.array_equal(np.concatenate([got[0][1], got[1][1]]), dxw)          # the global batch's spacings, split by rank
        shapes.add(tuple(int(v) for v in sw))
    assert len(shapes) > 1 and all(192 <= h <= 384 and 192 <= w <= 384 for h, w in shapes)
    # the data streams differ between ranks, the shape streams do not
    a, b = ranks
    assert a.rng.uniform() != b.rng.uniform() and a.shape_rng.uniform() == b.shape_rng.uniform()
    # no shard: one stream for both (a single process draws exactly as before)
    rng, srng, sh = _streams(3, None)
    assert rng is srng and sh == (0, 1)
    with pytest.raises(ValueError):
        _streams(0, (2, 2))
    # numerical generator: constructor plumbing (its draws need the GPU kernels; covered by tests/test_gpu_dp.py)
    n0 = numerical_dataset_generator(batch_size=2, seed=5, device='cpu', shard=(1, 2), output_shape='random')
    assert n0.shard == (1, 2) and n0.rng is not n0.shape_rng

"""Data-parallel training step on the GPU: 2 ranks (gloo backend, both on cuda:0 - the test box has one GPU) must produce
the same averaged gradient and the same updated weights as one process on the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed_w=3):
    from oracle import hpnn as ohpnn
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.hpnn_tiny()
    model = Homogeneous_Poisson_NN_Legacy(**full['model'])
    model.set_weights(ohpnn.init_params(full['model'], seed=seed_w, gain=1.4, randomize_all=True))
    model.compile(loss=loss_wrapper(global_batch_size=4, **full['training']['loss_parameters']), optimizer=Adam(learning_rate=1e-3))
    return model


def _batch():
    rng = np.random.default_rng(17)
    rhs = rng.uniform(-1, 1, (4, 1, 40, 44)).astype(np.float32)
    dx = rng.uniform(5e-3, 5e-2, (4, 1)).astype(np.float32)
    tgt = (rng.standard_normal((4, 1, 40, 44)) * 0.1).astype(np.float32)
    return rhs, dx, tgt


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    from poisson_cnn_amd import parallel
    dp = parallel.DataParallel.from_env(backend='gloo')
    model = _make(seed_w=3 + rank)          # different initial weights per rank: attach() must replicate rank 0's
    dp.attach(model)
    rhs, dx, tgt = _batch()
    sl = slice(rank * 2, rank * 2 + 2)
    logs = model.train_step(((rhs[sl], dx[sl]), tgt[sl]))
    torch.cuda.synchronize()
    q.put((rank, float(logs['loss']), model.store.flat_g.cpu().numpy().copy(), model.store.flat_w.cpu().numpy().copy()))
    dp.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process():
    ref = _make(seed_w=3)
    rhs, dx, tgt = _batch()
    logs = ref.train_step(((rhs, dx), tgt))
    torch.cuda.synchronize()
    g_ref, w_ref, loss_ref = ref.store.flat_g.cpu().numpy().copy(), ref.store.flat_w.cpu().numpy().copy(), float(logs['loss'])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, g0, w0), (_, l1, g1, w1) = res
    assert np.array_equal(g0, g1) and np.array_equal(w0, w1)                       # ranks agree bit for bit after the all-reduce
    assert l0 == l1 and abs(l0 - loss_ref) < 1e-5 * abs(loss_ref)                   # train_step reports the GLOBAL loss on every rank (sum of the shares)
    assert np.linalg.norm(g0 - g_ref) / np.linalg.norm(g_ref) < 2e-5               # summed shard gradients = full-batch gradient
    assert np.abs(w0 - w_ref).max() < 5e-4 * 1e-3 + 1e-7 or np.linalg.norm(w0 - w_ref) / np.linalg.norm(w_ref) < 1e-5


_RCCL_SNIPPET = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1)       # backend "nccl" IS RCCL on ROCm
from poisson_cnn_amd import parallel
dp = parallel.DataParallel(rank=0, world_size=1, local_rank=0, backend='nccl')
bucket = torch.arange(5556956, dtype=torch.float32, device='cuda') * 1e-3   # the 22.2 MB flat gradient bucket of hpnn.json
ref = bucket.clone()
dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
dist.broadcast(bucket, src=0)
t = torch.tensor([1.25], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(bucket, ref) and float(t) == 1.25
print('RCCL_OK', torch.cuda.nccl.version())
dist.destroy_process_group()
"""


def test_rccl_backend_initialises_and_reduces_the_gradient_bucket():
    """RCCL's load / init / all-reduce path on real hardware (world size 1: the test box has one GPU): the collective the N-GPU bench and
    train.py use (train/hpnn_legacy_train.py:37-41's MirroredStrategy reduction)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_SNIPPET % root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_bench_two_ranks_bare_invocation_on_gpu():
    """`python bench.py --gpus 2` with NO launcher: the parent spawns the ranks itself.  Both ranks share the box's one GPU, so the gradient
    all-reduce goes through gloo here (RCCL refuses two ranks on one device); the JSON must say so."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['PCNN_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--workload', 'small', '--steps', '1', '--warmup', '1',
                        '--no-cpu-baseline', '--no-dataset', '--modes', 'split_f16,fp32'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 4 and out['value'] > 0
    assert 'gloo' in out['config']['collective'] and out['dtype'] == 'f32'
    assert len(lines[0]) < 3072 and out['config']['ranks_seen'] == 2                     # the compact line; both ranks joined the group
    assert out['split_f16']['value'] > 0 and out['split_f16']['fwd_rel_l2'] < 1e-5
    with open(os.path.join(root, out['detail'])) as f:                                  # everything else lives in the detail file
        detail = json.load(f)
    assert detail['split_f16']['accuracy_vs_fp32']['forward_output']['rel_l2'] < 1e-5 and detail['roofline']['bound'] == 'hbm'


def test_c_abi_collective_single_rank():
    """pcnn_comm_unique_id / pcnn_comm_init / pcnn_allreduce / pcnn_broadcast (RCCL bound inside libpcnn, include/pcnn.h) with world size 1:
    RCCL loads, a communicator comes up on the handle's device, the 22.2 MB gradient bucket goes through ncclAllReduce on the handle's stream
    and comes back unchanged; then the same through DataParallel.enable_c_abi_collective() and a model train step."""
    import ctypes
    from poisson_cnn_amd import ops
    from poisson_cnn_amd.parallel import DataParallel
    h = ops.handle()
    probe = torch.zeros(4, device='cuda')
    with pytest.raises(RuntimeError, match='no communicator'):
        h.call('pcnn_allreduce', ctypes.c_void_p(probe.data_ptr()), ctypes.c_size_t(4))
    ident = (ctypes.c_ubyte * 128)()
    h.call('pcnn_comm_unique_id', ident)
    assert any(ident)
    with pytest.raises(RuntimeError, match='rank 3 outside'):
        h.call('pcnn_comm_init', ident, ctypes.c_int(3), ctypes.c_int(2))
    h.call('pcnn_comm_init', ident, ctypes.c_int(0), ctypes.c_int(1))
    with pytest.raises(RuntimeError, match='already has a communicator'):
        h.call('pcnn_comm_init', ident, ctypes.c_int(0), ctypes.c_int(1))
    g = torch.Generator(device='cuda').manual_seed(0)
    flat = torch.randn(5_560_000, device='cuda', generator=g)
    ref = flat.clone()
    h.call('pcnn_allreduce', ctypes.c_void_p(flat.data_ptr()), ctypes.c_size_t(flat.numel()))
    h.call('pcnn_broadcast', ctypes.c_void_p(flat.data_ptr()), ctypes.c_size_t(flat.numel()), ctypes.c_int(0))
    torch.cuda.synchronize()
    assert torch.equal(flat, ref)
    h.call('pcnn_comm_destroy')
    # the same path behind the trainer's hook
    dp = DataParallel(0, 1, 0, None).enable_c_abi_collective()
    assert 'pcnn_allreduce' in dp.collective_name()
    m1, m2 = _make(), dp.attach(_make())
    rhs, dx, tgt = _batch()
    l1 = m1.train_step(((rhs, dx), tgt))
    l2 = m2.train_step(((rhs, dx), tgt))
    torch.cuda.synchronize()
    assert float(l1['loss']) == float(l2['loss'])
    assert torch.equal(m1.store.flat_w, m2.store.flat_w)
    h.call('pcnn_comm_destroy')


def _train_main_worker(rank, world, port, cfg_path, ckpt_dir, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', PCNN_DIST_BACKEND='gloo')
    try:
        from poisson_cnn_amd import dataset as D, models as M, train as T
        shapes, captured = [], {}
        orig_get = D.reverse_poisson_dataset_generator.__getitem__     # train.main imports the class from the package at call time

        def getitem(self, idx=0):
            inp, soln = orig_get(self, idx)
            shapes.append((tuple(soln.shape), float(inp[0].double().abs().sum())))
            return inp, soln
        D.reverse_poisson_dataset_generator.__getitem__ = getitem
        orig_fit = M.Homogeneous_Poisson_NN_Legacy.fit

        def fit(self, *a, **kw):
            captured['model'] = self
            return orig_fit(self, *a, **kw)
        M.Homogeneous_Poisson_NN_Legacy.fit = fit
        os.makedirs(os.path.join(ckpt_dir, 'r%d' % rank), exist_ok=True)
        T.main([cfg_path, '--epochs', '1', '--checkpoint_dir', os.path.join(ckpt_dir, 'r%d' % rank)])
        torch.cuda.synchronize()
        m = captured['model']
        q.put((rank, shapes, m.store.flat_w.cpu().numpy().copy()))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except BaseException as e:       # a crashed rank must fail the test at once, not leave the parent waiting on the queue
        q.put((rank, 'error: %r' % (e,), None))
        raise


def test_train_main_two_ranks_share_the_grid_shape_and_the_weights(tmp_path):
    """VERDICT r2 #13 / SURVEY 8(e): `python -m poisson_cnn_amd.train cfg.json` under two ranks with hpnn.json-style RANDOM grid shapes: every step
    has one (H, W) on both ranks (drawn from the shared shape stream) while the samples differ, and after three optimizer steps the weights
    are bit-equal on both ranks (identical all-reduced gradient, identical Adam step)."""
    import json
    from poisson_cnn_amd import configs
    cfg = configs.hpnn_tiny()
    cfg['dataset'].update(batch_size=4, batches_per_epoch=3, random_output_shape_range=[[40, 72], [40, 72]])
    cfg['training']['loss_parameters']['integral_loss_config']['n_quadpts'] = 11
    path = str(tmp_path / 'cfg.json')
    with open(path, 'w') as f:
        json.dump(cfg, f)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_main_worker, args=(r, 2, port, path, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    assert not any(isinstance(r[1], str) for r in res), res
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, s0, w0), (_, s1, w1) = res
    assert len(s0) == len(s1) == 3
    assert [a[0] for a in s0] == [b[0] for b in s1]                 # one grid shape per step on all ranks ...
    assert all(a[0][0] == 2 for a in s0)                            # ... each holding its half of the global batch
    assert len({a[0] for a in s0}) > 1                              # and the shape does change from step to step
    assert all(a[1] != b[1] for a, b in zip(s0, s1))                # different samples per rank
    assert np.array_equal(w0, w1) and np.isfinite(w0).all()


_CABI2_SNIPPET = """
import json, os, sys, torch
sys.path.insert(0, %r)
torch.cuda.set_device(0)
from poisson_cnn_amd import parallel
dp = parallel.DataParallel.from_env(backend='gloo')               # rendezvous over gloo: RANK / WORLD_SIZE / MASTER_* from the environment
ident = dp.exchange_unique_id()                                    # rank 0: pcnn_comm_unique_id; everybody: the broadcast 128 bytes
out = {'rank': dp.rank, 'ranks_seen': dp.ranks_seen(), 'ident': ident, 'init': None, 'sum': None}
try:
    import ctypes
    from poisson_cnn_amd import ops
    h = ops.handle()
    h.call('pcnn_comm_init', (ctypes.c_ubyte * 128)(*ident), dp.rank, dp.world_size)
    out['init'] = 'ok'
    buf = torch.full((1 << 20,), float(dp.rank + 1), device='cuda')
    h.call('pcnn_allreduce', ctypes.c_void_p(buf.data_ptr()), buf.numel())
    torch.cuda.synchronize()
    out['sum'] = [float(buf.min()), float(buf.max())]
    h.call('pcnn_comm_destroy')
except RuntimeError as e:
    out['init'] = str(e)
print('CABI2 ' + json.dumps(out), flush=True)
dp.barrier()
torch.distributed.destroy_process_group()
"""


def test_c_abi_collective_rendezvous_with_two_processes():
    """the C-ABI collective's rendezvous at world size 2 - two bare processes, gloo between them, both on this box's one GPU.
    Rank 0's RCCL unique id (pcnn_comm_unique_id) must reach rank 1 unchanged (DataParallel.exchange_unique_id), and both ranks must come out of
    pcnn_comm_init TOGETHER: either with a communicator - then pcnn_allreduce of (1, 2) gives 3 everywhere - or, because RCCL refuses two ranks on
    one device, each with the library's ordinary error return (no hang, no crash).  The first real 2-GPU all-reduce happens on the driver's node."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK='0', WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0',
                   PCNN_DIST_TIMEOUT_S='120', NCCL_DEBUG='WARN')
        procs.append(subprocess.Popen([sys.executable, '-c', _CABI2_SNIPPET % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, (o[-1000:], e[-3000:])
            line = [ln for ln in o.splitlines() if ln.startswith('CABI2 ')]
            assert len(line) == 1, (o[-1000:], e[-2000:])
            outs.append(json.loads(line[0][6:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs.sort(key=lambda d: d['rank'])
    assert [d['rank'] for d in outs] == [0, 1] and all(d['ranks_seen'] == 2 for d in outs)
    assert outs[0]['ident'] == outs[1]['ident'] and any(outs[0]['ident'])            # the id crossed the process boundary intact
    if outs[0]['init'] == 'ok':
        assert outs[1]['init'] == 'ok' and outs[0]['sum'] == [3.0, 3.0] and outs[1]['sum'] == [3.0, 3.0]
    else:                                                                              # RCCL's refusal of a duplicate device: an error return on both ranks
        assert all('pcnn_comm_init' in d['init'] for d in outs), outs
        print('two ranks on one device: %s' % outs[0]['init'][:200])


_CABI2DEV_SNIPPET = """
import json, os, sys, torch
sys.path.insert(0, %r)
rank = int(os.environ['RANK'])
torch.cuda.set_device(rank)                                        # one device per rank: RCCL has no reason to refuse
from poisson_cnn_amd import parallel
import torch.distributed as dist
dp = parallel.DataParallel.from_env(backend='nccl')
dp.enable_c_abi_collective()                                       # must succeed here: an exception fails the rank
n = 5556956                                                        # hpnn.json's flat gradient bucket: 22.2 MB
g = torch.Generator(device='cuda').manual_seed(11 + rank)
a = torch.randn(n, device='cuda', generator=g)
b = a.clone()
dp.all_reduce_sum(a)                                               # pcnn_allreduce (RCCL bound inside libpcnn)
dist.all_reduce(b, op=dist.ReduceOp.SUM)                           # torch.distributed's RCCL
torch.cuda.synchronize()
print('CABI2DEV ' + json.dumps({'rank': rank, 'ranks_seen': dp.ranks_seen(), 'name': dp.collective_name(), 'equal': bool(torch.equal(a, b)),
                                'finite': bool(torch.isfinite(a).all()), 'moved': bool((a != 0).any())}), flush=True)
dp.barrier()
dist.destroy_process_group()
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: on a one-GPU box the rendezvous test above covers what can be covered')
def test_c_abi_allreduce_on_two_devices_equals_torch_distributed_bit_for_bit():
    """with two devices there is no excuse - both ranks MUST obtain a communicator through the C-ABI rendezvous, reduce the real
    22.2 MB gradient bucket with pcnn_allreduce and get exactly the bits torch.distributed's RCCL all-reduce gives (two ranks: one addition per
    element, no ordering freedom)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0',
                   PCNN_DIST_TIMEOUT_S='120', NCCL_DEBUG='WARN')
        procs.append(subprocess.Popen([sys.executable, '-c', _CABI2DEV_SNIPPET % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=300)
            assert p.returncode == 0, (o[-1000:], e[-3000:])
            line = [ln for ln in o.splitlines() if ln.startswith('CABI2DEV ')]
            assert len(line) == 1, (o[-1000:], e[-2000:])
            outs.append(json.loads(line[0][9:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for d in outs:
        assert d['ranks_seen'] == 2 and 'C-ABI' in d['name'] and d['equal'] and d['finite'] and d['moved'], outs


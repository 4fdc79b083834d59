"""GPU parity of Dirichlet_BC_NN_Legacy_2: forward vs the fp64 numpy oracle, every gradient of a full
training step vs the fp64 autograd twin, and the new kernels on their own."""
import numpy as np
import pytest
import torch

from oracle import dbcnn as odb, np_ops, torch_twin, loss as oloss
from poisson_cnn_amd import configs

pytestmark = pytest.mark.gpu
TOL_FWD = 1e-5


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def f32(a):
    return np.asarray(a).astype(np.float32).astype(np.float64)


def make_inputs(N, L, seed):
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 1, L)
    bc = sum(rng.standard_normal((N, 1, 1)) * np.sin((k + 1) * np.pi * t + rng.uniform(0, 3, (N, 1, 1))) for k in range(5))
    bc /= np.abs(bc).max(axis=2, keepdims=True)
    return f32(bc), f32(rng.uniform(5e-3, 5e-2, (N, 1)))


def build(cfg, seed, gain=1.5):
    from poisson_cnn_amd.models import Dirichlet_BC_NN_Legacy_2
    model = Dirichlet_BC_NN_Legacy_2(**cfg)
    p = odb.init_params(cfg, seed=seed, gain=gain, randomize_all=True)
    assert list(p.keys()) == model.weight_names
    model.set_weights(p)
    return model, p


@pytest.mark.parametrize('L,X,post', [(40, 37, 0), (61, 48, 2)])
def test_tiny_forward(L, X, post):
    cfg = configs.dbcnn_tiny()['model']
    cfg['postsmoother_iterations'] = post
    model, p = build(cfg, 3)
    bc, dx = make_inputs(3, L, 5)
    ref = odb.forward(np_ops, cfg, p, bc, dx, X)
    y = model([bc, dx, X]).cpu().numpy()
    assert y.shape == ref.shape == (3, 1, X, L)
    assert np.array_equal(y[:, :, 0, :], bc.astype(np.float32))          # the first row IS the boundary condition
    assert rel(y, ref) < TOL_FWD


def test_dbcnn_json_forward():
    """The shipped dbcnn.json model (k = 19 ... 5 boundary convolutions with batch norm, 123-bin pyramid, 27 sinh modes)."""
    cfg = configs.dbcnn()['model']
    model, p = build(cfg, 7, gain=1.3)
    bc, dx = make_inputs(2, 200, 9)
    ref = odb.forward(np_ops, cfg, p, bc, dx, 96)
    y = model([bc, dx, 96]).cpu().numpy()
    assert np.isfinite(y).all()
    assert rel(y, ref) < TOL_FWD


def _train_reference(cfg, p, bc, dx, target, lossp, gbs):
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=not k.endswith(('moving_mean', 'moving_variance'))) for k, v in p.items()}
    pred = odb.forward(torch_twin, cfg, pt, torch.tensor(bc), torch.tensor(dx), target.shape[2])
    Lf = oloss.loss_wrapper(global_batch_size=gbs, **lossp)
    loss = Lf(target, pred, torch.zeros_like(pred), np.concatenate([dx, dx], 1))
    loss.backward()
    return float(loss.detach()), pred.detach().numpy(), {k: v.grad.numpy() for k, v in pt.items() if v.requires_grad}


@pytest.mark.parametrize('post', [0, 1])
def test_tiny_train_step(post):
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.dbcnn_tiny()
    cfg = full['model']
    cfg['postsmoother_iterations'] = post
    lossp = dict(full['training']['loss_parameters'])
    model, p = build(cfg, 13)
    bc, dx = make_inputs(3, 44, 15)
    target = f32(np.random.default_rng(2).standard_normal((3, 1, 39, 44)) * 0.3)
    ref_loss, ref_pred, ref_g = _train_reference(cfg, p, bc, dx, target, lossp, 3)
    model.compile(loss=loss_wrapper(global_batch_size=3, **lossp), optimizer=Adam(learning_rate=1e-3))
    logs = model.train_step(((bc, dx), target))
    assert abs(float(logs['loss']) - ref_loss) < 2e-5 * abs(ref_loss)
    g = {n: model.store.g[n].cpu().numpy() for n in model.store.trainable_names()}
    flat = np.concatenate([g[n].ravel() for n in g]); flat_ref = np.concatenate([ref_g[n].ravel() for n in g])
    assert rel(flat, flat_ref) < 3e-4
    for n in g:
        assert rel(g[n], ref_g[n]) < 3e-3, n


def test_max_pyramid_pooling_forward_and_train_step():
    """spp_config pooling_type = "max" (layers/SpatialPyramidPool.py:17-24; the shipped config uses "average"): forward and every gradient."""
    from poisson_cnn_amd.losses import loss_wrapper
    from poisson_cnn_amd.train import Adam
    full = configs.dbcnn_tiny()
    cfg = full['model']
    cfg['spp_config'] = dict(cfg['spp_config'], pooling_type='max')
    lossp = dict(full['training']['loss_parameters'])
    model, p = build(cfg, 17)
    bc, dx = make_inputs(3, 44, 19)
    assert rel(model([bc, dx, 39]).cpu().numpy(), odb.forward(np_ops, cfg, p, bc, dx, 39)) < TOL_FWD
    target = f32(np.random.default_rng(3).standard_normal((3, 1, 39, 44)) * 0.3)
    ref_loss, ref_pred, ref_g = _train_reference(cfg, p, bc, dx, target, lossp, 3)
    model.compile(loss=loss_wrapper(global_batch_size=3, **lossp), optimizer=Adam(learning_rate=1e-3))
    logs = model.train_step(((bc, dx), target))
    assert abs(float(logs['loss']) - ref_loss) < 2e-5 * abs(ref_loss)
    g = {n: model.store.g[n].cpu().numpy() for n in model.store.trainable_names()}
    flat = np.concatenate([g[n].ravel() for n in g]); flat_ref = np.concatenate([ref_g[n].ravel() for n in g])
    assert rel(flat, flat_ref) < 3e-4


def test_set_max_magnitude_bwd_and_expand_bwd():
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(0)
    x = f32(rng.standard_normal((3, 17, 11)))
    x[1, 4, 2] = -x[1].__abs__().max() * 1.0      # a negative maximum
    x[2, 0, 0] = x[2, 3, 3] = np.abs(x[2]).max()  # a tie: the gradient of the maximum is split
    g = f32(rng.standard_normal(x.shape))
    xt = torch.tensor(x, requires_grad=True)
    (torch_twin.set_max_magnitude_in_batch(xt, 1.0) * torch.tensor(g)).sum().backward()
    xd, gd = torch.tensor(x, dtype=torch.float32).cuda(), torch.tensor(g, dtype=torch.float32).cuda()
    y, fac = ops.set_max_magnitude_fwd(xd, 1.0)
    assert rel(y.cpu().numpy(), np_ops.set_max_magnitude_in_batch(x, 1.0)) < 1e-6
    assert rel(ops.set_max_magnitude_bwd(xd, gd, 1.0).cpu().numpy(), xt.grad.numpy()) < 2e-6
    # einsum expansion and its adjoint
    N, Lh, M, X = 2, 13, 5, 9
    f = f32(rng.standard_normal((N, M, Lh))); d = f32(rng.standard_normal((N, M))); sh = f32(odb.sinh_basis(M, X))
    go = f32(rng.standard_normal((N, M + 2, X, Lh)))
    ft, dt = torch.tensor(f, requires_grad=True), torch.tensor(d, requires_grad=True)
    out = torch.einsum('bmy,mx,bm->bmxy', ft, torch.tensor(sh), dt)
    (out * torch.tensor(go[:, :M])).sum().backward()
    fd = torch.tensor(f.transpose(0, 2, 1)[:, None], dtype=torch.float32).contiguous().cuda()      # (N,1,L,M)
    dd_, shd = torch.tensor(d, dtype=torch.float32).cuda(), torch.tensor(sh, dtype=torch.float32).cuda()
    o = ops.dbc_expand_fwd(fd, shd, dd_).cpu().numpy()                                                # (N,X,L,M+2)
    assert rel(o[..., :M], out.detach().numpy().transpose(0, 2, 3, 1)) < 1e-6
    assert rel(o[..., M:], odb.position_embeddings(N, X, Lh).transpose(0, 2, 3, 1)) < 1e-6
    god = torch.tensor(go.transpose(0, 2, 3, 1), dtype=torch.float32).contiguous().cuda()
    df, ddg = ops.dbc_expand_bwd(god, fd, shd, dd_)
    assert rel(df.cpu().numpy()[:, 0].transpose(0, 2, 1), ft.grad.numpy()) < 2e-6
    assert rel(ddg.cpu().numpy(), dt.grad.numpy()) < 2e-6

"""GPU parity: fused pad+conv+epilogue HIP kernel (through the C-ABI) vs the fp64 numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import np_ops

pytestmark = pytest.mark.gpu
TOL = 2e-6   # per-layer rel-L2 bound (fp32 MFMA fmaf chain vs fp64); north-star end-to-end bound is 1e-5


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def nhwc(a):
    return torch.tensor(np.ascontiguousarray(a.transpose(0, 2, 3, 1)), dtype=torch.float32, device='cuda')


def nchw(t):
    return t.detach().cpu().numpy().transpose(0, 3, 1, 2)


CASES = [
    # k, Cin, Cout, H, W, mode, act
    (3, 8, 8, 20, 37, 'CONSTANT', 'leaky_relu'),
    (3, 4, 1, 33, 31, 'CONSTANT', 'linear'),
    (5, 32, 32, 16, 32, 'CONSTANT', 'leaky_relu'),
    (7, 64, 32, 19, 45, 'CONSTANT', 'leaky_relu'),
    (15, 3, 4, 40, 50, 'SYMMETRIC', 'leaky_relu'),
    (13, 4, 16, 35, 33, 'SYMMETRIC', 'leaky_relu'),
    (11, 16, 32, 17, 64, 'SYMMETRIC', 'tanh'),
    (13, 32, 28, 30, 41, 'CONSTANT', 'leaky_relu'),
    (9, 28, 24, 25, 36, 'REFLECT', 'leaky_relu'),
    (7, 24, 20, 18, 70, 'CONSTANT', 'relu'),
    (5, 20, 16, 48, 16, 'CONSTANT', 'leaky_relu'),
    (3, 12, 12, 5, 7, 'SYMMETRIC', 'leaky_relu'),
    (15, 32, 32, 33, 47, 'CONSTANT', 'leaky_relu'),
    (4, 6, 5, 13, 14, 'CONSTANT', 'linear'),
    (5, 32, 32, 1, 1, 'CONSTANT', 'leaky_relu'),
]


@pytest.mark.parametrize('k,Cin,Cout,H,W,mode,act', CASES)
def test_padded_conv_matches_oracle(k, Cin, Cout, H, W, mode, act):
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(k * 1000 + Cin * 10 + Cout)
    N = 2
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = np_ops.padded_conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), mode, 0.3, act)
    pb, _ = np_ops.advanced_pad_amounts(k)
    y = ops.conv2d_fwd(nhwc(x), torch.tensor(w, device='cuda'), torch.tensor(b, device='cuda'), pad_top=pb, pad_left=pb,
                       pad_mode=mode, pad_value=0.3, act=act)
    torch.cuda.synchronize()
    assert rel(nchw(y), ref) < TOL


def test_conv_epilogue_bn_residual_slices():
    """BN affine + residual + act_out, reading and writing channel slices of wider buffers."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(5)
    N, H, W, C = 2, 21, 35, 32
    xb = torch.tensor(rng.standard_normal((N, H, W, 64)), dtype=torch.float32, device='cuda')
    x = xb[..., 32:]
    w = torch.tensor(rng.standard_normal((7, 7, C, C)) / 40, dtype=torch.float32, device='cuda')
    b = torch.tensor(rng.standard_normal(C), dtype=torch.float32, device='cuda')
    sc = torch.tensor(rng.uniform(0.5, 1.5, C), dtype=torch.float32, device='cuda')
    sh = torch.tensor(rng.standard_normal(C), dtype=torch.float32, device='cuda')
    res = torch.tensor(rng.standard_normal((N, H, W, C)), dtype=torch.float32, device='cuda')
    outb = torch.zeros((N, H, W, 64), dtype=torch.float32, device='cuda')
    a_out = torch.empty((N, H, W, C), dtype=torch.float32, device='cuda')
    ops.conv2d_fwd(x, w, b, pad_top=3, pad_left=3, pad_mode='SYMMETRIC', act='leaky_relu', bn_scale=sc, bn_shift=sh,
                   residual=res, out=outb[..., :32], act_out=a_out)
    torch.cuda.synchronize()
    xn = x.cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)
    a = np_ops.padded_conv2d(xn, w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64), 'SYMMETRIC', 0.0, 'leaky_relu')
    ref = a * sc.cpu().numpy()[None, :, None, None] + sh.cpu().numpy()[None, :, None, None] + res.cpu().numpy().transpose(0, 3, 1, 2)
    assert rel(nchw(a_out), a) < TOL
    assert rel(nchw(outb[..., :32]), ref) < TOL
    assert float(outb[..., 32:].abs().max()) == 0.0


def test_flip_transpose_and_data_gradient():
    """dgrad = conv of dz (zero padded) with the flipped/transposed filter: check against the oracle's adjoint."""
    from poisson_cnn_amd import ops
    rng = np.random.default_rng(9)
    N, H, W, Cin, Cout, k = 1, 18, 23, 12, 20, 5
    w = rng.standard_normal((k, k, Cin, Cout)).astype(np.float32)
    wt = ops.flip_transpose_weights(torch.tensor(w, device='cuda')).cpu().numpy()
    assert np.array_equal(wt, w[::-1, ::-1].transpose(0, 1, 3, 2))
    dz = rng.standard_normal((N, Cout, H, W)).astype(np.float32)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    # <conv(x), dz> == <x, dgrad(dz)> for the zero-padded conv
    dx = ops.conv2d_fwd(nhwc(dz), torch.tensor(wt, device='cuda'), None, pad_top=k - 1 - k // 2, pad_left=k - 1 - k // 2)
    torch.cuda.synchronize()
    y = np_ops.padded_conv2d(x.astype(np.float64), w.astype(np.float64), None, 'CONSTANT', 0.0, 'linear')
    lhs = float((y * dz).sum()); rhs = float((nchw(dx).astype(np.float64) * x).sum())
    assert abs(lhs - rhs) < 1e-5 * abs(lhs)


def test_error_behaviour_of_the_c_abi():
    """Bad arguments fail loudly through pcnn_last_error (RuntimeError in the shim), never silently: unsupported channel counts, padding
    beyond what tf.pad allows, an undersized workspace, a null pointer."""
    import ctypes
    from poisson_cnn_amd import ops
    x = torch.randn(1, 20, 20, 4, device='cuda')
    with pytest.raises(RuntimeError, match='Cout'):
        ops.conv2d_fwd(x, torch.randn(3, 3, 4, 80, device='cuda'), None, pad_top=1, pad_left=1)
    with pytest.raises(RuntimeError, match='padding exceeds'):
        ops.conv2d_fwd(torch.randn(1, 4, 4, 4, device='cuda'), torch.randn(15, 15, 4, 4, device='cuda'), None, pad_top=7, pad_left=7, pad_mode='REFLECT')
    d = ops.conv_desc(x.shape, 4, (3, 3, 4, 4), (20, 20), 4, 1, 1)
    dz, dw, ws = torch.randn(1, 20, 20, 4, device='cuda'), torch.empty(3, 3, 4, 4, device='cuda'), torch.empty(16, device='cuda')
    with pytest.raises(RuntimeError, match='workspace too small'):
        ops.handle().call('pcnn_conv2d_wgrad', ctypes.byref(d), ops._p(x), ops._p(dz), ops._p(dw), ops._p(ws), ctypes.c_size_t(64))
    with pytest.raises(RuntimeError, match='null'):
        ops.handle().call('pcnn_conv2d_wgrad', ctypes.byref(d), ops._p(None), ops._p(dz), ops._p(dw), ops._p(ws), ctypes.c_size_t(64))
    # the handle stays usable after an error
    y = ops.conv2d_fwd(x, torch.randn(3, 3, 4, 4, device='cuda'), None, pad_top=1, pad_left=1)
    assert torch.isfinite(y).all()

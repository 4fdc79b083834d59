"""Pin the oracle on the only reference-authored known answers that exist for this path."""
import json
import os

import numpy as np

from oracle import np_ops, loss as oloss

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_fd_coefficients_golden():
    # tests/golden/fd_coefficients.json was produced by importing the reference's own
    # dataset/utils/get_fd_coefficients.py (numpy+scipy only) - see tests/golden/make_fd_golden.py
    with open(os.path.join(GOLD, 'fd_coefficients.json')) as f:
        cases = json.load(f)
    assert len(cases) >= 6
    for c in cases:
        got = np_ops.get_fd_coefficients(c['stencil_positions'], c['order'])
        assert np.allclose(got, c['coefficients'], rtol=1e-9, atol=1e-11), c
    st = np_ops.build_fd_coefficients([3, 3], [2, 2], 2)
    assert np.allclose(st[0], [[0, 1, 0], [0, -2, 0], [0, 1, 0]]) and np.allclose(st[1], [[0, 0, 0], [1, -2, 1], [0, 0, 0]])


def test_split_indices_docstring_value():
    assert list(np_ops.split_indices(229, 4)) == [0, 58, 115, 172, 229]   # dataset/utils/split_indices.py:13


def test_integral_loss_known_answer():
    # losses/integral_loss.py:181-203: integrate (xyz)^(2/3) over [0,1]x[0,2]x[1,3.5] = 4.84711 within 1 %
    x = np.linspace(0.0, 1.0, 150); y = np.linspace(0.0, 2.0, 200); z = np.linspace(1.0, 3.5, 175)
    t = np.einsum('i,j,k->ijk', x, y, z) ** (1 / 3)
    t = t[None, None]
    dx = np.array([[x[1] - x[0], y[1] - y[0], z[1] - z[0]]])
    val = oloss.integral_lp(t, np.zeros_like(t), (25, 13, 28), p=2, dx=dx)
    assert val.shape == (1, 1)
    assert abs(val[0, 0] - 4.84711) / 4.84711 < 0.01


def test_oracle_reproduces_committed_model_vectors():
    """The CPU oracle against the committed forward fixtures (tiny model; the full hpnn case is checked on the GPU box)."""
    from oracle import hpnn as ohpnn
    from poisson_cnn_amd import configs
    g = np.load(os.path.join(GOLD, 'hpnn_forward_golden.npz'))
    cfg = configs.hpnn_tiny()['model']
    cfg['bc_type'] = 'neumann'
    p = ohpnn.init_params(cfg, seed=5, gain=1.6, randomize_all=True)
    y = ohpnn.forward(np_ops, cfg, p, g['tiny_rhs'].astype(np.float64), g['tiny_dx'].astype(np.float64))
    assert np.linalg.norm(y - g['tiny_out']) / np.linalg.norm(g['tiny_out']) < 1e-6

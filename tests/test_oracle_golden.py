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


# ---- Dirichlet_BC_NN_Legacy_2 / Poisson_CNN_Legacy oracle (oracle/dbcnn.py)
def test_flip_and_rotate_matches_rot90():
    """flip_and_rotate_tensor builds rotations from a transpose + axis reversals; its own __main__ demo
    (dataset/utils/flip_and_rotate_tensor.py:49-61) compares against tf.image.rot90 - here numpy's rot90 is the known answer."""
    from oracle import dbcnn as odb
    x = np.arange(2 * 1 * 4 * 5, dtype=np.float64).reshape(2, 1, 4, 5)
    for k in (1, 2, 3):
        assert np.array_equal(odb.flip_and_rotate(np_ops, x, rotation_count=k), np.rot90(x, k, axes=(2, 3)))
    assert np.array_equal(odb.flip_and_rotate(np_ops, x, rotation_count=0, flip_axes=(2,)), x[:, :, ::-1, :])
    assert np.array_equal(odb.flip_and_rotate(np_ops, x, rotation_count=1, flip_axes=(2,)), np.transpose(x, (0, 1, 3, 2)))


def test_dbcnn_oracle_structure_and_invariants():
    import json
    from oracle import dbcnn as odb
    from poisson_cnn_amd import configs
    cfg = configs.dbcnn()['model']
    meta, spec = odb.build_structure(cfg)
    # experiments/dbcnn.json: 8 boundary stages (conv + BN + 3-conv resnet with 2 BN), 3 dense layers, 6 final stages + 2 tail convs
    assert len([n for n, _, _ in spec if n.endswith('/kernel')]) == 8 * 4 + 3 + 6 * 4 + 2
    assert dict((n, s) for n, s, _ in spec)['bc/stage0/conv/kernel'] == (1, 19, 3, 2)
    assert dict((n, s) for n, s, _ in spec)['mlp/dense0/kernel'] == (3 + 123, 512)
    sh = odb.sinh_basis(27, 64)
    assert np.allclose(np.abs(sh).max(axis=1), 1.0) and np.allclose(sh[:, -1], 0.0)       # sinh(m pi (x - 1)) vanishes at x = 1
    tiny = configs.dbcnn_tiny()['model']
    p = odb.init_params(tiny, seed=0, randomize_all=True)
    rng = np.random.default_rng(1)
    bc, dx = rng.standard_normal((2, 1, 30)), rng.uniform(5e-3, 5e-2, (2, 1))
    y = odb.forward(np_ops, tiny, p, bc, dx, 25)
    assert y.shape == (2, 1, 25, 30) and np.array_equal(y[:, :, 0, :], bc)
    assert np.all(np.abs(y[:, :, 1:, :]) <= 1.0 + 1e-12)                                   # set_max_magnitude_in_batch(out, 1.0)


def test_dbcnn_golden_fixture_is_reproduced_by_the_oracle():
    """tests/golden/dbcnn_forward_golden.npz (make_dbcnn_golden.py): the composite-model vectors are cheap enough to recompute on the CPU."""
    from oracle import dbcnn as odb, hpnn as ohpnn
    from poisson_cnn_amd import configs
    g = np.load(os.path.join(GOLD, 'dbcnn_forward_golden.npz'))
    hcfg, dcfg = configs.hpnn_tiny()['model'], configs.dbcnn_tiny()['model']
    hp, dp = ohpnn.init_params(hcfg, seed=3, gain=1.5, randomize_all=True), odb.init_params(dcfg, seed=4, gain=1.5, randomize_all=True)
    inp = [g['pcnn_' + k].astype(np.float64) for k in ('rhs', 'left', 'top', 'right', 'bottom', 'dx')]
    y = odb.pcnn_forward(np_ops, hcfg, hp, dcfg, dp, *inp)
    assert np.linalg.norm(y - g['pcnn_out']) / np.linalg.norm(g['pcnn_out']) < 1e-6

"""Generates tests/golden/hpnn_forward_golden.npz: inputs and fp64 ORACLE outputs (oracle/hpnn.py on oracle/np_ops.py) of the
shipped hpnn.json model (Keras-default weights from oracle.hpnn.init_params, seed 11, gain 1.6, randomised biases / BN) on a
112 x 120 Dirichlet grid, and of the reduced `hpnn_tiny` model on 53 x 47 (Neumann).  These are regression vectors of the
CPU oracle (the TensorFlow reference cannot run in the build container - DESIGN.md section 2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hpnn as ohpnn, np_ops  # noqa: E402
from poisson_cnn_amd import configs  # noqa: E402


def inputs(N, H, W, seed):
    rng = np.random.default_rng(seed)
    rhs = rng.uniform(-1, 1, (N, 1, H, W))
    rhs /= np.abs(rhs).max(axis=(1, 2, 3), keepdims=True)
    dx = rng.uniform(5e-3, 5e-2, (N, 1))
    return rhs.astype(np.float32), dx.astype(np.float32)


out = {}
cfg = configs.hpnn()['model']
p = ohpnn.init_params(cfg, seed=11, gain=1.6, randomize_all=True)
rhs, dx = inputs(1, 112, 120, 13)
out['hpnn_rhs'], out['hpnn_dx'] = rhs, dx
out['hpnn_out'] = ohpnn.forward(np_ops, cfg, p, rhs.astype(np.float64), dx.astype(np.float64)).astype(np.float32)
cfg = configs.hpnn_tiny()['model']
cfg['bc_type'] = 'neumann'
p = ohpnn.init_params(cfg, seed=5, gain=1.6, randomize_all=True)
rhs, dx = inputs(2, 53, 47, 7)
out['tiny_rhs'], out['tiny_dx'] = rhs, dx
out['tiny_out'] = ohpnn.forward(np_ops, cfg, p, rhs.astype(np.float64), dx.astype(np.float64)).astype(np.float32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hpnn_forward_golden.npz'), **out)
print({k: v.shape for k, v in out.items()})

"""Generates tests/golden/dbcnn_forward_golden.npz: inputs and fp64 ORACLE outputs (oracle/dbcnn.py on oracle/np_ops.py) of the shipped
dbcnn.json model (Keras-default weights from oracle.dbcnn.init_params, seed 7, gain 1.3, randomised biases / BN) on a 96 x 200 domain,
and of the composite Poisson_CNN_Legacy built from the reduced `hpnn_tiny` / `dbcnn_tiny` models on a 44 x 38 grid.  Regression
vectors of the CPU oracle (the TensorFlow reference cannot run in the build container - DESIGN.md section 2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dbcnn as odb, hpnn as ohpnn, np_ops  # noqa: E402
from poisson_cnn_amd import configs  # noqa: E402


def f32(a):
    return np.asarray(a).astype(np.float32)


out = {}
rng = np.random.default_rng(9)
t = np.linspace(0, 1, 200)
bc = sum(rng.standard_normal((2, 1, 1)) * np.sin((k + 1) * np.pi * t + rng.uniform(0, 3, (2, 1, 1))) for k in range(5))
bc = f32(bc / np.abs(bc).max(axis=2, keepdims=True))
dx = f32(rng.uniform(5e-3, 5e-2, (2, 1)))
cfg = configs.dbcnn()['model']
p = odb.init_params(cfg, seed=7, gain=1.3, randomize_all=True)
out['dbcnn_bc'], out['dbcnn_dx'] = bc, dx
out['dbcnn_out'] = f32(odb.forward(np_ops, cfg, p, bc.astype(np.float64), dx.astype(np.float64), 96))

hcfg, dcfg = configs.hpnn_tiny()['model'], configs.dbcnn_tiny()['model']
hp, dp = ohpnn.init_params(hcfg, seed=3, gain=1.5, randomize_all=True), odb.init_params(dcfg, seed=4, gain=1.5, randomize_all=True)
N, H, W = 2, 44, 38
rhs = f32(rng.uniform(-2, 2, (N, 1, H, W)))
edges = [f32(np.cumsum(rng.standard_normal((N, 1, n)), axis=2) * 0.2) for n in (W, H, W, H)]
dx2 = f32(rng.uniform(5e-3, 5e-2, (N, 1)))
for name, v in zip(('rhs', 'left', 'top', 'right', 'bottom', 'dx'), [rhs] + edges + [dx2]):
    out['pcnn_' + name] = v
out['pcnn_out'] = f32(odb.pcnn_forward(np_ops, hcfg, hp, dcfg, dp, *[v.astype(np.float64) for v in [rhs] + edges + [dx2]]))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dbcnn_forward_golden.npz'), **out)
print({k: v.shape for k, v in out.items()})

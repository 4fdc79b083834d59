"""Where does fp32 rounding error accumulate along the forward graph? (diagnostic, GPU box)"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import hpnn as ohpnn, np_ops
from poisson_cnn_amd import configs
from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


for gain, rand in ((1.6, True), (1.0, False), (1.3, True)):
    cfg = configs.hpnn()['model']
    model = Homogeneous_Poisson_NN_Legacy(**cfg)
    p = ohpnn.init_params(cfg, seed=11, gain=gain, randomize_all=rand)
    model.set_weights(p)
    rng = np.random.default_rng(13)
    rhs = rng.uniform(-1, 1, (1, 1, 112, 120)).astype(np.float32).astype(np.float64)
    dx = rng.uniform(5e-3, 5e-2, (1, 1)).astype(np.float32).astype(np.float64)
    taps = {}
    ref = ohpnn.forward(np_ops, cfg, p, rhs, dx, taps=taps)
    y = model.call([rhs, dx], training=True)
    sv = model._saved
    nchw = lambda t: t.cpu().numpy().transpose(0, 3, 1, 2)
    print('gain', gain, 'rand', rand)
    print('  initial   ', rel(nchw(sv['initial']), taps['initial']), 'norm', np.abs(taps['initial']).mean())
    print('  post_merge', rel(nchw(sv['scale_in']), taps['post_merge']), 'norm', np.abs(taps['post_merge']).mean())
    print('  output    ', rel(y.cpu().numpy(), ref), 'norm', np.abs(ref).mean())

"""dbcnn.json train steps only (batch 50, 288 x 288), for a rocprofv3 kernel ranking of that model.   python tools/probe_dbcnn_step.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import configs, ops
from poisson_cnn_amd.losses import loss_wrapper
from poisson_cnn_amd.models import Dirichlet_BC_NN_Legacy_2
from poisson_cnn_amd.train import Adam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ops.set_math_mode(os.environ.get('PCNN_MATH', 'fp32'))
g = torch.Generator().manual_seed(0)
H = W = 288
cfg = configs.dbcnn()
N = cfg['dataset']['batch_size']
model = Dirichlet_BC_NN_Legacy_2(**cfg['model'])
model.compile(loss=loss_wrapper(global_batch_size=N, **cfg['training']['loss_parameters']), optimizer=Adam(**cfg['training']['optimizer_parameters']))
bc = torch.cumsum(torch.randn(N, 1, W, generator=g) * 0.1, 2).cuda()
dx = (torch.rand(N, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
tgt = (torch.randn(N, 1, H, W, generator=g) * 0.1).cuda()
for _ in range(steps):
    out = model.train_step(((bc, dx), tgt))
torch.cuda.synchronize()
print('ok', float(out['loss']))

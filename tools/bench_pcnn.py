"""Times full training steps of the two 'next row' models on their shipped configurations (experiments/dbcnn.json: batch 50,
experiments/pcnn_end_to_end.json: batch 5; grids of 288 x 288, the middle of the configs' 192...384 range) with on-device data."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import configs, ops
from poisson_cnn_amd.losses import loss_wrapper
from poisson_cnn_amd.models import Dirichlet_BC_NN_Legacy_2, Homogeneous_Poisson_NN_Legacy, Poisson_CNN_Legacy
from poisson_cnn_amd.train import Adam
from poisson_cnn_amd.graphs import GraphedInference, GraphedTrainStep


def timeit(fn, steps=3, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ops.set_math_mode(os.environ.get('PCNN_MATH', 'fp32'))
    g = torch.Generator().manual_seed(0)
    H = W = 288
    # ---- dbcnn.json, batch 50
    cfg = configs.dbcnn()
    N = cfg['dataset']['batch_size']
    model = Dirichlet_BC_NN_Legacy_2(**cfg['model'])
    model.compile(loss=loss_wrapper(global_batch_size=N, **cfg['training']['loss_parameters']), optimizer=Adam(**cfg['training']['optimizer_parameters']))
    bc = torch.cumsum(torch.randn(N, 1, W, generator=g) * 0.1, 2).cuda()
    dx = (torch.rand(N, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
    tgt = (torch.randn(N, 1, H, W, generator=g) * 0.1).cuda()
    t = timeit(lambda: model.train_step(((bc, dx), tgt)))
    print('Dirichlet_BC_NN_Legacy_2 (dbcnn.json, %d params): train step, %d x %dx%d: %.1f ms -> %.1f grids/s' % (model.count_params(), N, H, W, 1e3 * t, N / t))
    t = timeit(lambda: model([bc, dx, H]))
    print('Dirichlet_BC_NN_Legacy_2 inference, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, W, 1e3 * t, N / t))
    step = GraphedTrainStep(model, ((bc, dx), tgt))
    t = timeit(lambda: step(((bc, dx), tgt)), steps=10)
    print('Dirichlet_BC_NN_Legacy_2 train step as ONE hipGraph replay + eager Adam, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, W, 1e3 * t, N / t))
    inf = GraphedInference(model, [bc, dx, H])
    t = timeit(lambda: inf([bc, dx, H]), steps=10)
    print('Dirichlet_BC_NN_Legacy_2 inference as ONE hipGraph replay, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, W, 1e3 * t, N / t))
    del model, step, inf
    # ---- pcnn_end_to_end.json, batch 5
    cfg = configs.pcnn_end_to_end()
    N = cfg['dataset']['batch_size']
    model = Poisson_CNN_Legacy(Homogeneous_Poisson_NN_Legacy(**cfg['hpnn_model']), Dirichlet_BC_NN_Legacy_2(**cfg['dbcnn_model']))
    model.compile(loss=loss_wrapper(global_batch_size=N, **cfg['training']['loss_parameters']), optimizer=Adam(**cfg['training']['optimizer_parameters']))
    rhs = torch.randn(N, 1, H, W, generator=g).cuda()
    edges = [torch.cumsum(torch.randn(N, 1, n, generator=g) * 0.1, 2).cuda() for n in (W, H, W, H)]
    dx = (torch.rand(N, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
    tgt = (torch.randn(N, 1, H, W, generator=g) * 0.1).cuda()
    inp = [rhs] + edges + [dx]
    t = timeit(lambda: model.train_step((inp, tgt)))
    print('Poisson_CNN_Legacy (pcnn_end_to_end.json, %d params): train step, %d x %dx%d: %.1f ms -> %.1f grids/s' % (model.count_params(), N, H, W, 1e3 * t, N / t))
    t = timeit(lambda: model(inp))
    print('Poisson_CNN_Legacy inference, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, W, 1e3 * t, N / t))
    step = GraphedTrainStep(model, (inp, tgt))
    t = timeit(lambda: step((inp, tgt)), steps=10)
    print('Poisson_CNN_Legacy train step as ONE hipGraph replay + eager Adam, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, W, 1e3 * t, N / t))
    inf = GraphedInference(model, inp)
    t = timeit(lambda: inf(inp), steps=10)
    print('Poisson_CNN_Legacy inference as ONE hipGraph replay, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, W, 1e3 * t, N / t))


if __name__ == '__main__':
    main()

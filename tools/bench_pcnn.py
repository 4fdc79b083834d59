"""Times full training steps of the 'next row' models on their shipped configurations (experiments/dbcnn.json: batch 50,
experiments/pcnn_end_to_end.json: batch 5; grids of 288 x 288, the middle of the configs' 192...384 range) with on-device data, and of the two
metalearning models on the only configurations the reference holds for them: Homogeneous_Poisson_NN_Metalearning with the
hyper-parameters of its own example (models/Homogeneous_Poisson_NN_Metalearning.py:334-377 = configs.hpnn_metalearning()) at hpnn.json's batch
size on 288 x 288 and 200 x 200 grids, Dirichlet_BC_NN_Metalearning at its __main__ block's batch 10 on 101 x 75 (:211-250).
PCNN_GROUPED_VALU=1 times the metalearning models on the round-3 vector-ALU kernels (the matrix-core grouped kernels are the default)."""
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


def metalearning():
    from poisson_cnn_amd.hpnn_models import Homogeneous_Poisson_NN_Metalearning
    from poisson_cnn_amd.dbcnn_models import Dirichlet_BC_NN_Metalearning
    route = 'vector-ALU grouped kernels (PCNN_GROUPED_VALU)' if os.environ.get('PCNN_GROUPED_VALU') else 'matrix-core grouped kernels'
    g = torch.Generator().manual_seed(1)
    cfg = configs.hpnn_metalearning()
    mc = {k: v for k, v in cfg['model'].items() if k != 'model_type'}
    for N, H in ((cfg['dataset']['batch_size'], 288), (10, 200)):
        model = Homogeneous_Poisson_NN_Metalearning(**mc)
        model.compile(loss=loss_wrapper(global_batch_size=N, **cfg['training']['loss_parameters']), optimizer=Adam(**cfg['training']['optimizer_parameters']))
        rhs = (torch.rand(N, 1, H, H, generator=g) * 2 - 1).cuda()
        dx = (torch.rand(N, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
        tgt = (torch.randn(N, 1, H, H, generator=g) * 0.1).cuda()
        t = timeit(lambda: model.train_step(((rhs, dx), tgt)))
        print('Homogeneous_Poisson_NN_Metalearning (%d params, %s): train step, %d x %dx%d: %.1f ms -> %.1f grids/s' % (model.count_params(), route, N, H, H, 1e3 * t, N / t))
        t = timeit(lambda: model([rhs, dx]))
        print('Homogeneous_Poisson_NN_Metalearning inference, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, H, H, 1e3 * t, N / t))
        del model
    N, nx, ny = 10, 101, 75
    model = Dirichlet_BC_NN_Metalearning(**configs.dbcnn_metalearning_main())
    lossp = dict(configs.dbcnn()['training']['loss_parameters'])
    model.compile(loss=loss_wrapper(global_batch_size=N, **lossp), optimizer=Adam(learning_rate=1e-4))
    bc = torch.cumsum(torch.randn(N, 1, ny, generator=g) * 0.1, 2).cuda()
    dx = (torch.rand(N, 1, generator=g) * 4.5e-2 + 5e-3).cuda()
    dx2 = dx.repeat(1, 2).contiguous()
    tgt = (torch.randn(N, 1, nx, ny, generator=g) * 0.1).cuda()
    model([bc, dx2, nx])
    t = timeit(lambda: model.train_step(((bc, dx), tgt)))
    print('Dirichlet_BC_NN_Metalearning (__main__ configuration, %d params, %s): train step, %d x %dx%d: %.1f ms -> %.1f grids/s' % (model.count_params(), route, N, nx, ny, 1e3 * t, N / t))
    t = timeit(lambda: model([bc, dx2, nx]))
    print('Dirichlet_BC_NN_Metalearning inference, %d x %dx%d: %.1f ms -> %.1f grids/s' % (N, nx, ny, 1e3 * t, N / t))


if __name__ == '__main__':
    if 'metalearning' in sys.argv[1:]:
        metalearning()
    else:
        main()
        metalearning()

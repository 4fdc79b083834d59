"""Throughput of the on-device reference-solution generators at 512^2 (BASELINE configs[4]) next to the CPU stand-in for
pyamg (scipy sparse direct solve of the same 5-point system); GPU box only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dataset as ods  # noqa: E402
from poisson_cnn_amd import configs  # noqa: E402
from poisson_cnn_amd.dataset import numerical_dataset_generator, reverse_poisson_dataset_generator  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


N, H = 32, 512
gen = numerical_dataset_generator(batch_size=N, batches_per_epoch=1, randomize_rhs_smoothness=True, rhs_random_smoothness_range=[3, 8], seed=0,
                                  output_shape=[H, H], return_rhs=True, return_boundaries=True, return_dx=True, boundary_smoothness=5)
t = timeit(lambda: gen[0], 5)
print('numerical (control points -> bicubic -> DST-I fp64 solve): %d x %d^2 in %.2f ms -> %.0f samples/s' % (N, H, t * 1e3, N / t))
dcfg = dict(configs.hpnn()['dataset']); dcfg.update(batch_size=N)
rg = reverse_poisson_dataset_generator(seed=0, **dcfg); rg.fixed_output_shape = (H, H)
t = timeit(lambda: rg[0], 5)
print('reverse (sine series + polynomial pairs): %d x %d^2 in %.2f ms -> %.0f samples/s' % (N, H, t * 1e3, N / t))
inp, soln = gen[0]
rhs, left, top, right, bottom, dx = [x.cpu().numpy().astype(np.float64) for x in inp]
t0 = time.perf_counter()
ref = ods.multigrid_poisson_solve(rhs[:2, 0], {'left': left[:2, 0], 'right': right[:2, 0], 'top': top[:2, 0], 'bottom': bottom[:2, 0]}, dx[:2, 0])
tc = (time.perf_counter() - t0) / 2
err = np.linalg.norm(soln[:2, 0].cpu().numpy() - ref) / np.linalg.norm(ref)
print('CPU stand-in (scipy splu of the same system, factorisation amortised over 2 samples): %.2f s/sample -> %.2f samples/s; GPU vs CPU rel-L2 %.2e' % (tc, 1 / tc, err))

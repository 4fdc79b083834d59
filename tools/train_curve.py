"""End-to-end sanity run at BASELINE configs[4] scale (GPU box only): the shipped hpnn.json model trained for a few dozen optimizer steps on
ground truth generated on the device - either the reference's own training data (`--data reverse`: the analytic sine/polynomial pairs of
experiments/hpnn.json at a fixed 512^2) or the finite-difference solver (`--data numerical`) - printing the loss per step and the sustained
rate INCLUDING data generation.   python tools/train_curve.py [--steps 60] [--batch 32] [--hw 512] [--lr 1e-4] [--data reverse|numerical]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import configs  # noqa: E402
from poisson_cnn_amd.dataset import numerical_dataset_generator, reverse_poisson_dataset_generator  # noqa: E402
from poisson_cnn_amd.losses import loss_wrapper  # noqa: E402
from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy  # noqa: E402
from poisson_cnn_amd.train import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--hw', type=int, default=512)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--data', default='reverse')
    a = ap.parse_args()
    cfg = configs.hpnn()
    model = Homogeneous_Poisson_NN_Legacy(**cfg['model'])
    model.compile(loss=loss_wrapper(global_batch_size=a.batch, **cfg['training']['loss_parameters']), optimizer=Adam(learning_rate=a.lr))
    if a.data == 'reverse':
        d = dict(cfg['dataset']); d.update(batch_size=a.batch)
        gen = reverse_poisson_dataset_generator(seed=0, **d)
        gen.fixed_output_shape = (a.hw, a.hw)
    else:
        gen = numerical_dataset_generator(batch_size=a.batch, batches_per_epoch=1, randomize_rhs_smoothness=True, rhs_random_smoothness_range=[3, 8], seed=0,
                                          output_shape=[a.hw, a.hw], return_rhs=True, return_boundaries=False, return_dx=True, nonzero_boundaries=[], normalize_by_domain_size=True)
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(a.steps):
        inp, soln = gen[step]
        logs = model.train_step((tuple(inp), soln))
        losses.append(float(logs['loss']))
        if step % 5 == 0 or step == a.steps - 1:
            print('step %3d  loss %.6f  mse %.4e' % (step, losses[-1], float(logs['mse'])), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    head, tail = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    print('%d steps of %d x %d^2 (%s data generated on device each step) in %.1f s: %.1f grids/s incl. data generation; mean loss first 5 steps %.5f -> last 5 steps %.5f'
          % (a.steps, a.batch, a.hw, a.data, dt, a.steps * a.batch / dt, head, tail))
    assert all(l == l for l in losses), 'NaN loss'


if __name__ == '__main__':
    main()

"""Generates tests/golden/atsize_golden.npz: fp64 ORACLE vectors at BASELINE.json's full grid sizes.

  C2: hpnn.json model, forward of ONE sample (index 5) of the 16 x 256^2 Dirichlet batch (seed 2).  Samples are independent
      (inference-mode BN), so sample k of the GPU's 16 x 256^2 batch must reproduce it.
  C3: hpnn.json model, full training-step quantities of ONE 512^2 sample (seed 3) for bc_type dirichlet and neumann:
      loss (global_batch_size = 32, so it is this sample's share of the 32 x 512^2 batch loss), prediction, the gradient of the
      first and of the last two convolution layers, and the L2 norm of every parameter's gradient.

  C4: hpnn.json model, forward of ONE sample (index 3) of the 8 x 1024^2 Dirichlet batch (seed 4) - the bench workload's size.

The oracle is oracle/hpnn.py on oracle/torch_twin.py (fp64 torch-CPU; its forward is pinned to the numpy oracle oracle/np_ops.py
in tests/test_oracle_ops.py - at these sizes the pure-numpy convolution would take hours).  The TensorFlow reference cannot run
in the build container.  Run time on 8 cores: about 15 minutes, ~30 GB of memory.

    python tests/golden/make_atsize_golden.py [c2] [c3] [c4]        (c4 alone: about 10 minutes)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hpnn as ohpnn, torch_twin, loss as oloss  # noqa: E402
from poisson_cnn_amd import configs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, 'atsize_golden.npz')
WEIGHT_SEED, WEIGHT_GAIN = 11, 1.6


def c2_inputs():
    """rhs = 2U-1 of shape [16,1,256,256] (seed 2) scaled to max-abs 1 per sample, dx ~ U(5e-3, 5e-2) per sample."""
    rng = np.random.default_rng(2)
    rhs = rng.uniform(-1, 1, (16, 1, 256, 256))
    rhs /= np.abs(rhs).max(axis=(1, 2, 3), keepdims=True)
    dx = rng.uniform(5e-3, 5e-2, (16, 1))
    return rhs.astype(np.float32), dx.astype(np.float32)


def c4_inputs():
    """SURVEY 8d C4 (per-GPU batch): rhs = 2U-1 of shape [8,1,1024,1024] (seed 4) scaled to max-abs 1 per sample, dx ~ U(5e-3, 5e-2)."""
    rng = np.random.default_rng(4)
    rhs = rng.uniform(-1, 1, (8, 1, 1024, 1024))
    rhs /= np.abs(rhs).max(axis=(1, 2, 3), keepdims=True)
    dx = rng.uniform(5e-3, 5e-2, (8, 1))
    return rhs.astype(np.float32), dx.astype(np.float32)


def c3_inputs(n=32):
    """[32,1,512,512] seed 3; the target is a smooth analytic-looking field (low-order sine series) so that the loss
    terms have the magnitudes they have in training."""
    rng = np.random.default_rng(3)
    rhs = rng.uniform(-1, 1, (n, 1, 512, 512))
    rhs /= np.abs(rhs).max(axis=(1, 2, 3), keepdims=True)
    dx = rng.uniform(5e-3, 5e-2, (n, 1))
    t = np.linspace(0, np.pi, 512)
    coef = rng.standard_normal((n, 4, 4)) * 0.1
    S = np.stack([np.sin((a + 1) * t) for a in range(4)])            # (4, 512)
    tgt = np.einsum('nab,ah,bw->nhw', coef, S, S)[:, None]
    return rhs.astype(np.float32), dx.astype(np.float32), tgt.astype(np.float32)


def main():
    which = set(sys.argv[1:]) or {'c2', 'c3', 'c4'}
    out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    full = configs.hpnn()
    if 'c2' in which:
        cfg = full['model']
        p = ohpnn.init_params(cfg, seed=WEIGHT_SEED, gain=WEIGHT_GAIN, randomize_all=True)
        rhs, dx = c2_inputs()
        k = 5
        t0 = time.time()
        with torch.no_grad():
            y = ohpnn.forward(torch_twin, cfg, {n: torch.tensor(v) for n, v in p.items()}, torch.tensor(rhs[k:k + 1].astype(np.float64)),
                              torch.tensor(dx[k:k + 1].astype(np.float64)))
        out['c2_sample'] = np.int64(k)
        out['c2_out'] = y.numpy().astype(np.float32)
        print('c2: %.1f s, max|y| %.4g' % (time.time() - t0, np.abs(out['c2_out']).max()), flush=True)
        np.savez_compressed(PATH, **out)
    if 'c4' in which:
        cfg = full['model']
        p = ohpnn.init_params(cfg, seed=WEIGHT_SEED, gain=WEIGHT_GAIN, randomize_all=True)
        rhs, dx = c4_inputs()
        k = 3
        t0 = time.time()
        with torch.no_grad():
            y = ohpnn.forward(torch_twin, cfg, {n: torch.tensor(v) for n, v in p.items()}, torch.tensor(rhs[k:k + 1].astype(np.float64)),
                              torch.tensor(dx[k:k + 1].astype(np.float64)))
        out['c4_sample'] = np.int64(k)
        out['c4_out'] = y.numpy().astype(np.float32)
        print('c4: %.1f s, max|y| %.4g' % (time.time() - t0, np.abs(out['c4_out']).max()), flush=True)
        np.savez_compressed(PATH, **out)
    if 'c3' in which:
        rhs, dx, tgt = c3_inputs()
        for bc in ('dirichlet', 'neumann'):
            cfg = configs.hpnn()['model']
            cfg['bc_type'] = bc
            p = ohpnn.init_params(cfg, seed=WEIGHT_SEED, gain=WEIGHT_GAIN, randomize_all=True)
            pt = {n: torch.tensor(v, dtype=torch.float64, requires_grad=not n.endswith(('moving_mean', 'moving_variance'))) for n, v in p.items()}
            t0 = time.time()
            r64, d64 = rhs[:1].astype(np.float64), dx[:1].astype(np.float64)
            pred = ohpnn.forward(torch_twin, cfg, pt, torch.tensor(r64), torch.tensor(d64))
            L = oloss.loss_wrapper(global_batch_size=32, **full['training']['loss_parameters'])
            loss = L(tgt[:1].astype(np.float64), pred, torch.tensor(r64), np.concatenate([d64, d64], 1))
            loss.backward()
            names = [n for n, v in pt.items() if v.requires_grad]
            out['c3_%s_loss' % bc] = np.float64(loss.detach())
            out['c3_%s_pred' % bc] = pred.detach().numpy().astype(np.float32)
            out['c3_%s_grad_names' % bc] = np.array(names)
            out['c3_%s_grad_norms' % bc] = np.array([float(pt[n].grad.norm()) for n in names])
            for n in ('pre/conv0/kernel', 'final/out0/kernel', 'final/out0/bias', 'final/out1/kernel', 'final/out1/bias', 'final/stage0/conv/bias'):
                out['c3_%s_grad:%s' % (bc, n.replace('/', '.'))] = pt[n].grad.numpy().astype(np.float64)
            print('c3 %s: %.1f s, loss %.6g' % (bc, time.time() - t0, float(loss.detach())), flush=True)
            np.savez_compressed(PATH, **out)
            del pt, pred, loss
    print({k: getattr(v, 'shape', v) for k, v in out.items()})


if __name__ == '__main__':
    main()

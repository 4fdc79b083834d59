"""Generates tests/golden/atsize_golden.npz: fp64 ORACLE vectors at BASELINE.json's full grid sizes.

  C1: hpnn.json model, forward of the single 128^2 Dirichlet grid of BASELINE configs[0] (seed 1, dx = 0.02).
  C2: hpnn.json model, forward of ONE sample (index 5) of the 16 x 256^2 Dirichlet batch (seed 2).  Samples are independent
      (inference-mode BN), so sample k of the GPU's 16 x 256^2 batch must reproduce it.
  C3: hpnn.json model, full training-step quantities of ONE 512^2 sample (seed 3) for bc_type dirichlet and neumann:
      loss (global_batch_size = 32, so it is this sample's share of the 32 x 512^2 batch loss), prediction, the gradient of the
      first and of the last two convolution layers, and the L2 norm of every parameter's gradient.

  C4: hpnn.json model, forward of ONE sample (index 3) of the 8 x 1024^2 Dirichlet batch (seed 4) - the bench workload's size.

  C3-tanh (round 5, `c3tanh`): the C3 training step of ONE 512^2 sample through the SAME graph with tanh activations (tanh_config): loss, prediction, all
      286 gradient norms and every 16th entry of every parameter's gradient - the smooth graph on which a flat-gradient bound of 3e-4 is meaningful.

The oracle is oracle/hpnn.py on oracle/torch_twin.py (fp64 torch-CPU; its forward is pinned to the numpy oracle oracle/np_ops.py
in tests/test_oracle_ops.py - at these sizes the pure-numpy convolution would take hours).  The TensorFlow reference cannot run
in the build container.  Run time on 8 cores: about 15 minutes, ~30 GB of memory.

  C4 backward (round 4, `c4grad`): full training-step quantities of that SAME 1024^2 sample (global_batch_size = 8: its share of the bench
      batch's loss) - loss, first / last-layer gradients, gradients of wide-filter layers that switch to 64-point tiles only at this size
      (15 / 13 / 11 taps), every parameter's gradient norm.  PyTorch's fp64 CPU convolution cannot do this (it unfolds the image: 60 GB and
      5.5 minutes of backward for ONE 15 x 15 x 32 layer), so the twin evaluates its large convolutions by overlap-save FFT
      (oracle/torch_twin.set_fft_conv; pinned to F.conv2d in tests/test_oracle_ops.py), and every convolution-like op runs under
      torch.utils.checkpoint (only its input stays alive; identical arithmetic).  About 15 minutes, ~25 GB.

    python tests/golden/make_atsize_golden.py [c1] [c2] [c3] [c3tanh] [c4] [c4grad]        (c4 alone: about 10 minutes)
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


def c1_inputs():
    """SURVEY 8d C1 (BASELINE configs[0] at its literal size): rhs = 2U-1 of shape [1,1,128,128] (seed 1) scaled to max-abs 1, dx = [[0.02]]."""
    rng = np.random.default_rng(1)
    rhs = rng.uniform(-1, 1, (1, 1, 128, 128))
    rhs /= np.abs(rhs).max(axis=(1, 2, 3), keepdims=True)
    return rhs.astype(np.float32), np.array([[0.02]], dtype=np.float32)


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


def tanh_config():
    """hpnn.json's model with every leaky-ReLU replaced by tanh: a smooth graph, in which fp32 and fp64 cannot pick different activation slopes - the
    flat gradient of a whole training step then has to agree with the oracle to fp32 rounding, so a systematic error of 1e-3 in ONE mid-stack layer's
    data gradient fails the comparison."""
    import json
    cfg = configs.hpnn()['model']
    return json.loads(json.dumps(cfg).replace('tf.nn.leaky_relu', 'tf.nn.tanh'))


C3T_STRIDE = 16          # the fixture keeps every 16th entry of every parameter's gradient (5.56 M floats would be 22 MB) plus all 286 gradient norms


def c4_target():
    """Target of the C4 training step: the smooth low-order sine series of c3_inputs at 1024^2."""
    rng = np.random.default_rng(44)
    t = np.linspace(0, np.pi, 1024)
    coef = rng.standard_normal((8, 4, 4)) * 0.1
    S = np.stack([np.sin((a + 1) * t) for a in range(4)])
    return np.einsum('nab,ah,bw->nhw', coef, S, S)[:, None].astype(np.float32)


class _CheckpointedTwin:
    """oracle.torch_twin with its convolution-like ops run under torch.utils.checkpoint (identical arithmetic; an op's internal intermediates
    - padded copy, pre-activation - are recomputed in the backward pass instead of being kept)."""
    _WRAP = ('padded_conv2d', 'same_conv2d', 'conv2d_transpose_same', 'resize2d', 'pool2d_same')

    def __getattr__(self, name):
        fn = getattr(torch_twin, name)
        if name not in self._WRAP:
            return fn
        from torch.utils.checkpoint import checkpoint

        def wrapped(x, *a, **k):
            r = checkpoint(lambda x_: fn(x_, *a, **k), x, use_reentrant=False)
            if os.environ.get('PCNN_GOLDEN_VERBOSE'):
                import psutil
                print('   %s %s -> %s, rss %.1f GB' % (name, tuple(x.shape), tuple(r.shape), psutil.Process().memory_info().rss / 1e9), flush=True)
            return r
        return wrapped


C4_GRADS = ('pre/conv0/kernel', 'pre/conv2/kernel', 'final/out0/kernel', 'final/out0/bias', 'final/out1/kernel', 'final/out1/bias', 'final/stage0/conv/bias',
            'final/stage0/conv/kernel', 'final/stage0/res/conv1/kernel', 'final/stage1/conv/kernel', 'final/stage2/conv/kernel', 'final/stage2/res/conv0/bias',
            'deconv_f2/res0/conv1/kernel', 'post_merge_conv/kernel')


def main():
    which = set(sys.argv[1:]) or {'c2', 'c3', 'c4'}
    out = dict(np.load(PATH)) if os.path.exists(PATH) else {}
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    full = configs.hpnn()
    if 'c1' in which:
        cfg = full['model']
        p = ohpnn.init_params(cfg, seed=WEIGHT_SEED, gain=WEIGHT_GAIN, randomize_all=True)
        rhs, dx = c1_inputs()
        t0 = time.time()
        with torch.no_grad():
            y = ohpnn.forward(torch_twin, cfg, {n: torch.tensor(v) for n, v in p.items()}, torch.tensor(rhs.astype(np.float64)), torch.tensor(dx.astype(np.float64)))
        out['c1_out'] = y.numpy().astype(np.float32)
        print('c1: %.1f s, max|y| %.4g' % (time.time() - t0, np.abs(out['c1_out']).max()), flush=True)
        np.savez_compressed(PATH, **out)
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
    if 'c4grad' in which:
        import threading
        import psutil
        proc = psutil.Process()

        def _guard():                         # the build container has 64 GB and no swap: report the resident set, stop before the kernel does
            while True:
                r = proc.memory_info().rss / 1e9
                print('   [rss %.1f GB]' % r, flush=True)
                if r > 45:
                    print('resident set above 45 GB - aborting', flush=True)
                    os._exit(3)
                time.sleep(15)
        threading.Thread(target=_guard, daemon=True).start()
        cfg = configs.hpnn()['model']
        p = ohpnn.init_params(cfg, seed=WEIGHT_SEED, gain=WEIGHT_GAIN, randomize_all=True)
        rhs, dx = c4_inputs()
        tgt = c4_target()
        k = 3
        pt = {n: torch.tensor(v, dtype=torch.float64, requires_grad=not n.endswith(('moving_mean', 'moving_variance'))) for n, v in p.items()}
        t0 = time.time()
        r64, d64 = rhs[k:k + 1].astype(np.float64), dx[k:k + 1].astype(np.float64)
        torch_twin.set_fft_conv(min_pixels=200 * 200, tile=256)
        try:
            pred = ohpnn.forward(_CheckpointedTwin(), cfg, pt, torch.tensor(r64), torch.tensor(d64))
            print('c4grad: forward %.1f s' % (time.time() - t0), flush=True)
            L = oloss.loss_wrapper(global_batch_size=8, **full['training']['loss_parameters'])
            loss = L(tgt[k:k + 1].astype(np.float64), pred, torch.tensor(r64), np.concatenate([d64, d64], 1))
            loss.backward()
        finally:
            torch_twin.set_fft_conv(None)
        names = [n for n, v in pt.items() if v.requires_grad]
        out['c4_loss'] = np.float64(loss.detach())
        # the FFT-evaluated forward against the F.conv2d-evaluated fixture of the same sample (c4_out): the two evaluations agree to fp32 storage
        out['c4_pred_vs_c4_out'] = np.float64(np.linalg.norm(pred.detach().numpy().astype(np.float32) - out['c4_out']) / np.linalg.norm(out['c4_out'])) if 'c4_out' in out else np.float64(-1)
        out['c4_grad_names'] = np.array(names)
        out['c4_grad_norms'] = np.array([float(pt[n].grad.norm()) for n in names])
        for n in C4_GRADS:
            g = pt[n].grad.numpy()
            # the wide-filter gradients are kept in float32: the rounding (6e-8) is far below the test tolerance and the fixture stays a few MB
            out['c4_grad:%s' % n.replace('/', '.')] = g.astype(np.float32 if g.size > 4096 else np.float64)
        print('c4grad: %.1f s, loss %.6g, pred vs c4_out %.3g' % (time.time() - t0, float(loss.detach()), float(out['c4_pred_vs_c4_out'])), flush=True)
        np.savez_compressed(PATH, **out)
        del pt, pred, loss
    if 'c3tanh' in which:
        rhs, dx, tgt = c3_inputs()
        cfg = tanh_config()
        p = ohpnn.init_params(cfg, seed=WEIGHT_SEED, gain=1.0, randomize_all=True)
        pt = {n: torch.tensor(v, dtype=torch.float64, requires_grad=not n.endswith(('moving_mean', 'moving_variance'))) for n, v in p.items()}
        t0 = time.time()
        r64, d64 = rhs[:1].astype(np.float64), dx[:1].astype(np.float64)
        pred = ohpnn.forward(torch_twin, cfg, pt, torch.tensor(r64), torch.tensor(d64))
        L = oloss.loss_wrapper(global_batch_size=32, **full['training']['loss_parameters'])
        loss = L(tgt[:1].astype(np.float64), pred, torch.tensor(r64), np.concatenate([d64, d64], 1))
        loss.backward()
        names = [n for n, v in pt.items() if v.requires_grad]
        out['c3tanh_loss'] = np.float64(loss.detach())
        out['c3tanh_pred'] = pred.detach().numpy().astype(np.float32)
        out['c3tanh_grad_names'] = np.array(names)
        out['c3tanh_grad_norms'] = np.array([float(pt[n].grad.norm()) for n in names])
        out['c3tanh_grad_sub'] = np.concatenate([pt[n].grad.numpy().reshape(-1)[::C3T_STRIDE] for n in names]).astype(np.float64)
        print('c3tanh: %.1f s, loss %.6g, %d sampled gradient entries' % (time.time() - t0, float(loss.detach()), out['c3tanh_grad_sub'].size), flush=True)
        np.savez_compressed(PATH, **out)
        del pt, pred, loss
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

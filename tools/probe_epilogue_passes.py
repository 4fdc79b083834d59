"""Which activation-backward passes (pcnn_conv2d_epilogue_bwd*) does a c4 train step still run as separate kernels?  Lists (pixels, channels, GB moved) per call; GPU box only."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import _lib, configs
from poisson_cnn_amd.losses import loss_wrapper
from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
from poisson_cnn_amd.train import Adam


def main():
    cfg = configs.hpnn()
    model = Homogeneous_Poisson_NN_Legacy(**cfg['model'])
    model.compile(loss=loss_wrapper(global_batch_size=8, **cfg['training']['loss_parameters']), optimizer=Adam(**cfg['training']['optimizer_parameters']))
    g = torch.Generator(device='cuda').manual_seed(0)
    rhs = torch.randn(8, 1, 1024, 1024, device='cuda', generator=g)
    dx = torch.full((8, 1), 0.01, device='cuda')
    tgt = torch.randn(8, 1, 1024, 1024, device='cuda', generator=g)
    model.train_step(((rhs, dx), tgt))
    calls = []
    orig = _lib.Handle.call

    def spy(self, name, *a):
        if 'epilogue_bwd' in name:
            calls.append((int(a[0].value), int(a[1].value)))
        return orig(self, name, *a)
    _lib.Handle.call = spy
    try:
        model.train_step(((rhs, dx), tgt))
    finally:
        _lib.Handle.call = orig
    torch.cuda.synchronize()
    agg = collections.Counter(calls)
    tot = 0.0
    for (pix, c), n in sorted(agg.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[1]):
        gb = 3 * 4.0 * pix * c * n / 1e9
        tot += gb
        print('%10d pixels x %2d channels  x%2d  %.2f GB' % (pix, c, n, gb))
    print('total %.1f GB in %d calls' % (tot, len(calls)))


if __name__ == '__main__':
    main()

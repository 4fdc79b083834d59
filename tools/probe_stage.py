"""Narrow resnet stage (3x3, C -> C three times) at 8 x 1024^2: ONE launch (pcnn_resnet3_fwd) against the three-launch chain, training and inference; GPU box only.
Prints ms per stage and the north star's figure: algorithmic bytes of the UNFUSED layers / time as a fraction of 8 TB/s."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import ops


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    N, HW = 8, 1024
    for C in (8, 4):
        g = torch.Generator(device='cuda').manual_seed(C)
        x = torch.randn(N, HW, HW, C, device='cuda', generator=g)
        ws = [torch.randn(3, 3, C, C, device='cuda', generator=g) * 0.1 for _ in range(3)]
        bs = [torch.randn(C, device='cuda', generator=g) * 0.1 for _ in range(3)]
        bufs = [torch.empty_like(x) for _ in range(4)]

        def chain(training):
            o0 = ops.conv2d_fwd(x, ws[0], bs[0], pad_top=1, pad_left=1, act='leaky_relu', out=bufs[0])
            o1 = ops.conv2d_fwd(o0, ws[1], bs[1], pad_top=1, pad_left=1, act='leaky_relu', residual=x, out=bufs[1], act_out=bufs[2] if training else None)
            return ops.conv2d_fwd(o1, ws[2], bs[2], pad_top=1, pad_left=1, act='leaky_relu', out=bufs[3])

        T = 4.0 * N * HW * HW * C
        for training in (True, False):
            y3 = chain(training).clone()
            yf = ops.resnet3_fwd(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], act='leaky_relu', training=training)[0]
            same = bool(torch.equal(y3, yf))
            t3 = timeit(lambda: chain(training))
            t1 = timeit(lambda: ops.resnet3_fwd(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], act='leaky_relu', training=training, out=bufs[3]))
            nb = (8.0 if training else 7.0) * T
            print('C=%2d %-9s three launches %.3f ms (%.3f of 8 TB/s) | one launch %.3f ms (%.3f; moves %d passes) | bit-identical %s'
                  % (C, 'training' if training else 'inference', t3 * 1e3, nb / t3 / 8e12, t1 * 1e3, nb / t1 / 8e12, 5 if training else 2, same), flush=True)


if __name__ == '__main__':
    main()

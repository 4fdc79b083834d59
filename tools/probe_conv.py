"""Quick timing probe of the conv kernels on the GPU box (not part of the test-suite)."""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from poisson_cnn_amd import ops


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    N, H, W = 2, 1024, 1024
    for (k, ci, co) in [(15, 32, 32), (13, 28, 28), (9, 24, 24), (7, 32, 32), (5, 16, 16), (3, 32, 32), (3, 8, 8), (3, 4, 1), (15, 3, 4)]:
        x = torch.randn(N, H, W, ci, device='cuda')
        w = torch.randn(k, k, ci, co, device='cuda') * 0.01
        b = torch.zeros(co, device='cuda')
        y = torch.empty(N, H, W, co, device='cuda')
        t = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=k // 2, pad_left=k // 2, act='leaky_relu', out=y))
        flops = 2.0 * N * H * W * k * k * ci * co
        byts = 4.0 * N * H * W * (ci + co)
        print('fwd k=%2d %2d->%2d: %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s' % (k, ci, co, t * 1e3, flops / t / 1e12, byts / t / 1e9), flush=True)


if __name__ == '__main__':
    main()

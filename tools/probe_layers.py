"""Per-layer-shape timing of the three MFMA conv kernels at the c4 workload size (8 x 1024^2); GPU box only."""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from poisson_cnn_amd import ops


def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    shapes = [(15, 3, 4, 1024, 1), (13, 4, 16, 1024, 1), (11, 16, 32, 1024, 1), (5, 32, 32, 1024, 1), (7, 64, 32, 1024, 1), (7, 32, 32, 1024, 3),
              (15, 32, 32, 1024, 4), (13, 32, 28, 1024, 1), (13, 28, 28, 1024, 3), (9, 28, 24, 1024, 1), (9, 24, 24, 1024, 3), (7, 24, 20, 1024, 1),
              (7, 20, 20, 1024, 3), (5, 20, 16, 1024, 1), (5, 16, 16, 1024, 3), (3, 16, 12, 1024, 1), (3, 12, 12, 1024, 3), (3, 12, 8, 1024, 1),
              (3, 8, 8, 1024, 3), (3, 8, 4, 1024, 1), (3, 4, 1, 1024, 1), (11, 32, 32, 512, 7), (9, 32, 32, 342, 7), (7, 32, 32, 256, 7), (7, 32, 32, 128, 7)]
    tot = [0.0, 0.0, 0.0, 0.0]
    print('%-22s %10s %8s | %10s %8s | %10s %8s' % ('layer', 'fwd ms', 'TF/s', 'dgrad ms', 'TF/s', 'wgrad ms', 'TF/s'))
    for (k, ci, co, hw, count) in shapes:
        x = torch.randn(N, hw, hw, ci, device='cuda')
        dz = torch.randn(N, hw, hw, co, device='cuda')
        w = torch.randn(k, k, ci, co, device='cuda') * 0.01
        wt = ops.flip_transpose_weights(w)
        b = torch.zeros(co, device='cuda')
        y = torch.empty(N, hw, hw, co, device='cuda')
        dx = torch.empty(N, hw, hw, ci, device='cuda')
        dw = torch.empty_like(w)
        p = k // 2
        tf = timeit(lambda: ops.conv2d_fwd(x, w, b, pad_top=p, pad_left=p, act='leaky_relu', out=y))
        td = timeit(lambda: ops.conv2d_fwd(dz, wt, None, pad_top=p, pad_left=p, out=dx))
        tw = timeit(lambda: ops.conv2d_wgrad(x, dz, w.shape, pad_top=p, pad_left=p, out=dw))
        fl = 2.0 * N * hw * hw * k * k * ci * co
        fused = ops.conv2d_bwd_fused(x, dz, w.shape, wt, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None)
        tb = timeit(lambda: ops.conv2d_bwd_fused(x, dz, w.shape, wt, pad_top=p, pad_left=p, pad_mode='CONSTANT', pad_value=0.0, dw=dw, residual=None)) if fused is not None else float('nan')
        print('k%2d %2d->%2d @%4d x%d %10.3f %8.1f | %10.3f %8.1f | %10.3f %8.1f | fused bwd %8.3f' % (k, ci, co, hw, count, tf * 1e3, fl / tf / 1e12, td * 1e3, fl / td / 1e12,
                                                                                  tw * 1e3, fl / tw / 1e12, tb * 1e3), flush=True)
        tot[0] += tf * count; tot[1] += td * count; tot[2] += tw * count
        tot[3] += (tb if fused is not None else td + tw) * count
        del x, dz, y, dx
    print('weighted totals per step: fwd %.1f ms, dgrad %.1f ms, wgrad %.1f ms; backward with the fused route where eligible %.1f ms' % (tot[0] * 1e3, tot[1] * 1e3, tot[2] * 1e3, tot[3] * 1e3))


if __name__ == '__main__':
    main()

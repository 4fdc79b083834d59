"""Forward-only (inference) throughput of the shipped hpnn.json model at the BASELINE.json grid sizes: 16 x 256^2 (configs[1]: "fwd-only conv stack"), 32 x 512^2 and
8 x 1024^2, both math modes; GPU box only.  The parity side of configs[1] is tests/test_gpu_atsize.py::test_c2_forward_16x256_matches_oracle."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import configs, ops
from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy


def main():
    # --ws-limit-mb M: cap the spectral workspace of every handle (pcnn_set_workspace_limit; the route then runs in chunks, bit-identical) - the
    # "workspace-limited host" scenario; PCNN_SPECTRAL=off in the environment: no spectral route at all (every wide layer on the direct kernels)
    import ctypes
    limit = int(sys.argv[sys.argv.index('--ws-limit-mb') + 1]) << 20 if '--ws-limit-mb' in sys.argv else 0
    sizes = ((8, 1024),) if '--only-1024' in sys.argv else ((16, 256), (32, 512), (8, 1024))
    model = Homogeneous_Poisson_NN_Legacy(**configs.hpnn()['model'])
    if limit:
        real_handle = ops.handle

        def limited():
            h = real_handle()
            if not getattr(h, '_limited', False):
                h.call('pcnn_set_workspace_limit', ctypes.c_size_t(limit))
                h._limited = True
            return h
        ops.handle = limited
        print('spectral workspace capped at %d MB per handle' % (limit >> 20), flush=True)
    g = torch.Generator().manual_seed(0)
    for mode in ('fp32', 'split_f16'):
        ops.set_math_mode(mode)
        for N, H in sizes:
            rhs = (torch.rand((N, 1, H, H), generator=g) * 2 - 1).cuda()
            dx = (torch.rand((N, 1), generator=g) * 4.5e-2 + 5e-3).cuda()
            for _ in range(3):
                y = model([rhs, dx])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                y = model([rhs, dx])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print('%-9s inference %2d x %4d^2: %7.2f ms per batch = %8.1f grids/s' % (mode, N, H, dt * 1e3, N / dt), flush=True)
    ops.set_math_mode('fp32')


if __name__ == '__main__':
    main()

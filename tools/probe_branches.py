"""Time of the convolution stages (down-sampled part) of each of the eight bottleneck branches of hpnn.json at 8 x 1024^2, forward and backward, run back to
back on one stream - what running the five coarse branches (f >= 8: images of 128^2 and below, launches that fill a fraction of the chip) beside the large ones
on other streams could hide at most."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poisson_cnn_amd import configs, ops
from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
model = Homogeneous_Poisson_NN_Legacy(**configs.hpnn()['model'])
model.ctx.use_side = False
N, H = 8, 1024
initial = torch.randn(N, H, H, 32, device='cuda')
blocks = model.bottleneck_deconv_blocks + model.bottleneck_multilinear_blocks
pyr = model._pool_pyramid(initial, blocks)
tot_f = tot_b = 0.0
for b in blocks:
    pooled = pyr[b.f][0] if b.f in pyr else None
    def fwd(): return b._down_and_convs(initial, True, pooled)
    o = fwd(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record(); o = fwd(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    tf = float(np.median(ts))
    d_in = torch.zeros_like(initial)
    tb = []
    for _ in range(5):
        o = fwd(); dco = torch.randn_like(o); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record(); b._backward_convs_and_down(dco, d_in); e.record(); torch.cuda.synchronize(); tb.append(s.elapsed_time(e))
    tbm = float(np.median(tb))
    print('%-18s f=%3d coarse %4d^2: conv stages forward %.3f ms, backward %.3f ms' % (b.name, b.f, o.shape[1], tf, tbm))
    if b.f >= 8: tot_f += tf; tot_b += tbm
print('coarse branches (f >= 8) together: forward %.2f ms + backward %.2f ms per step' % (tot_f, tot_b))

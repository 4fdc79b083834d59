"""Producer-write = consumer-read check of the FETCH_SIZE correction.  MI355X_MICROARCH.md calibrates "bytes read = 2 x FETCH_SIZE" for 16-byte-per-
lane streaming loads only; the 32-point spectral kernels load 4 bytes per lane.  The spectra give an independent check, because WRITE_SIZE needs no correction and every
spectrum written by one kernel is read exactly once by the next:
    forward transforms write X^  ->  mixing / weight-gradient kernels read it;   mixing writes Y^  ->  inverse transforms read it.
usage: python tools/fetch_check.py profiles/r04_c4_pmc_summary_fp32.json"""
import json
import sys

k = json.load(open(sys.argv[1]))['kernels']


def tot(pred, field):
    return sum(v[field] * v.get('launches_per_step', v['launches']) for n, v in k.items() if pred(n)) / 1e9


rd = lambda p: 2.0 * tot(p, 'fetch_size_bytes_per_launch_raw')
wr = lambda p: tot(p, 'write_size_bytes_per_launch')
fwd = lambda n: n.startswith('spec_fwd') or n.startswith('spec64_fwd')
inv = lambda n: n.startswith('spec_inv') or n.startswith('spec64_inv')
mix = lambda n: n.startswith('spec_mix')
wmix = lambda n: n.startswith('spec_wmix')
print('GB per training step (2 x FETCH_SIZE for reads, WRITE_SIZE for writes)')
print('forward transforms write %.1f | mixing kernels (spec_mix: forward pass, spec_mixw: both gradients of the backward pass in one read of the dz spectrum) read %.1f '
      '+ separate weight-gradient GEMM %.1f = %.1f (since spec_mixw every spectrum that is written is read exactly once: the two sides must agree)'
      % (wr(fwd), rd(mix), rd(wmix), rd(mix) + rd(wmix)))
print('mixing writes %.1f (+ weight-gradient GEMM %.1f) | inverse transforms read %.1f (their residual / activation inputs included)' % (wr(mix), wr(wmix), rd(inv)))
print('mixing: reads %.1f vs writes %.1f (equal channel counts in and out for most layers)' % (rd(mix), wr(mix)))

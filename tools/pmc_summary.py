"""Builds profiles/<name>.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --math <mode> --steps 1 --warmup 0
--no-cpu-baseline --no-dataset`: per-kernel average bytes per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section),
stamped with the hash of the kernel sources it was measured on (poisson_cnn_amd._lib.source_hash) - bench.py reports `traffic` only when
the stamp matches the tree it runs from.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <math>"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNELS = (r'(conv_fwd_split_kernel<\d, \d>|conv_fwd_kernel<\d>|wgrad_split_kernel|wgrad_kernel|spec_fwd_kernel<\w+, \w+>|spec_inv_kernel|spec_mix_kernel<\d>|spec_wmix_kernel|'
           r'conv_small_fwd_kernel|conv_small_wgrad_kernel|split_convert_kernel|split_absmax_kernel|epilogue_bwd\w*|deconv_fwd_mfma_kernel|resize_fwd_kernel)')
CONV = ('conv_fwd', 'wgrad', 'spec_', 'conv_small')      # what bench.py's `roofline` covers: every convolution launch


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        m = re.search(KERNELS, r['Kernel_Name'])
        if m:
            acc[m.group(1)].append(float(r['Counter_Value']) * 1024.0)
    return acc


def main():
    from poisson_cnn_amd import _lib
    fetch, write, out, math = sys.argv[1:5]
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fr = sum(f[k]) / max(len(f[k]), 1)
        wr = sum(w[k]) / max(len(w[k]), 1)
        kernels[k] = {'launches': len(f[k]), 'fetch_size_bytes_per_launch_raw': fr, 'write_size_bytes_per_launch': wr,
                      'traffic_bytes_per_launch': 2.0 * fr + wr}
    conv = [v for k, v in kernels.items() if k.startswith(CONV)]
    total = sum(v['traffic_bytes_per_launch'] * v['launches'] for v in conv)
    kernels['conv (all convolution kernels of one training step)'] = {'launches': 1, 'fetch_size_bytes_per_launch_raw': None, 'write_size_bytes_per_launch': None,
                                                                     'traffic_bytes_per_launch': total}
    json.dump({'command': 'rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --math %s --steps 1 --warmup 0 '
                          '--no-cpu-baseline --no-dataset (two separate passes; tools/collect_pmc.sh)' % math,
               'source_hash': _lib.source_hash(),
               'note': 'bytes = Counter_Value * 1024; gfx950 correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of the bytes of '
                       '16-B-per-lane reads, so traffic = 2*FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are included in FETCH_SIZE, so this is '
                       'an upper bound on HBM reads.  The "conv (...)" row is the sum over all convolution kernels of the step.',
               'kernels': kernels}, open(out, 'w'), indent=1)
    for k, v in kernels.items():
        print('%-60s launches %4d  traffic/launch %.1f MB' % (k, v['launches'], v['traffic_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()

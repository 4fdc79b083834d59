"""Builds profiles/<name>.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 0
--no-cpu-baseline`: per-kernel average bytes per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section).

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [math]"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name']
        m = re.search(r'(conv_fwd_split_kernel<\d, \d>|conv_fwd_kernel<\d>|wgrad_split_kernel|wgrad_kernel|split_convert_kernel|split_absmax_kernel|epilogue_bwd\w*)', n)
        if m:
            acc[m.group(1)].append(float(r['Counter_Value']) * 1024.0)
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    math = sys.argv[4] if len(sys.argv) > 4 else 'split_f16'
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fr = sum(f[k]) / max(len(f[k]), 1)
        wr = sum(w[k]) / max(len(w[k]), 1)
        kernels[k] = {'launches': len(f[k]), 'fetch_size_bytes_per_launch_raw': fr, 'write_size_bytes_per_launch': wr,
                      'traffic_bytes_per_launch': 2.0 * fr + wr}
    json.dump({'command': 'rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 '
                          '--no-cpu-baseline --math %s (two separate passes)' % math,
               'note': 'bytes = Counter_Value * 1024; gfx950 correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of the bytes of '
                       '16-B-per-lane reads, so traffic = 2*FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are included in FETCH_SIZE, so this is '
                       'an upper bound on HBM reads',
               'kernels': kernels}, open(out, 'w'), indent=1)
    for k, v in kernels.items():
        print('%-28s launches %4d  traffic/launch %.1f MB' % (k, v['launches'], v['traffic_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()

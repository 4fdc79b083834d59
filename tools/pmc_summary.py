"""Builds profiles/<name>.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --math <mode> --steps 1 --warmup 1
--no-cpu-baseline --no-dataset --no-inference --no-c3 --overlap-wgrad 0` (tools/collect_pmc.sh; the warm-up step's rows are dropped here): per-kernel average bytes per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section),
stamped with the hash of the kernel sources it was measured on (poisson_cnn_amd._lib.source_hash) - bench.py reports `traffic` only when
the stamp matches the tree it runs from.

A third pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, optional) adds the matrix-pipe busy fraction per kernel: BUSY sums the busy cycles
of the 1024 SIMDs, GUI_ACTIVE the active cycles of the 8 XCDs, so busy fraction = BUSY / (128 * GUI_ACTIVE).

A kernel-stats CSV of `rocprofv3 --kernel-trace --stats` over the same command with S steps - or, better, that run's kernel TRACE csv, whose first
step is then dropped like the counter passes' warm-up step - (optional, with S) adds each kernel's time per step,
so that the summary carries the per-kernel roofline table bench.py prints: launches and ms per step, bytes per launch, achieved TB/s against
the 8 TB/s HBM peak, matrix-pipe busy fraction.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <math> [<mfma_counter_collection.csv>
                                   [<kernel_stats.csv> <steps covered by it>]]"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNELS = (r'(conv_fwd_split_kernel<\d, \d>|conv_fwd_kernel<\d>|wgrad_split_kernel|wgrad_kernel|spec_fwd_kernel<\w+, \w+>|spec_inv_kernel|spec_mix_kernel<\d>|spec_mix_lds_kernel<\d>|spec_wmix_kernel|spec_mixw_kernel|'
           r'spec64_fwd4?_kernel<\w+>|spec64_inv_kernel|fft32_fwd_kernel<\w+>|fft32_inv_kernel|fft64_fwd_kernel<\w+(?:, \w+)*>|fft64_inv_kernel|fft32_fwd_multi_kernel|fft64_fwd_multi_kernel|'
           r'conv_small_fwd_kernel|conv_small_wgrad_kernel|resnet3_stage_kernel|split_convert_kernel|split_absmax_kernel|epilogue_bwd\w*|deconv_fwd_mfma_kernel|resize_fwd_kernel)')
OPTIMIZER = re.compile(r'(adam|sgd\w*|rmsprop)_kernel')      # the last kernel of a training step, whatever the optimizer
DROPPED = {}                                                # counter file -> was a warm-up step found and dropped?
CONV = ('conv_fwd', 'wgrad', 'spec', 'fft', 'conv_small', 'resnet3_stage')      # what bench.py's `roofline` covers: every convolution launch


def per_kernel(path, counter, scale=1024.0):
    """The counter passes run ONE warm-up step in front of the counted one (tools/collect_pmc.sh: --steps 1 --warmup 1): the first step of a process
    fills the filter-spectrum cache with one launch per filter, which no later step repeats.  Rows up to the first optimizer launch (adam_kernel, the
    last kernel of a step) are the warm-up and are dropped; a file without two optimizer launches is taken whole and the
    summary says so ('warmup_dropped': false - its per-launch averages then include the cache-filling step)."""
    acc = collections.defaultdict(list)
    rows = sorted((r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter), key=lambda r: int(r['Dispatch_Id']))
    marks = [int(r['Dispatch_Id']) for r in rows if OPTIMIZER.search(r['Kernel_Name'])]
    first = marks[0] if len(marks) >= 2 else -1
    DROPPED[path] = len(marks) >= 2
    for r in rows:
        if int(r['Dispatch_Id']) <= first:
            continue
        m = re.search(KERNELS, r['Kernel_Name'])
        if m:
            acc[m.group(1)].append(float(r['Counter_Value']) * scale)
    return acc


def main():
    from poisson_cnn_amd import _lib
    fetch, write, out, math = sys.argv[1:5]
    mfma = sys.argv[5] if len(sys.argv) > 5 else None
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    busy = per_kernel(mfma, 'SQ_VALU_MFMA_BUSY_CYCLES', 1.0) if mfma else {}
    active = per_kernel(mfma, 'GRBM_GUI_ACTIVE', 1.0) if mfma else {}
    kernels = {}
    for k in sorted(set(f) | set(w)):
        fr = sum(f[k]) / max(len(f[k]), 1)
        wr = sum(w[k]) / max(len(w[k]), 1)
        kernels[k] = {'launches': len(f[k]), 'fetch_size_bytes_per_launch_raw': fr, 'write_size_bytes_per_launch': wr,
                      'traffic_bytes_per_launch': 2.0 * fr + wr}
        if k in busy and sum(active.get(k, [])) > 0:
            kernels[k]['mfma_busy_frac'] = sum(busy[k]) / (128.0 * sum(active[k]))
    if len(sys.argv) > 7:            # per-kernel time from the --stats run: average duration per launch, launches per step
        steps = float(sys.argv[7])
        rows = list(csv.DictReader(open(sys.argv[6])))
        if rows and 'Start_Timestamp' in rows[0]:
            # the run's kernel TRACE (one row per launch): the first step of the process - up to its optimizer launch - is dropped like the warm-up step of
            # the counter passes (it fills the filter-spectrum cache with one small launch per filter, which would dilute the per-launch averages)
            rows.sort(key=lambda r: int(r['Dispatch_Id']))
            marks = [i for i, r in enumerate(rows) if OPTIMIZER.search(r['Kernel_Name'])]
            if len(marks) >= 2:
                rows, steps = rows[marks[0] + 1:], float(len(marks) - 1)
            rows = [{'Name': r['Kernel_Name'], 'Calls': 1, 'TotalDurationNs': int(r['End_Timestamp']) - int(r['Start_Timestamp'])} for r in rows]
        for r in rows:
            m = re.search(KERNELS, r['Name'])
            if m and m.group(1) in kernels:
                kv = kernels[m.group(1)]
                kv['stats_calls'] = kv.get('stats_calls', 0) + int(r['Calls'])
                kv['stats_total_ns'] = kv.get('stats_total_ns', 0.0) + float(r['TotalDurationNs'])
        for k, kv in kernels.items():
            if kv.get('stats_calls'):
                avg_s = kv['stats_total_ns'] / kv['stats_calls'] * 1e-9
                kv['avg_launch_ms'] = avg_s * 1e3
                kv['ms_per_step'] = kv['stats_total_ns'] * 1e-6 / steps
                kv['launches_per_step'] = kv['stats_calls'] / steps
                kv['hbm_tbs'] = kv['traffic_bytes_per_launch'] / avg_s / 1e12
                kv['hbm_frac'] = kv['hbm_tbs'] / 8.0
    conv = [v for k, v in kernels.items() if k.startswith(CONV)]
    total = sum(v['traffic_bytes_per_launch'] * v['launches'] for v in conv)
    kernels['conv (all convolution kernels of one training step)'] = {'launches': 1, 'fetch_size_bytes_per_launch_raw': None, 'write_size_bytes_per_launch': None,
                                                                     'traffic_bytes_per_launch': total}
    cb = sum(sum(busy[k]) for k in busy if k.startswith(CONV))
    ca = sum(sum(active[k]) for k in active if k.startswith(CONV))
    if ca > 0:
        kernels['conv (all convolution kernels of one training step)']['mfma_busy_frac'] = cb / (128.0 * ca)
    json.dump({'command': 'rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --math %s --steps 1 --warmup 1 '
                          '--no-cpu-baseline --no-dataset --no-inference --no-c3 --overlap-wgrad 0 (the warm-up step is dropped; separate passes; a third one counts SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; tools/collect_pmc.sh)' % math,
               'source_hash': _lib.source_hash(),
               'warmup_dropped': all(DROPPED.values()),
               'note': 'bytes = Counter_Value * 1024; gfx950 correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE reports 1/2 of the bytes of '
                       '16-B-per-lane reads, so traffic = 2*FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are included in FETCH_SIZE, so this is '
                       'an upper bound on HBM reads.  The "conv (...)" row is the sum over all convolution kernels of the step.',
               'kernels': kernels}, open(out, 'w'), indent=1)
    for k, v in kernels.items():
        print('%-60s launches %4d  traffic/launch %.1f MB  mfma busy %s' % (k, v['launches'], v['traffic_bytes_per_launch'] / 1e6,
                                                                               ('%.2f' % v['mfma_busy_frac']) if 'mfma_busy_frac' in v else '-'))


if __name__ == '__main__':
    main()

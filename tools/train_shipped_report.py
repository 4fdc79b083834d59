"""Folds the three outputs of tools/collect_train_shipped.sh into one text report (profiles/<round>_train_shipped.txt)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
G = os.path.join(ROOT, 'gpurun_out')
main = json.load(open(os.path.join(G, 'train_shipped_%s.json' % tag)))
print('# the shipped training workload (experiments/hpnn.json:62-75 through train/hpnn_legacy_train.py:26-60) on one MI355X')
print(main['workload'], '| workspace pre-sized at compile():', main['presized'])
for k in ('pass1_every_shape_new', 'pass2_same_shapes_again', 'first_visit_cost_ms', 'first_visit_generation_cost_ms', 'memory_after_pass1', 'filter_cache'):
    if k in main:
        print('%-32s %s' % (k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in main[k].items()})))
steps = 40.0
kt = os.path.join(G, 'train_shipped_%s_under_kernel_trace.json' % tag)
ks = os.path.join(G, 'train_shipped_%s_kernel_stats.csv' % tag)
if os.path.exists(ks) and os.path.exists(kt):
    rows = list(csv.DictReader(open(ks)))
    tot_ns = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    u = json.load(open(kt))['pass1_every_shape_new']
    wall = u['mean_step_ms']
    print('\n## kernel time vs wall (rocprofv3 --kernel-trace --stats, %d steps, includes the first steps)' % steps)
    print('kernel-time sum per step %.2f ms | launches per step %.0f | wall per step under the tracer (batch end to batch end, generation included) %.2f ms, of which host time in dataset[i] %.2f ms | wall / kernel sum = %.3f'
          % (tot_ns * 1e-6 / steps, calls / steps, wall, u['mean_gen_ms'], wall / (tot_ns * 1e-6 / steps)))
    for ps in ('pass1_every_shape_new', 'pass2_same_shapes_again'):
        if ps in main:
            print('untraced wall per step (%s) %.2f ms  ->  untraced wall / kernel sum = %.3f' % (ps, main[ps]['mean_step_ms'], main[ps]['mean_step_ms'] / (tot_ns * 1e-6 / steps)))
    print('%-70s %8s %10s %8s' % ('kernel', 'calls/st', 'ms/step', '%'))
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
        print('%-70s %8.1f %10.3f %8.2f' % (r['Name'][:70], int(r['Calls']) / steps, float(r['TotalDurationNs']) * 1e-6 / steps, 100 * float(r['TotalDurationNs']) / tot_ns))
    small = [r for r in rows if float(r['TotalDurationNs']) / max(int(r['Calls']), 1) < 30e3]
    print('launches under 30 us: %.0f per step, %.2f ms per step' % (sum(int(r['Calls']) for r in small) / steps, sum(float(r['TotalDurationNs']) for r in small) * 1e-6 / steps))
hs = os.path.join(G, 'train_shipped_%s_hip_api_stats.csv' % tag)
if os.path.exists(hs):
    print('\n## HIP API calls per step (rocprofv3 --hip-trace --stats, %d steps incl. start-up)' % steps)
    rows = list(csv.DictReader(open(hs)))
    want = ('Synchronize', 'Malloc', 'Free', 'Memcpy', 'Memset', 'LaunchKernel', 'EventRecord', 'StreamWaitEvent')
    for r in sorted(rows, key=lambda r: -int(r['Calls'])):
        if any(w in r['Name'] for w in want):
            print('%-40s %9.1f per step   %10.3f ms per step' % (r['Name'], int(r['Calls']) / steps, float(r['TotalDurationNs']) * 1e-6 / steps))

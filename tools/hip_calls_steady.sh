#!/bin/bash
# Steady-state HIP API calls per training step of the shipped workload: the difference of two rocprofv3 --hip-trace --stats runs (A and B steps) removes start-up
# (model build, presize, the one-time filter-spectrum fills):  bash tools/hip_calls_steady.sh [A=30] [B=60]  -> gpurun_out/train_shipped_hip_steady.txt
set -e -o pipefail
ROOT=$(pwd); export TMPDIR=/tmp
A=${1:-30}; B=${2:-60}
for n in $A $B; do
  d=$ROOT/gpurun_out/hipst_$n; rm -rf "$d"
  (cd /tmp && rocprofv3 --hip-trace --stats -d "$d" --output-format csv -- python3 "$ROOT/tools/bench_train_shipped.py" --steps $n --second-pass 0 > "$d.log" 2>&1)
  cp "$d"/*/*_hip_api_stats.csv "$ROOT/gpurun_out/hipst_$n.csv"; rm -rf "$d"
done
python3 - $A $B <<'PY' | tee "$ROOT/gpurun_out/train_shipped_hip_steady.txt"
import csv, sys
A, B = int(sys.argv[1]), int(sys.argv[2])
def load(n):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open('gpurun_out/hipst_%d.csv' % n))}
a, b = load(A), load(B)
print('# steady-state HIP API calls per step of the shipped training workload: (run of %d steps - run of %d steps) / %d; start-up (model build, presize, one-time filter-spectrum fills) cancels' % (B, A, B - A))
for k in sorted(b, key=lambda k: -(b[k][0] - a.get(k, (0, 0))[0])):
    dc = (b[k][0] - a.get(k, (0, 0))[0]) / (B - A)
    dt = (b[k][1] - a.get(k, (0, 0))[1]) / (B - A) * 1e-6
    if any(w in k for w in ('Synchronize', 'Malloc', 'Free', 'Memcpy', 'Memset', 'LaunchKernel', 'EventRecord', 'StreamWaitEvent', 'EventQuery')):
        print('%-36s %9.2f per step %10.3f ms per step' % (k, dc, dt))
PY

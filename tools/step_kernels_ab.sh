#!/bin/bash
# Per-kernel ms/step of the bench's train step for several builds on ONE box (box-to-box variance is +-1..3 %):  bash tools/step_kernels_ab.sh lib1.so lib2.so ...   ("" = in-tree)
ROOT=$(pwd); export TMPDIR=/tmp
i=0
for lib in "$@"; do
  i=$((i+1)); d=$ROOT/gpurun_out/ab_$i; rm -rf "$d"
  (cd /tmp && PCNN_LIBRARY=${lib:+$ROOT/$lib} rocprofv3 --kernel-trace --stats -d "$d" --output-format csv -- python3 "$ROOT/bench.py" --workload c4 --math fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-dataset --no-inference --no-c3 --overlap-wgrad 0 > "$d.log" 2>&1)
  cp "$d"/*/*_kernel_stats.csv "$ROOT/gpurun_out/ab_${i}_kernel_stats.csv"; rm -rf "$d"
  echo "== $i: ${lib:-in-tree}"
done
python3 - "$@" <<'PY'
import csv, sys
libs = sys.argv[1:]
tabs = []
for i in range(len(libs)):
    d = {}
    for r in csv.DictReader(open('gpurun_out/ab_%d_kernel_stats.csv' % (i + 1))):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('pcnn_spec::', '').replace('void ', '')[:46]
        d[n] = d.get(n, 0) + float(r['TotalDurationNs']) / 1e6 / 4
    tabs.append(d)
keys = sorted(set().union(*tabs), key=lambda k: -max(t.get(k, 0) for t in tabs))
print('%-48s' % 'kernel (ms/step)' + ''.join('%10s' % (l.split('_')[-1][:9] if l else 'in-tree') for l in libs))
for k in keys[:22]:
    print('%-48s' % k + ''.join('%10.2f' % t.get(k, 0) for t in tabs))
print('%-48s' % 'total' + ''.join('%10.2f' % sum(t.values()) for t in tabs))
PY

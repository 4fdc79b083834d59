#!/bin/bash
# The shipped training workload on the GPU box:  gpurun -- 'bash tools/collect_train_shipped.sh [tag]'
#   1. tools/bench_train_shipped.py, 200 steps x 2 passes            -> gpurun_out/train_shipped_<tag>.json
#   2. the same loop, 40 steps, under rocprofv3 --kernel-trace --stats -> kernel-time sum per step, launches per step
#   3. the same loop, 40 steps, under rocprofv3 --hip-trace --stats    -> hipStreamSynchronize / hipMemcpy / hipMalloc calls per step
# and tools/train_shipped_report.py folds the three into gpurun_out/train_shipped_<tag>.txt (copy to profiles/).
set -e -o pipefail
ROOT=$(pwd)
TAG=${1:-r06}
export TMPDIR=/tmp
python3 tools/bench_train_shipped.py --steps ${STEPS:-200} --out "$ROOT/gpurun_out/train_shipped_${TAG}.json" > "$ROOT/gpurun_out/train_shipped_${TAG}.log" 2>&1
echo "pass 1/2 done"
k=$ROOT/gpurun_out/train_shipped_${TAG}_kernels
rm -rf "$k"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$k" --output-format csv -- python3 "$ROOT/tools/bench_train_shipped.py" --steps 40 --second-pass 0 --out "$ROOT/gpurun_out/train_shipped_${TAG}_under_kernel_trace.json" > "$k.log" 2>&1)
cp "$k"/*/*_kernel_stats.csv "$ROOT/gpurun_out/train_shipped_${TAG}_kernel_stats.csv"
echo "kernel trace done"
a=$ROOT/gpurun_out/train_shipped_${TAG}_hip
rm -rf "$a"
(cd /tmp && rocprofv3 --hip-trace --stats -d "$a" --output-format csv -- python3 "$ROOT/tools/bench_train_shipped.py" --steps 40 --second-pass 0 --out "$ROOT/gpurun_out/train_shipped_${TAG}_under_hip_trace.json" > "$a.log" 2>&1)
cp "$a"/*/*_hip_api_stats.csv "$ROOT/gpurun_out/train_shipped_${TAG}_hip_api_stats.csv" 2>/dev/null || cp "$a"/*/*hip*stats*.csv "$ROOT/gpurun_out/train_shipped_${TAG}_hip_api_stats.csv"
rm -rf "$k" "$a"          # the raw traces are large; the stats tables are what the report reads
echo "hip trace done"
python3 tools/train_shipped_report.py "$TAG" > "$ROOT/gpurun_out/train_shipped_${TAG}.txt"
cat "$ROOT/gpurun_out/train_shipped_${TAG}.txt"

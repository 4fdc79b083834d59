#!/bin/bash
# HBM-traffic counters of one training step, per workload and math mode (run on the GPU box from the repo root through gpurun):
#   gpurun -- 'bash tools/collect_pmc.sh [round [workloads [modes]]]'      e.g.  bash tools/collect_pmc.sh r03 "c4 c3" "fp32"
# Three separate --pmc passes each, of one warm-up step (it fills the filter-spectrum cache; tools/pmc_summary.py drops its rows) + one counted step (FETCH_SIZE and WRITE_SIZE do not fit one pass; the third counts matrix-pipe busy cycles and active cycles;
# never combined with --stats / sys-trace), the program directly after `--`; plus one `--kernel-trace --stats` run of 1 + 3 steps for the
# per-kernel time of the roofline table.  Output: gpurun_out/pmc_<round>_<workload>_<mode>_*/ and gpurun_out/<round>_<workload>_pmc_summary_<mode>.json
# (+ <round>_<workload>_kernel_stats_<mode>.csv): copy them to profiles/.
set -e -o pipefail
ROOT=$(pwd)
ROUND=${1:-r06}
WORKLOADS=${2:-c4}
MODES=${3:-fp32}
export TMPDIR=/tmp
for wl in $WORKLOADS; do
for mode in $MODES; do
  st=$ROOT/gpurun_out/stats_${ROUND}_${wl}_${mode}
  rm -rf "$st"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$st" --output-format csv -- python3 "$ROOT/bench.py" --workload $wl --math $mode --steps 3 --warmup 1 \
      --no-cpu-baseline --no-dataset --no-inference --no-c3 --overlap-wgrad 0 > "$ROOT/gpurun_out/stats_${ROUND}_${wl}_${mode}.log" 2>&1)
  cp "$st"/*/*_kernel_stats.csv "$ROOT/gpurun_out/${ROUND}_${wl}_kernel_stats_${mode}.csv"
  cp "$st"/*/*_kernel_trace.csv "$ROOT/gpurun_out/${ROUND}_${wl}_kernel_trace_${mode}.csv"
  echo "stats $wl $mode done"
  for ctr in FETCH_SIZE WRITE_SIZE MFMA; do
    out=$ROOT/gpurun_out/pmc_${ROUND}_${wl}_${mode}_${ctr}
    rm -rf "$out"
    pmc=$ctr; if [ $ctr = MFMA ]; then pmc="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; fi
    (cd /tmp && rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out" -- python3 "$ROOT/bench.py" --workload $wl --math $mode --steps 1 --warmup 1 \
        --no-cpu-baseline --no-dataset --no-inference --no-c3 --overlap-wgrad 0 > "$ROOT/gpurun_out/pmc_${ROUND}_${wl}_${mode}_${ctr}.log" 2>&1)
    echo "pass $wl $mode $ctr done"
  done
  f=$(ls $ROOT/gpurun_out/pmc_${ROUND}_${wl}_${mode}_FETCH_SIZE/*/*counter_collection.csv | head -1)
  w=$(ls $ROOT/gpurun_out/pmc_${ROUND}_${wl}_${mode}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  m=$(ls $ROOT/gpurun_out/pmc_${ROUND}_${wl}_${mode}_MFMA/*/*counter_collection.csv | head -1)
  python3 tools/pmc_summary.py "$f" "$w" "$ROOT/gpurun_out/${ROUND}_${wl}_pmc_summary_${mode}.json" $mode "$m" "$ROOT/gpurun_out/${ROUND}_${wl}_kernel_trace_${mode}.csv" 4 | tail -30
done
done

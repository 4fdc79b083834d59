#!/bin/bash
# HBM-traffic counters of one training step, per math mode (run on the GPU box from the repo root through gpurun):
#   gpurun -- 'bash tools/collect_pmc.sh'
# Three separate --pmc passes per mode (FETCH_SIZE and WRITE_SIZE do not fit one pass; the third counts matrix-pipe busy cycles and active
# cycles; never combined with --stats / sys-trace), the program
# directly after `--`.  Output: gpurun_out/pmc_r02_<mode>_{fetch,write}/ and profiles/r02_c4_pmc_summary_<mode>.json (copy it back).
set -e -o pipefail
ROOT=$(pwd)
export TMPDIR=/tmp
for mode in fp32 split_f16; do
  for ctr in FETCH_SIZE WRITE_SIZE MFMA; do
    out=$ROOT/gpurun_out/pmc_r02_${mode}_${ctr}
    rm -rf "$out"
    pmc=$ctr; if [ $ctr = MFMA ]; then pmc="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; fi
    (cd /tmp && rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d "$out" -- python3 "$ROOT/bench.py" --math $mode --steps 1 --warmup 0 --no-cpu-baseline --no-dataset \
        > "$ROOT/gpurun_out/pmc_r02_${mode}_${ctr}.log" 2>&1)
    echo "pass $mode $ctr done"
  done
  f=$(ls $ROOT/gpurun_out/pmc_r02_${mode}_FETCH_SIZE/*/*counter_collection.csv | head -1)
  w=$(ls $ROOT/gpurun_out/pmc_r02_${mode}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  m=$(ls $ROOT/gpurun_out/pmc_r02_${mode}_MFMA/*/*counter_collection.csv | head -1)
  python3 tools/pmc_summary.py "$f" "$w" "$ROOT/gpurun_out/r02_c4_pmc_summary_${mode}.json" $mode "$m" | tail -25
done

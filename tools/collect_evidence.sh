#!/bin/bash
# Everything under profiles/r02_* in one go (run on the GPU box from the repo root through gpurun, ~6 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh'
# then copy gpurun_out/evidence/* into profiles/ (gpurun merges gpurun_out/ back).  Order matters: the PMC summaries are written first and
# copied into profiles/ on the box, so that bench.py finds a summary whose source stamp matches the tree it runs from.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/evidence
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1; echo "tests rc=$?"; tail -1 "$OUT/gpu_tests.log"
for m in fp32 split_f16; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_ev_$m" --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --math $m \
      --no-cpu-baseline --no-dataset > "$ROOT/gpurun_out/prof_ev_$m.log" 2>&1)
  echo "stats $m rc=$?"
done
cp "$ROOT"/gpurun_out/prof_ev_fp32/*/*_kernel_stats.csv "$OUT/r02_c4_kernel_stats_fp32.csv"
cp "$ROOT"/gpurun_out/prof_ev_split_f16/*/*_kernel_stats.csv "$OUT/r02_c4_kernel_stats_split.csv"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_ev_probe" --output-format csv -- python3 "$ROOT/tools/probe_spectral.py" 15 32 32 \
    > "$ROOT/gpurun_out/prof_ev_probe.log" 2>&1)
cp "$ROOT"/gpurun_out/prof_ev_probe/*/*_kernel_stats.csv "$OUT/r02_spectral_kernel_stats.csv"
bash tools/collect_pmc.sh > "$ROOT/gpurun_out/collect_pmc_ev.log" 2>&1; echo "pmc rc=$?"
cp "$ROOT"/gpurun_out/r02_c4_pmc_summary_fp32.json "$ROOT"/gpurun_out/r02_c4_pmc_summary_split_f16.json "$OUT/"
cp "$OUT"/r02_c4_pmc_summary_*.json "$ROOT/profiles/"
python bench.py > "$OUT/r02_bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
python bench.py --workload c3 --no-cpu-baseline --no-dataset > "$OUT/r02_bench_c3.json" 2> "$OUT/bench_c3.err"; echo "bench c3 rc=$?"
python tools/probe_spectral.py 2>&1 | grep -v amdgpu > "$OUT/r02_probe_spectral.txt"
python tools/probe_layers.py 2>&1 | grep -v amdgpu > "$OUT/r02_probe_layers.txt"
python tools/bench_dataset.py 2>&1 | grep -v amdgpu > "$OUT/r02_dataset_throughput.txt"
echo "evidence written to $OUT"

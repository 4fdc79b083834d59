#!/bin/bash
# Everything under profiles/r04_* in one go (run on the GPU box from the repo root through gpurun, ~10 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh'
# then copy gpurun_out/evidence/* into profiles/ (gpurun merges gpurun_out/ back).  Order matters: the PMC summaries are written first and
# copied into profiles/ on the box, so that bench.py finds a summary whose source stamp matches the tree it runs from.
set -o pipefail
ROOT=$(pwd)
R=r04
OUT=$ROOT/gpurun_out/evidence
mkdir -p "$OUT"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -v > "$OUT/gpu_tests.log" 2>&1; echo "tests rc=$?"; tail -1 "$OUT/gpu_tests.log"
echo "source hash $(python -c 'from poisson_cnn_amd import _lib; print(_lib.source_hash())')"
bash tools/collect_pmc.sh $R "c4" "fp32 split_f16" > "$ROOT/gpurun_out/collect_pmc_ev.log" 2>&1; echo "pmc c4 rc=$?"
bash tools/collect_pmc.sh $R "c3" "fp32" >> "$ROOT/gpurun_out/collect_pmc_ev.log" 2>&1; echo "pmc c3 rc=$?"
cp "$ROOT"/gpurun_out/${R}_c?_pmc_summary_*.json "$ROOT"/gpurun_out/${R}_c?_kernel_stats_*.csv "$OUT/"
cp "$OUT"/${R}_c?_pmc_summary_*.json "$ROOT/profiles/"
python bench.py > "$OUT/${R}_bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"
python tools/probe_layers.py 2>&1 | grep -v amdgpu > "$OUT/${R}_probe_layers.txt"
python tools/probe_tile64.py 2>&1 | grep -v amdgpu > "$OUT/${R}_probe_tile64.txt"
python tools/bench_dataset.py 2>&1 | grep -v amdgpu > "$OUT/${R}_dataset_throughput.txt"
python tools/bench_pcnn.py 2>&1 | grep -v amdgpu > "$OUT/${R}_next_models_throughput.txt"
python tools/train_curve.py --steps 60 2>&1 | grep -v amdgpu > "$OUT/${R}_train_curve_reverse.txt"
python tools/train_curve.py --steps 80 --data numerical 2>&1 | grep -v amdgpu > "$OUT/${R}_train_curve_numerical.txt"
echo "evidence written to $OUT"

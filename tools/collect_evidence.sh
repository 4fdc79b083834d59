#!/bin/bash
# The recurring evidence under profiles/<round>_* (round 6: R=r06; the A/B tables of that round come from tools/step_kernels_ab.sh and environment switches, see
# profiles/README.md) (run on the GPU box from the repo root through gpurun; two calls, because the PMC summaries of stage 1 must sit
# in profiles/ - copied there by hand between the calls - before stage 2's bench.py looks for a summary whose source stamp matches its tree):
#   gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh 1'     GPU test suite, rocprofv3 kernel stats + three --pmc passes per workload / mode
#   cp gpurun_out/evidence/r06_c?_pmc_summary_*.json gpurun_out/evidence/r06_c?_kernel_stats_*.csv profiles/
#   gpurun --timeout 1200 -- 'bash tools/collect_evidence.sh 2'     bench.py (the driver's command), layer / model / dataset probes, training curves
# then copy gpurun_out/evidence/* into profiles/ (gpurun merges gpurun_out/ back).
set -o pipefail
ROOT=$(pwd)
R=${ROUND:-r06}
STAGE=${1:-1}
OUT=$ROOT/gpurun_out/evidence
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "source hash $(python -c 'from poisson_cnn_amd import _lib; print(_lib.source_hash())')"
if [ "$STAGE" = 1 ]; then
  python -m pytest tests -m gpu -x -v > "$OUT/${R}_gpu_tests.log" 2>&1; echo "tests rc=$?"; tail -1 "$OUT/${R}_gpu_tests.log"
  bash tools/collect_pmc.sh $R "c4" "fp32" > "$ROOT/gpurun_out/collect_pmc_ev.log" 2>&1; echo "pmc c4 rc=$?"
  bash tools/collect_pmc.sh $R "c3" "fp32" >> "$ROOT/gpurun_out/collect_pmc_ev.log" 2>&1; echo "pmc c3 rc=$?"
  cp "$ROOT"/gpurun_out/${R}_c?_pmc_summary_*.json "$ROOT"/gpurun_out/${R}_c?_kernel_stats_*.csv "$OUT/"
else
  python bench.py > "$OUT/${R}_bench_c4.json" 2> "$OUT/bench_c4.err"; echo "bench c4 rc=$?"; cp bench_detail.json "$OUT/${R}_bench_c4_detail.json"
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/${R}_smoke.log" 2>&1; echo "smoke rc=$?"
  python bench.py --no-c3 --no-cpu-baseline --no-dataset --detail "$OUT/${R}_bench_c4_only_detail.json" > "$OUT/${R}_bench_c4_only.json" 2>/dev/null   # filter_cache block: the c4 model alone
  python tools/bench_inference.py 2>&1 | grep -v amdgpu > "$OUT/${R}_inference_throughput.txt"
  python tools/probe_xform.py 8 2>&1 | grep -v amdgpu > "$OUT/${R}_probe_xform.txt"
  python tools/probe_narrow_mfma.py 2>&1 | grep -v amdgpu > "$OUT/${R}_probe_narrow_mfma.txt"
  python tools/probe_layers.py 2>&1 | grep -v amdgpu > "$OUT/${R}_probe_layers.txt"
  python tools/bench_dataset.py 2>&1 | grep -v amdgpu > "$OUT/${R}_dataset_throughput.txt"
  python tools/bench_pcnn.py 2>&1 | grep -v amdgpu > "$OUT/${R}_next_models_throughput.txt"
  python tools/train_curve.py --steps 60 2>&1 | grep -v amdgpu > "$OUT/${R}_train_curve_reverse.txt"
  python tools/train_curve.py --steps 80 --data numerical 2>&1 | grep -v amdgpu > "$OUT/${R}_train_curve_numerical.txt"
  bash tools/collect_train_shipped.sh $R > "$OUT/collect_train_shipped.log" 2>&1; cp "$ROOT/gpurun_out/train_shipped_${R}.txt" "$OUT/${R}_train_shipped.txt"     # the reference's real training workload
  bash tools/hip_calls_steady.sh 30 60 > /dev/null 2>&1; cp "$ROOT/gpurun_out/train_shipped_hip_steady.txt" "$OUT/${R}_train_shipped_hip_steady.txt"
  python tools/probe_tile_shipped.py 50 2>&1 | grep -v amdgpu > "$OUT/${R}_probe_tile_shipped.txt"
fi
echo "evidence stage $STAGE written to $OUT"

#!/bin/bash
# Per-kernel averages of one layer's forward convolution at 8 x 1024^2 (tools/study_fft.py under rocprofv3 --kernel-trace --stats), for each library given:
#   bash tools/layer_kernels.sh "<taps> <tile>" lib1.so [lib2.so ...]      ("" = the in-tree library)
ROOT=$(pwd)
export TMPDIR=/tmp
read K T <<< "$1"; shift
for lib in "$@"; do
  d=$ROOT/gpurun_out/lk_$(basename "${lib:-default}" .so)_${K}_${T}
  rm -rf "$d"
  echo "== ${lib:-in-tree library}  k $K tile $T"
  (cd /tmp && PCNN_LIBRARY=${lib:+$ROOT/$lib} rocprofv3 --kernel-trace --stats -d "$d" --output-format csv -- python3 "$ROOT/tools/study_fft.py" $K $T 2>&1 | grep "study bits")
  python3 tools/kstats.py "$d" | grep -E "fft|spec_mix" | grep -v multi
  rm -rf "$d"
done

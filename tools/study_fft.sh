#!/bin/bash
# Removal study of the FFT transform kernels on the GPU box: per-kernel average duration (rocprofv3 --kernel-trace --stats) for each PCNN_FFT_STUDY setting.
#   bash tools/study_fft.sh <taps> <tile>            (BITS="0 32 64 96" TAG=mix: the mixing kernel without its MFMAs / with adjacent real and imaginary rows)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/study_${TAG:-fft}_${1:-15}_${2:-64}.txt
: > $OUT
for bits in ${BITS:-0 1 2 3 4 7}; do
  d=$ROOT/gpurun_out/study_$bits
  rm -rf $d
  (cd /tmp && PCNN_LIBRARY=$ROOT/build/study/libpcnn_study.so PCNN_FFT_STUDY=$bits rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python3 $ROOT/tools/study_fft.py ${1:-15} ${2:-64} > $d.log 2>&1)
  echo "bits $bits: $(grep 'study bits' $d.log)" >> $OUT
  python3 - $d >> $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'fft' in r['Name'] or 'spec_mix' in r['Name']:
        print('    %-78s calls %4s avg %8.1f us' % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $OUT

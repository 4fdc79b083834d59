# What would fewer MFMAs buy the mixing kernel?  (make -C poisson_cnn_amd/csrc study_mix; then on the GPU box: bash tools/study_mix_keep.sh)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/study_mix_keep.txt
: > $OUT
for k in 4 3 2 0; do for bits in 0 64; do
  d=$ROOT/gpurun_out/study_keep${k}_$bits
  rm -rf $d
  (cd /tmp && PCNN_LIBRARY=$ROOT/build/study/libpcnn_keep$k.so PCNN_FFT_STUDY=$bits rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python3 $ROOT/tools/study_fft.py 7 32 > $d.log 2>&1)
  echo "keep $k of 4 MFMAs, rows bit $bits: $(grep 'study bits' $d.log)" >> $OUT
  python3 - $d >> $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'spec_mix' in r['Name']:
        print('    %-60s calls %4s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done; done
cat $OUT

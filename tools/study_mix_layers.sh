#!/bin/bash
# Upper bounds on whole layers (tools/probe_xform.py, FFT columns) for two changes of the mixing kernels that were NOT built: Gauss's three-multiplication
# complex product (3 of every 4 matrix k-steps kept) and the imaginary spectrum row stored next to the real one (PCNN_FFT_STUDY bit 64: timing only, the
# results are wrong; the committed source has that switch in spec_mix_kernel only - profiles/r05_study_mix_layers_upper_bounds.txt says how spec_mixw_kernel was
# patched for the measurement).  make -C poisson_cnn_amd/csrc study_mix first.
ROOT=$(pwd)
for k in 4 3; do for bits in 0 64; do
  echo "== keep $k of 4 MFMAs, rows bit $bits"
  PCNN_LIBRARY=$ROOT/build/study/libpcnn_keep$k.so PCNN_FFT_STUDY=$bits python3 tools/probe_xform.py 8 2>&1 | grep -v amdgpu.ids | grep "k 7 32->32 @1024\|k15 32->32\|k 9 24\|sums"
done; done

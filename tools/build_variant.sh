#!/bin/bash
# A/B builds of the spectral kernels:  bash tools/build_variant.sh <name> [-DMACRO=value ...]  ->  build/variants/libpcnn_<name>.so
# (spectral_fft.hip + spectral_conv.hip + spectral64.hip recompiled with the extra flags, everything else from the in-tree objects; select with PCNN_LIBRARY).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants
mkdir -p "$OUT"
cd "$ROOT/poisson_cnn_amd/csrc"
make -j8 > /dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $*"
/opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -c spectral_fft.hip -o "$OUT/spectral_fft_$NAME.o" &
/opt/rocm/bin/hipcc $FLAGS -c spectral_conv.hip -o "$OUT/spectral_conv_$NAME.o" &
/opt/rocm/bin/hipcc $FLAGS -c spectral64.hip -o "$OUT/spectral64_$NAME.o" &
wait
OBJS=$(ls *.o | grep -v -E "^(spectral_fft|spectral_conv|spectral64)\.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$OUT/spectral_fft_$NAME.o" "$OUT/spectral_conv_$NAME.o" "$OUT/spectral64_$NAME.o" -o "$OUT/libpcnn_$NAME.so"
rm -f "$OUT"/*_$NAME.o
echo "$OUT/libpcnn_$NAME.so"

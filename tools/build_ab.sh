#!/bin/bash
# build_ab.sh NAME "EXTRA_FLAGS" [files...] -- an A/B flavour of the library: ab_libs/libdsp_NAME.so
# (DSP_HIP_LIB points the Python layer at it).  Objects whose flags do not differ are NOT shared with
# the product build: every flavour is compiled, LINTED (tools/isa_lint.py) and linked on its own.
set -e
NAME=$1; EXTRA=$2; EXTRA_LAGR=${3-$2}    # third argument: other flags for the Lagrange part
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/dsp_amd/csrc
OUT=$ROOT/ab_libs/$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-inline-asm -Wno-unused-function $EXTRA"
SFLAGS="--offload-device-only -S -Wno-unused-command-line-argument"
pids=()
for f in dsp_api sync_kernels tile_kernels fir_kernels synth_kernels segment_kernels nco_kernels preamble_kernels cfo_kernels; do
  ( hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o && hipcc $FLAGS $SFLAGS $SRC/$f.hip -o $OUT/$f.lint.s ) & pids+=($!)
done
FLAGS_L="${FLAGS% $EXTRA} $EXTRA_LAGR"
for part in 1 2 3 4 5 6 7 8; do
( hipcc $FLAGS_L -DDSP_SYNC_PART=$part -c $SRC/sync_kernels.hip -o $OUT/sync_kernels_p$part.o &&
  hipcc $FLAGS_L -DDSP_SYNC_PART=$part $SFLAGS $SRC/sync_kernels.hip -o $OUT/sync_kernels_p$part.lint.s ) & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
python3 $ROOT/tools/isa_lint.py $OUT/*.lint.s
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/libdsp_$NAME.so $OUT/*.o
rm -f $OUT/*.lint.s $OUT/*.o
echo "built ab_libs/libdsp_$NAME.so"

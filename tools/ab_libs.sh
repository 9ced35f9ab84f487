#!/bin/bash
# ab_libs.sh "B1 B2 ..." lib1 lib2 ... -- the fused kernel's launch time per library flavour, interleaved
# (three rounds per size, so that a box's drift shows), bench.py's own HIP-event timing.  "product" = the
# in-tree libdsp_hip.so; other names are ab_libs/libdsp_NAME.so (tools/build_ab.sh).
SIZES=$1; shift
mkdir -p gpurun_out/ab
for B in $SIZES; do for round in 1 2 3; do for L in "$@"; do
  if [ "$L" = product ]; then LIB=dsp_amd/libdsp_hip.so; else LIB=ab_libs/libdsp_$L.so; fi
  DSP_HIP_LIB=$LIB timeout -k 10 120 python bench.py --no-cpu --no-capture --bursts $B --steps 30 --warmup 5 $AB_FLAGS \
      > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err || { tail -5 gpurun_out/ab/b.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab/b.json')); r=d['roofline']
print('$L', $B, 'round', $round, 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'match', d['config'].get('min_symbol_match_rate'))"
done; done; done

#!/bin/bash
# ab_nospec.sh "B1 B2 ..." -- the linear two-wave kernel with and without its speculative segments
# (ab_libs/libdsp_nospec.so: tools/build_ab.sh nospec "-DDSP_NOSPEC"), interleaved, three rounds per size.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
for B in $1; do for round in 1 2 3; do
  timeout -k 10 120 python3 $R/tools/time_kernel.py $B
  DSP_HIP_LIB=$R/ab_libs/libdsp_nospec.so timeout -k 10 120 python3 $R/tools/time_kernel.py $B spec=0
done; done

#!/bin/bash
# The size / variant sweep behind DESIGN.md section 5's round-3 numbers (one GPU box, no profiler):
#   bash tools/sweep_round3.sh > gpurun_out/sweep_r03.log
# every step under its own timeout; the first failure ends the script.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
for B in 4096 8192 16384 32768 49152 65536 98304 131072 196608; do
    timeout -k 10 120 python3 $R/tools/time_kernel.py $B
done
for H in 1 2 3 4 6 8; do
    timeout -k 10 120 python3 $R/tools/time_kernel.py 65536 half=$H
    timeout -k 10 120 python3 $R/tools/time_kernel.py 32768 half=$H
done
timeout -k 10 120 python3 $R/tools/time_kernel.py 65536 drop=rec,err,mu,sig,freq,ef
timeout -k 10 120 python3 $R/tools/time_kernel.py 65536 drop=rec,err,sig,ef
timeout -k 10 120 python3 $R/tools/bench_sync_c128.py
timeout -k 10 120 python3 $R/tools/bench_fir_c128.py
timeout -k 10 120 python3 $R/tools/bench_latency.py
timeout -k 10 120 python3 $R/tools/bench_rx_front.py 14
timeout -k 10 120 python3 $R/tools/bench_diff_corr.py

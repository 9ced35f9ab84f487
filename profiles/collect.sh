#!/bin/bash
# Collect the round's rocprofv3 evidence for the headline kernel on the GPU box:
#   bash profiles/collect.sh rNN        (run through gpurun from the repository root)
# kernel trace + stats of the default bench.py run, then HBM traffic counters in separate --pmc
# passes (FETCH_SIZE / WRITE_SIZE cannot share a pass with the SQ counters), summarised into
# profiles/rNN_* by profiles/summarize.py.  Outputs land in gpurun_out/prof_rNN (scratch).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --no-cpu --no-capture --no-parts --steps 20 --warmup 3 > $O/kt_bench.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --no-cpu --no-capture --no-parts --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --no-cpu --no-capture --no-parts --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_sq -o p --output-format csv -- python3 $R/bench.py --no-cpu --no-capture --no-parts --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
grep -h '^{' $O/kt_bench.log | tail -1 > $O/bench_line.json
echo collected $O

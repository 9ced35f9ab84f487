#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, summary printed and kept under gpurun_out/<tag>/:
#   bash tools/ktrace.sh <tag> -- python3 /root/repo/tools/bench_fir.py 16 both
set -e
TAG=$1; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o p --output-format csv -- "$@" > $O/run.log 2>&1
f=$(find /tmp/kt_$TAG -name '*kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%10.1f us avg %6d calls %6.2f%%  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"]), r["Name"][:120]))
PY
tail -2 $O/run.log

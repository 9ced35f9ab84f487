#!/bin/bash
# rocprofv3 --kernel-trace of one command; prints every kernel of the LAST call sequence with its start offset,
# duration and the gap to its predecessor (where the host, not the GPU, sets the pace):
#   bash tools/ktrace_gaps.sh <tag> <first-kernel-substring> -- python3 /root/repo/tools/bench_capture.py 30 16384
set -e
TAG=$1; FIRST=$2; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktg_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/ktg_$TAG -o p --output-format csv -- "$@" > $O/run.log 2>&1
f=$(find /tmp/ktg_$TAG -name '*kernel_trace.csv' | head -1)
python3 - $f "$FIRST" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i0 = starts[-1] if starts else max(0, len(rows) - 40)
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
print("span %.1f us" % ((prev_end - t0) / 1e3))
PY
tail -1 $O/run.log | cut -c200-330

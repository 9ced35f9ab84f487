set -e
R=$PWD; mkdir -p $R/gpurun_out/lat
timeout -k 10 120 python tools/bench_latency.py
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/lat -o p --output-format csv -- python3 $R/tools/bench_latency.py > $R/gpurun_out/lat/log.txt 2>&1
f=$(find /tmp/lat -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s %5s %9.3f ms avg %9.3f total" % (r["Name"][:68], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY

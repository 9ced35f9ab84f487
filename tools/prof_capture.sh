# rocprofv3 kernel stats of tools/bench_capture.py on the GPU box:
#   bash tools/prof_capture.sh <log2_samples> <seg_len, 0 = auto> <tag> [warm-up symbols] [trim samples]
# leaves gpurun_out/cap/kstats_<tag>.csv, gpurun_out/cap/kstats_<tag>.txt (the table kept under profiles/)
set -e
R=$PWD; mkdir -p $R/gpurun_out/cap
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pc_$3 -o p --output-format csv -- python3 $R/tools/bench_capture.py $1 $2 ${4:-4096} ${5:-0} > $R/gpurun_out/cap/prof_$3.log 2>&1
f=$(find /tmp/pc_$3 -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/cap/kstats_$3.csv
python3 - $R/gpurun_out/cap/kstats_$3.csv $R/gpurun_out/cap/prof_$3.log > $R/gpurun_out/cap/kstats_$3.txt <<'PY'
import csv, json, sys
line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
d = json.loads(line)
print("workload:", d["workload"])
print("whole call under the profiler (HIP events around dsp_sync_capture_segmented_dev): %.2f ms, %d windows" % (d["ms"], d["segments"]))
print()
print("%-100s %5s %9s %7s" % ("kernel", "calls", "avg ms", "%"))
for r in csv.DictReader(open(sys.argv[1])):
    print("%-100s %5s %9.3f %7.2f" % (r["Name"][:98], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
cat $R/gpurun_out/cap/kstats_$3.txt
# the timed call's timeline: every dispatch with its start relative to the call's first kernel and the idle gap in front of it
t=$(find /tmp/pc_$3 -name '*kernel_trace.csv' | head -1)
python3 - $t > $R/gpurun_out/cap/timeline_$3.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
peaks = [i for i, r in enumerate(rows) if "iq_i16_peak" in r["Kernel_Name"]]
rows = rows[peaks[-1]:]                      # the last (timed) call starts with the peak search
t0, prev, gaps = int(rows[0]["Start_Timestamp"]), None, 0.0
print("%-70s %10s %9s %9s" % ("kernel", "start us", "dur us", "gap us"))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    gaps += max(gap, 0.0)
    print("%-70s %10.1f %9.1f %9.1f" % (r["Kernel_Name"][:68], (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev = e
print("first kernel start -> last kernel end: %.1f us, idle gaps %.1f us" % ((prev - t0) / 1e3, gaps))
PY
cat $R/gpurun_out/cap/timeline_$3.txt

#!/bin/bash
# pmc_copy.sh -- the same-box copy ceiling with counters: tools/ubench/copy_bw under rocprofv3,
# kernel trace + FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE in passes of their own, per copy shape:
# duration, HBM bytes moved, busy cycles -> clock.  Output: gpurun_out/copy_pmc/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/copy_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp_kt /tmp/cp_f /tmp/cp_w
timeout -k 10 120 $R/tools/ubench/copy_bw 4 > $O/plain.txt 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/cp_kt -o p --output-format csv -- $R/tools/ubench/copy_bw 4 > $O/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /tmp/cp_f -o p --output-format csv -- $R/tools/ubench/copy_bw 4 > $O/f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d /tmp/cp_w -o p --output-format csv -- $R/tools/ubench/copy_bw 4 > $O/w.log 2>&1
cp $(find /tmp/cp_kt -name '*kernel_trace.csv') $O/kernel_trace.csv
cp $(find /tmp/cp_f -name '*counter_collection.csv') $O/fetch.csv
cp $(find /tmp/cp_w -name '*counter_collection.csv') $O/write.csv
python3 - $O <<'PY' | tee $O/summary.txt
import csv, sys, collections, os
O = sys.argv[1]
print(open(os.path.join(O, "plain.txt")).read())
# kernel trace: launches in order; group by (kernel name, grid)
tr = list(csv.DictReader(open(os.path.join(O, "kernel_trace.csv"))))
def key(r): return (r["Kernel_Name"][:60], r.get("Grid_Size") or r.get("Grid_Size_X"))
dur = collections.defaultdict(list)
for r in tr: dur[key(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
pdur = collections.defaultdict(list)
for f in ("fetch.csv", "write.csv"):
    seen = set()
    for r in csv.DictReader(open(os.path.join(O, f))):
        k = (r["Kernel_Name"][:60], r.get("Grid_Size") or r.get("Grid_Size_X"))
        cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = (k, r["Dispatch_Id"], f)
        if d not in seen:
            seen.add(d); pdur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
BYTES = 4 * 2**30
print("%-62s %9s %9s %11s %11s %9s" % ("kernel (first 60 chars), grid", "us", "TB/s r+w", "fetch GB x2", "write GB", "clock MHz"))
for k, d in dur.items():
    us = min(d) / 1e3
    c = cnt.get(k, {})
    fe = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 * 2 / 1e9 if "FETCH_SIZE" in c else float("nan")
    wr = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024 / 1e9 if "WRITE_SIZE" in c else float("nan")
    clk = float("nan")
    if "GRBM_GUI_ACTIVE" in c and pdur.get(k):
        clk = (sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])) / (sum(pdur[k]) / len(pdur[k]) / 1e3)
    print("%-62s %9.1f %9.3f %11.3f %11.3f %9.0f" % (k[0] + " " + str(k[1]), us, 2 * BYTES / (us * 1e-6) / 1e12, fe, wr, clk))
PY

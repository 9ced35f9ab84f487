#!/usr/bin/env python3
"""profiles/rNN_capture_2p30.json from the kernel trace (tools/prof_capture.sh) and the PMC passes
(tools/pmc_passes.sh <tag> "" -- python3 tools/bench_capture.py 30) of the segment-parallel capture call.
Usage: python profiles/summarize_capture.py <kstats.csv> <kstats.txt> <pmc pass dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

kcsv, ktxt, pdir, out = sys.argv[1:5]
head = open(ktxt).read().splitlines()
call_ms = float(re.search(r"([\d.]+) ms", head[1]).group(1))
n_samples = 1 << 30
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(pdir, "pass*.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = []
tot_f = tot_w = 0.0
for r in csv.DictReader(open(kcsv)):
    name = r["Name"]
    if "synth_stream" in name or name.startswith("__amd"):
        continue                      # the generator and runtime copies are not part of the call's work
    c = acc.get(name, {})
    fk = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0])))
    wk = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0])))
    # KiB counters; FETCH_SIZE x2 on gfx950 for wide streaming reads (MI355X_MICROARCH.md, HBM section)
    fb, wb = 2 * fk * 1024, wk * 1024
    per_call = int(r["Calls"]) / 2    # bench_capture.py makes two calls (one warm-up)
    tot_f += fb * per_call
    tot_w += wb * per_call
    kernels.append({"name": name[:120], "launches_per_call": per_call, "avg_ms": float(r["AverageNs"]) / 1e6,
                    "fetch_bytes": fb, "write_bytes": wb})
alg = n_samples * 4 + (n_samples // 4) * (1 + 4 + 4)     # int16 IQ in; decisions + mu + freq per symbol out
json.dump({"workload": head[0].replace("workload: ", ""), "call_ms": call_ms, "samples": n_samples,
           "hbm_bytes_per_call": {"fetch": tot_f, "write": tot_w, "total": tot_f + tot_w,
                                  "per_sample": (tot_f + tot_w) / n_samples},
           "algorithmic_bytes_per_call": alg, "traffic_over_algorithmic": (tot_f + tot_w) / alg,
           "kernels": kernels}, open(out, "w"), indent=1)
print(open(out).read()[:1500])

#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (csv) into the small text/JSON files kept under
profiles/.  Usage: python profiles/summarize.py <prof_dir> <out_prefix> [kernel-substring]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (stdlib-only import: the stamp bench.py checks)

prof, out = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "sync_"
summary = {"source": os.path.basename(prof.rstrip("/")), "kernel_filter": want}

ks = glob.glob(os.path.join(prof, "kt", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    summary["kernel_stats"] = [
        {"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
         "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
         "pct": float(r["Percentage"])} for r in rows]

pmc = {}
for d in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(f[0])):
        if want in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        pmc[k] = {"per_launch": sum(v) / len(v), "launches": len(v)}
    if dur:
        pmc.setdefault("_duration_us", {})[os.path.basename(d)] = sum(dur) / len(dur) / 1e3
summary["pmc"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch_kib, write_kib = pmc["FETCH_SIZE"]["per_launch"], pmc["WRITE_SIZE"]["per_launch"]
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
    # exactly half the bytes of a wide (16 B/lane) coalesced streaming read -> double it.
    summary["hbm_traffic"] = {
        "fetch_bytes_raw": fetch_kib * 1024, "fetch_bytes_corrected": 2 * fetch_kib * 1024,
        "write_bytes": write_kib * 1024,
        "total_bytes_corrected": 2 * fetch_kib * 1024 + write_kib * 1024,
        "note": "reads are 16 B/lane LDS-DMA streams (x2 correction applies); the stores are 4/8/1 "
                "B per lane (WRITE_SIZE calibrated only for 16 B/lane: treat as indicative)"}
bl = os.path.join(prof, "bench_line.json")
if os.path.exists(bl):
    try:
        summary["bench_line"] = json.load(open(bl))
    except ValueError:
        pass
json.dump(summary, open(out + ".json", "w"), indent=1)
if "hbm_traffic" in summary and "bench_line" in summary:
    cfg = summary["bench_line"]["config"]
    json.dump({"kernel": summary["bench_line"]["roofline"]["kernel"],
               "bursts_per_gpu": cfg["bursts_per_gpu"], "samples_per_burst": cfg["samples_per_burst"],
               "hbm_bytes_per_launch": summary["hbm_traffic"]["total_bytes_corrected"],
               "fetch_bytes_corrected": summary["hbm_traffic"]["fetch_bytes_corrected"],
               "write_bytes": summary["hbm_traffic"]["write_bytes"],
               # bench.py reports this figure only while the kernel's sources still hash to this
               "kernel_source_sha256": kernel_source_hash(),
               "source": os.path.basename(out) + ".json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                         "FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section)"},
              open(os.path.join(os.path.dirname(out) or ".", "traffic.json"), "w"), indent=1)
with open(out + ".txt", "w") as fh:
    for r in summary.get("kernel_stats", []):
        fh.write(f"{r['avg_us']:12.1f} us avg  {r['calls']:5d} calls  {r['pct']:6.2f}%  {r['name'][:110]}\n")
    fh.write("\n")
    for k, v in pmc.items():
        fh.write(f"{k}: {v}\n")
    if "hbm_traffic" in summary:
        fh.write(f"\nhbm_traffic: {summary['hbm_traffic']}\n")
print(open(out + ".txt").read())

#!/bin/bash
# rocprofv3 counter passes over one command, one pass per run (FETCH_SIZE / WRITE_SIZE cannot share a
# pass with each other; at most 8 SQ counters a pass), summarised per kernel:
#   bash tools/pmc_passes.sh <out-dir-under-gpurun_out> <kernel-substring> -- python3 tools/bench_fir.py 16
# The program after -- must be the program itself (no env / bash -c wrappers under the profiler).
set -e
TAG=$1; KSUB=$2; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf /tmp/pmc_${TAG}_$i
    timeout -s KILL 180 rocprofv3 --pmc $set -d /tmp/pmc_${TAG}_$i -o p --output-format csv -- "$@" > $O/pass$i.log 2>&1
    n=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | wc -l)
    [ "$n" = 1 ] || { echo "pass $i: $n counter files (expected 1), see $O/pass$i.log"; exit 1; }
    cp "$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv')" $O/pass$i.csv
done
python3 - $O "$KSUB" <<'PY' | tee $O/summary.txt
import csv, sys, collections, glob, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pass*.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sys.argv[2] not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in acc.items():
    print(k[:150])
    print("   launches per pass ~%d, mean duration under the profiler %.1f us" % (len(dur[k]) / 4, sum(dur[k]) / len(dur[k]) / 1e3))
    for c, v in sorted(d.items()):
        print("   %-24s %18.0f per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fk, wk = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        print("   HBM bytes per launch: fetch %.4g (KiB counter x 1024 x 2, gfx950 correction for wide streaming reads: %.4g), write %.4g"
              % (fk * 1024, 2 * fk * 1024, wk * 1024))
PY

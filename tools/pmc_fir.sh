# one --pmc pass over tools/bench_fir.py (fir_tile_kernel): LDS and VALU counters
set -e
R=$PWD; mkdir -p $R/gpurun_out/firpmc
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d /tmp/firpmc -o p --output-format csv -- python3 $R/tools/bench_fir.py 16 > $R/gpurun_out/firpmc/log.txt 2>&1
f=$(find /tmp/firpmc -name '*counter_collection.csv' | head -1)
python3 - $f <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "fir_tile" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %16.0f per launch" % (c, v / n[(k, c)]))
PY

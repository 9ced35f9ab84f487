# capture-mode sweep: log2(samples), seg_len (0 = auto), warm-up symbols
mkdir -p gpurun_out/cap
for cfg in "30 0 4096" "30 0 2048" "30 65536 2048" "30 65536 4096" "28 0 4096" "28 0 2048" "26 0 4096" "24 0 4096" "22 0 4096"; do
  timeout -k 10 120 python tools/bench_capture.py $cfg > gpurun_out/cap/s.log 2>&1 || { cat gpurun_out/cap/s.log; exit 1; }
  python - "$cfg" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/cap/s.log").read().strip().splitlines()[-1])
print(sys.argv[1], "|", d["workload"].split("segment-parallel: ")[1], "|", d["segments"], round(d["ms"],3), d["symbol_match_rate"])
PY
done

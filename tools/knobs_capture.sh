mkdir -p gpurun_out/cap
for o in "prio_period=1" "prio_period=2" "prio_period=8" "prio_period=16" "prio_period=64" "prio_period=4,queue_block=64"; do
  DSP_CAPTURE_OPTS=$o timeout -k 10 120 python tools/bench_capture.py 30 > gpurun_out/cap/s.log 2>&1 || { cat gpurun_out/cap/s.log; exit 1; }
  python - "$o" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/cap/s.log").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(d["ms"],3), d["symbol_match_rate"])
PY
done

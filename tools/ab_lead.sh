mkdir -p gpurun_out/cap
timeout -k 10 300 python -u -m pytest tests/test_segmented.py tests/test_gpu_capture_fullsize.py tests/test_gpu_fuzz.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/cap/t.log 2>&1; tail -3 gpurun_out/cap/t.log
grep -q failed gpurun_out/cap/t.log && exit 1
for lg in 30 28; do for o in "lead_copy=1" "lead_copy=0"; do
  DSP_CAPTURE_OPTS=$o timeout -k 10 120 python tools/bench_capture.py $lg > gpurun_out/cap/s.log 2>&1 || { cat gpurun_out/cap/s.log; exit 1; }
  python - "$lg $o" <<'PY'
import json, sys
d=json.loads(open("gpurun_out/cap/s.log").read().strip().splitlines()[-1])
print(sys.argv[1], "|", round(d["ms"],3), d["symbol_match_rate"])
PY
done; done

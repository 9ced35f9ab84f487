mkdir -p gpurun_out/sw
timeout -k 10 400 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sync.py tests/test_segmented.py tests/test_gpu_capture_fullsize.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/sw/t.log 2>&1; tail -3 gpurun_out/sw/t.log
grep -q passed gpurun_out/sw/t.log || exit 1
grep -q "failed\|error" gpurun_out/sw/t.log && exit 1
for B in 20000 32768 65536; do for sw in 0 1 2 -1; do
  timeout -k 10 120 python bench.py --no-cpu --bursts $B --spare-wave $sw --steps 20 --warmup 3 > gpurun_out/sw/b.json 2>gpurun_out/sw/b.err || { tail -5 gpurun_out/sw/b.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/sw/b.json')); print($B, $sw, round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['config']['min_symbol_match_rate'])"
done; done
for sw in 1 2; do DSP_CAPTURE_OPTS=spare_wave=$sw timeout -k 10 120 python tools/bench_capture.py 30 > gpurun_out/sw/c.log 2>&1 || exit 1; python -c "
import json; d=json.loads(open('gpurun_out/sw/c.log').read().strip().splitlines()[-1]); print('capture 2^30 spare', $sw, round(d['ms'],3))"; done

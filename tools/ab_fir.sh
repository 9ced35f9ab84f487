mkdir -p gpurun_out/fir
timeout -k 10 300 python -u -m pytest tests/test_gpu_fir.py tests/test_segmented.py tests/test_gpu_capture_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/fir/t.log 2>&1; tail -12 gpurun_out/fir/t.log
grep -q failed gpurun_out/fir/t.log && exit 1
timeout -k 10 120 python tools/bench_fir.py 16 || exit 1
timeout -k 10 120 python tools/bench_capture.py 30 | cut -c300-420

mkdir -p gpurun_out/ab
run() { timeout -k 10 120 python bench.py --no-cpu --no-capture --bursts 16384 --steps 30 --warmup 5 "$@" > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err || { tail -5 gpurun_out/ab/b.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/ab/b.json')); r=d['roofline']
print(' '.join(sys.argv[1:]), 'kernel_ms', round(r['kernel_ms'],4))" "$@"; }
run --ring-rows 128 --queue-block 1
run --ring-rows 128 --queue-block 2
run --ring-rows 128 --queue-block 4
run --ring-rows 128 --queue-block 7
run --ring-rows 64 --queue-block 6
run --ring-rows 64 --queue-block 3
run --ring-rows 64 --queue-block 2
run --ring-rows 64 --queue-block 6 --lean

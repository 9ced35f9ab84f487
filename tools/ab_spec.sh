#!/bin/bash
# ab_spec.sh "B1 B2 ..." -- speculative segments on / off, interleaved, bench.py's HIP-event time per launch
mkdir -p gpurun_out/ab
for B in $1; do for round in 1 2; do for sp in 0 1; do
  timeout -k 10 120 python bench.py --no-cpu --no-capture --bursts $B --steps 30 --warmup 5 --spec $sp $AB_FLAGS \
      > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err || { tail -5 gpurun_out/ab/b.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab/b.json')); r=d['roofline']
print('spec', $sp, 'B', $B, 'round', $round, 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'match', d['config'].get('min_symbol_match_rate'))"
done; done; done

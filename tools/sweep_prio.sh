#!/bin/bash
# priority rotation period x batch size (and the lean / Lagrange forms) of the recurrence kernel
# usage (GPU box, repo root): bash tools/sweep_prio.sh > gpurun_out/sweep_prio.txt
set -u
run() {
    local tag="$1"; shift
    out=$(timeout -k 10 120 python -u bench.py --no-cpu --no-capture --fir-reps 1 "$@" 2>/dev/null) || { echo "$tag FAILED"; exit 1; }
    python - "$tag" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:40s} {d['ms_per_step']:.4f} ms  frac {d['roofline']['frac']:.4f}", flush=True)
PY
}
for b in 4096 32768 49152 65536 98304; do
    for p in 0 8 16; do run "B $b prio $p" --bursts $b --prio-period $p; done
done
for p in 0 8 16; do run "lean prio $p" --lean --prio-period $p; done
for p in 0 8 16; do run "lagrange h=1 prio $p" --interp-half-len 1 --prio-period $p; done
for p in 0 2 8 16 32; do run "B 65536 prio $p (2nd)" --prio-period $p; done

#!/bin/bash
# option sweep of the recurrence kernel at the headline size; one JSON line per setting
# usage (GPU box, repo root): bash tools/sweep_opts.sh > gpurun_out/sweep_opts.txt
set -u
run() {
    local tag="$1"; shift
    out=$(timeout -k 10 120 python -u bench.py --no-cpu --no-capture --fir-reps 1 "$@" 2>/dev/null) || { echo "$tag FAILED"; exit 1; }
    python - "$tag" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:40s} {d['ms_per_step']:.4f} ms  frac {d['roofline']['frac']:.4f}")
PY
}
run "default"
run "prio 4" --prio-period 4
run "prio 16" --prio-period 16
run "prio 64" --prio-period 64
run "queue 4" --queue-block 4
run "queue 8" --queue-block 8
run "queue 16" --queue-block 16
run "ring 64" --ring-rows 64
run "ring 32" --ring-rows 32
run "spare 0" --spare-wave 0
run "spare 1" --spare-wave 1
run "spare 2" --spare-wave 2
run "default again"

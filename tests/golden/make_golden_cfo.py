#!/usr/bin/env python3
"""Golden vectors for the coarse frequency estimators, produced by IMPORTING
the reference here (never shipped):
  * estimate_freq.estimate_initial_frequency (estimate_freq.py:4-58), fourth-power lag-1 estimator;
  * the preamble-aided f_rel that worker.Processing.process prints per found burst
    (worker.py:288-295), on the seeded worker.Generation signal of make_golden_preamble.py.
The three preamble-aided estimators of diff_method.py:267-292 on the config 2 capture are already
in g_config2.npz (f_rel, aligned, derot) with the preamble in g_diffcorr.npz (pre).
Run once in the build container:  python tests/golden/make_golden_cfo.py  -> tests/golden/g_cfo.npz
"""
import contextlib
import io
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.chdir(REF)
sys.path.insert(0, REF)
import matplotlib  # noqa: E402

matplotlib.use = lambda *a, **k: None          # diff_method.py:4 asks for TkAgg (no tkinter here)
with contextlib.redirect_stdout(io.StringIO()):
    import estimate_freq as ref_ef  # noqa: E402
    import worker as ref_w  # noqa: E402

out = {}
# ---- fourth-power estimator: the module's own demo shape (estimate_freq.py:60-63) seeded, plus noise
rng = np.random.default_rng(4)
sym = rng.choice(np.array([1 + 1j, 1 - 1j, -1 + 1j, -1 - 1j]), 4000) / np.sqrt(2)
x = sym * np.exp(1j * 2 * np.pi * 0.005242 * np.arange(4000))
x = x + 0.05 * (rng.standard_normal(4000) + 1j * rng.standard_normal(4000))
out["fp_x"] = x
ns = np.array([4000, 1000, 257, 2, 1, 0, 10000], dtype=np.int64)      # > len(signal): clipped (:32); < 2: 0.0 (:33)
with contextlib.redirect_stdout(io.StringIO()):
    out["fp_f"] = np.array([ref_ef.estimate_initial_frequency(x, int(n)) for n in ns], dtype=np.float64)
out["fp_n"] = ns
# a batch of rows with different offsets (each row = one call of the reference)
rows = np.stack([sym[i * 300:(i + 1) * 300] * np.exp(1j * 2 * np.pi * f * np.arange(300))
                 for i, f in enumerate([-0.02, -0.004, 0.0, 0.0031, 0.011, 0.03])])
rows = rows + 0.1 * (rng.standard_normal(rows.shape) + 1j * rng.standard_normal(rows.shape))
with contextlib.redirect_stdout(io.StringIO()):
    out["fp_rows"] = rows
    out["fp_rows_f"] = np.array([ref_ef.estimate_initial_frequency(r, 300) for r in rows])

# ---- worker.Processing.process: f_rel per found burst (printed at worker.py:295)
np.random.seed(20251114)
gen = ref_w.Generation(constellation="qpsk", preamble_length=64, signal_length=400, sps=4, Frel=0.0004,
                       Noise_power=0.02, N=3, Delay=30)
with contextlib.redirect_stdout(io.StringIO()):
    w_pre, w_samples, w_symbols = gen.generate()
for interp in (False, True):
    pr = ref_w.Processing(signal=w_samples, preamble=w_pre, packet_symb=w_symbols, sps=4, constellation="qpsk",
                          interp=interp)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        peaks, found = pr.process()
    vals = [float(v) for v in re.findall(r"f_rel ([-+0-9.eE]+)", buf.getvalue())]
    out["w_frel_" + ("i" if interp else "n")] = np.array(vals, dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "g_cfo.npz"), **out)
print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
print(out["fp_f"], out["fp_rows_f"], out["w_frel_n"], out["w_frel_i"])

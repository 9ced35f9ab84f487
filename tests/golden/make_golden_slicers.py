#!/usr/bin/env python3
"""Golden vectors for the constellation switch of worker.Processing.fll_func / decision_direct
(worker.py:18-170), produced by IMPORTING the reference here (never shipped).
For each of bpsk / qpsk / 8psk / 16psk / 16apsk: a seeded symbol stream with a small frequency
offset and noise -> Processing(constellation=c).fll_func(x, Bn, 1) outputs, and the decisions
decision_direct makes on the loop output (read from its own print, with numpy print options opened
up so nothing is elided) together with the match percentages it prints per trial rotation.
Run once in the build container: python tests/golden/make_golden_slicers.py -> g_slicers.npz
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
    import worker as ref_w  # noqa: E402
import matplotlib.pyplot as plt  # noqa: E402

np.set_printoptions(threshold=sys.maxsize, linewidth=10**6,
                    formatter={"complex_kind": lambda z: f"{float(z.real)!r}|{float(z.imag)!r}"})


def points(c):
    if c == "bpsk":
        return np.array([1 + 0j, -1 + 0j])
    if c == "qpsk":
        return np.array([1 + 1j, -1 + 1j, -1 - 1j, 1 - 1j]) / np.sqrt(2)
    if c == "8psk":
        return np.exp(1j * (np.pi / 4 * np.arange(8) - np.pi))
    if c == "16psk":
        return np.exp(1j * (np.pi / 8 * np.arange(16) - np.pi))
    return np.concatenate([0.5 * np.exp(1j * (np.pi / 4 + np.arange(4) * np.pi / 2)),
                           1 * np.exp(1j * (np.pi / 12 + np.arange(12) * np.pi / 6))])


out = {}
rng = np.random.default_rng(77)
M = 700
for c, bn, cfo, sigma in (("bpsk", 0.03, 6e-4, 0.05), ("qpsk", 0.03, 8e-4, 0.05), ("8psk", 0.03, 5e-4, 0.03),
                          ("16psk", 0.02, 2e-4, 0.012), ("16apsk", 0.02, 2e-4, 0.015)):
    pts = points(c)
    sym = pts[rng.integers(0, len(pts), M)]
    x = sym * np.exp(1j * (2 * np.pi * cfo * np.arange(M) + 0.05))
    x = x + sigma * (rng.standard_normal(M) + 1j * rng.standard_normal(M))
    x[5] = 0.0                                        # np.sign(0) = 0 / equidistant from every PSK point
    pr = ref_w.Processing(signal=None, preamble=None, packet_symb=sym.astype(np.complex64), sps=4, constellation=c)
    with contextlib.redirect_stdout(io.StringIO()):
        sl, cf, cfl, ef = pr.fll_func(x, Bn=bn, y=1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pr.decision_direct(sl)
    plt.close("all")
    txt = buf.getvalue()
    body = txt[txt.index("decision_direct [") + len("decision_direct ["):]
    body = body[:body.index("]")]
    dec = np.array([complex(*map(float, t.split("|"))) for t in body.split()], dtype=np.complex64)
    assert dec.size == M, dec.size
    rates = [float(v) for v in re.findall(r"count, ([0-9.eE+-]+)% ", txt)]
    out.update({f"{c}_x": x, f"{c}_sym": sym, f"{c}_bn": np.float64(bn), f"{c}_sig": sl, f"{c}_cf": np.float64(cf),
                f"{c}_freq": cfl, f"{c}_ef": ef, f"{c}_dec": dec, f"{c}_rates": np.array(rates)})
    print(c, "cf", cf, "rates", rates)

# ---- end to end with a table slicer: seeded worker.Generation('8psk') -> worker.Processing.process
np.random.seed(20251115)
gen = ref_w.Generation(constellation="8psk", preamble_length=64, signal_length=300, sps=4, Frel=0.0003,
                       Noise_power=0.02, N=2, Delay=30)
with contextlib.redirect_stdout(io.StringIO()):
    p8, s8, y8 = gen.generate()
pr = ref_w.Processing(signal=s8, preamble=p8, packet_symb=y8, sps=4, constellation="8psk", interp=False)
with contextlib.redirect_stdout(io.StringIO()):
    peaks, found = pr.process()
plt.close("all")
out.update(w8_pre=p8, w8_samples=s8, w8_symbols=y8, w8_peaks=np.asarray(peaks, dtype=np.int64),
           w8_nfound=np.int64(len(found)))
for j, f in enumerate(found):
    out[f"w8_found{j}"] = np.asarray(f)
print("8psk worker: peaks", peaks, "found", len(found))
np.savez_compressed(os.path.join(HERE, "g_slicers.npz"), **out)

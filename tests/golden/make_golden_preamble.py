#!/usr/bin/env python3
"""Golden vectors for the differential-correlation preamble search:
diff_method.find_preamble_offset (diff_method.py:92-162) and find_optimal_sampling_phase
(:41-89), produced by IMPORTING the reference here (never shipped).  Run once in the build
container:  python tests/golden/make_golden_preamble.py   -> tests/golden/g_diffcorr.npz
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
os.chdir(REF)
sys.path.insert(0, REF)
import matplotlib  # noqa: E402

matplotlib.use = lambda *a, **k: None          # diff_method.py:4 asks for TkAgg (no tkinter here)
with contextlib.redirect_stdout(io.StringIO()):
    import diff_method as ref_dm  # noqa: E402
from scipy.signal import firwin, lfilter  # noqa: E402

# config 2 flow, diff_method.py:164-190
raw = np.fromfile("files/sig_symb_x4_ncr_77671952566_logon_id_1_tx_id_7616.pcm", dtype=np.float32)
sig = raw[::2] + 1j * raw[1::2]
pre = np.fromfile("files/preambule_logon_id_2_tx_id_7616_float32.pcm", dtype=np.float32)
pre = pre[::2] + 1j * pre[1::2]
sig = sig * np.exp(-1j * 2 * np.pi * 0.0105 * np.arange(len(sig)))
lpf = firwin(101, cutoff=0.14, window="hamming", fs=1.0)
sig = lfilter(lpf, 1.0, sig)
sig = sig / np.std(sig)
N = 12000                                        # the burst sits at symbol 518 of phase 1
with contextlib.redirect_stdout(io.StringIO()):
    offset, aligned, conv, conv_max, phase = ref_dm.find_preamble_offset(sig[:N], pre, 4)
    offset_full, _, _, conv_max_full, phase_full = ref_dm.find_preamble_offset(sig, pre, 4)
out = dict(sig=sig[:N], pre=pre, offset=np.int64(offset), phase=np.int64(phase),
           conv_max=np.array(conv_max, dtype=np.float64), n_aligned=np.int64(len(aligned)),
           offset_full=np.int64(offset_full), phase_full=np.int64(phase_full), n_full=np.int64(len(sig)),
           conv_max_full=np.array(conv_max_full, dtype=np.float64))
for p in range(4):
    out[f"conv{p}"] = conv[p]
bp, corr = ref_dm.find_optimal_sampling_phase(aligned, pre, 4)
out["best_phase"] = np.int64(bp)
out["phase_corr"] = corr

# synthetic: complex64 input (numpy keeps complex64 products), planted preamble, a dead (all-zero)
# stretch whose windows have std == 0 (diff_method.py:145-148), sps 2
rng = np.random.default_rng(11)
p2 = np.exp(1j * np.pi / 2 * rng.integers(0, 4, 24)).astype(np.complex64)
x = (0.3 * (rng.standard_normal(900) + 1j * rng.standard_normal(900))).astype(np.complex64)
x[200:200 + 2 * 24:2] += p2 * np.exp(1j * 0.7)
x[600:700] = 0
with contextlib.redirect_stdout(io.StringIO()):
    o2, al2, c2, cm2, ph2 = ref_dm.find_preamble_offset(x, p2, 2)
out.update(syn_x=x, syn_pre=p2, syn_offset=np.int64(o2), syn_phase=np.int64(ph2),
           syn_conv0=c2[0], syn_conv1=c2[1], syn_conv_max=np.array(cm2, dtype=np.float64))
np.savez_compressed(os.path.join(HERE, "g_diffcorr.npz"), **out)
print("offset", offset, "phase", phase, "full", offset_full, phase_full, "syn", o2, ph2,
      {k: (v.shape, str(v.dtype)) for k, v in out.items() if hasattr(v, "shape") and v.ndim})

# ---------------------------------------------------------------------------
# worker.Processing.process (worker.py:270-337) on a seeded worker.Generation signal: the
# reference's end-to-end receiver (preamble search -> peaks -> per-burst f_rel, derotation,
# matched filter, Gardner, FLL).  np.random is seeded; prints are swallowed.
with contextlib.redirect_stdout(io.StringIO()):
    import worker as ref_w  # noqa: E402
np.random.seed(20251114)
gen = ref_w.Generation(constellation="qpsk", preamble_length=64, signal_length=400, sps=4, Frel=0.0004,
                       Noise_power=0.02, N=3, Delay=30)
with contextlib.redirect_stdout(io.StringIO()):
    w_pre, w_samples, w_symbols = gen.generate()
wout = dict(w_pre=w_pre, w_samples=w_samples, w_symbols=w_symbols)
for interp in (False, True):
    pr = ref_w.Processing(signal=w_samples, preamble=w_pre, packet_symb=w_symbols, sps=4, constellation="qpsk",
                          interp=interp)
    with contextlib.redirect_stdout(io.StringIO()):
        peaks, found = pr.process()
    tag = "i" if interp else "n"
    wout[f"w_peaks_{tag}"] = np.asarray(peaks, dtype=np.int64)
    wout[f"w_nfound_{tag}"] = np.int64(len(found))
    for j, s in enumerate(found):
        wout[f"w_found_{tag}{j}"] = np.asarray(s)
np.savez_compressed(os.path.join(HERE, "g_worker.npz"), **wout)
print("worker:", {k: (getattr(v, "shape", None)) for k, v in wout.items()})

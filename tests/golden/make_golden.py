#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (the reference does not exist on the GPU box):

    MPLBACKEND=Agg python3 tests/golden/make_golden.py

It imports the reference's own modules from /root/reference (read-only;
cwd is switched there because the scripts open their data files by relative
name), runs the hot-path functions on (i) the reference's bundled .pcm data
files and (ii) seeded synthetic bursts, and stores inputs + float64 outputs as
compressed .npz.  The bundled .pcm data files the scenarios read are copied to
tests/golden/data/ (they are data, not source).  Nothing but this script, the
.npz vectors and the data files is committed.

Environment: CPython 3.10.12, numpy 2.2.6, scipy 1.15.3, glibc libm.
"""
import contextlib
import io
import os
import shutil
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
DATA = os.path.join(HERE, "data")
os.makedirs(DATA, exist_ok=True)
sys.path.insert(0, REPO)

os.environ.setdefault("MPLBACKEND", "Agg")
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "hw1"))

import matplotlib  # noqa: E402

matplotlib.use = lambda *a, **k: None  # diff_method.py:4 asks for TkAgg (no tkinter here)

with contextlib.redirect_stdout(io.StringIO()):
    import fll_func as ref_fll  # noqa: E402
    import gardner2 as ref_g2  # noqa: E402
    import read_file as ref_rf  # noqa: E402
    import diff_method as ref_dm  # noqa: E402
    import gardner as ref_hw1g  # noqa: E402  (hw1/gardner.py; runs its script body)
    import synthesys as ref_syn  # noqa: E402  (hw1/synthesys.py)
    import test_lagrange_resampling as ref_lag  # noqa: E402

from dsp_amd.synth import synth_burst  # noqa: E402  (our seeded recipe, plain numpy)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB  " +
          ", ".join(f"{k}{getattr(v, 'shape', '')}" for k, v in arrs.items()))


def quad(z):
    return ((z.real < 0) + 2 * (z.imag < 0)).astype(np.uint8)


def copy_data(rel):
    dst = os.path.join(DATA, os.path.basename(rel))
    if not os.path.exists(dst):
        shutil.copyfile(os.path.join(REF, rel), dst)
    return os.path.basename(rel)


# ---------------------------------------------------------------------------
# G6: rrc_filter (read_file.py:5-36) for several parameter sets, incl. the
# |t| == sps/(4 alpha) special case (sps 4, alpha .25 -> t = +-4) and an
# asymmetric grid (odd span*sps).
params = [(4, 10, 0.35), (8, 10, 0.35), (2, 6, 0.25), (4, 10, 0.25), (3, 5, 0.5), (5, 7, 0.2),
          (100, 10, 0.35)]
arrs = {}
for i, (sps, span, al) in enumerate(params):
    arrs[f"h{i}"] = ref_rf.rrc_filter(sps, span, al)
    assert np.array_equal(arrs[f"h{i}"], ref_dm.rrc_filter(sps, span, al))
arrs["params"] = np.array(params, dtype=np.float64)
save("g_rrc_taps", **arrs)
h = ref_rf.rrc_filter(4, 10, 0.35)

# ---------------------------------------------------------------------------
# G1: config 1 -- qpsk_high_snr_sps_4_float32.pcm: load -> RRC 'same' -> /std ->
# gardner_timing_recovery(alpha .007) -> fll_func(Bn .03) (worker.py:299-304 flow),
# plus the gardner_real_file_2.py:9-29 variant (derotate 0.016428, alpha .04).
f1 = copy_data("qpsk_high_snr_sps_4_float32.pcm")
raw = np.fromfile("qpsk_high_snr_sps_4_float32.pcm", dtype=np.float32)
x = raw[::2] + 1j * raw[1::2]
assert x.dtype == np.complex64
yf = np.convolve(x, h, mode="same")
std = np.std(yf)
y = yf / std
rec, te, mu = ref_g2.gardner_timing_recovery(y, 4, alpha=0.007)
sl, cf, cfl, ef = ref_fll.fll_func(rec, 0.03, 1)
xd = x * np.exp(-1j * 2 * np.pi * 0.016428 * np.arange(x.size, dtype=np.float32))
yd = np.convolve(xd, h, mode="same")
yd = yd / np.std(yd)
rec2, te2, mu2 = ref_g2.gardner_timing_recovery(yd, 4, alpha=0.04)
sl2, cf2, cfl2, ef2 = ref_fll.fll_func(rec2, 0.03, 1)
save("g_config1", datafile=np.array(f1), h=h, conv=yf, std=np.float64(std), y=y,
     rec=rec, terr=te, mu=mu, fll_out=sl, fll_cf=np.float64(cf), fll_freq=cfl, fll_ef=ef,
     dec=quad(sl),
     yd=yd, rec2=rec2, terr2=te2, mu2=mu2, fll2_out=sl2, fll2_freq=cfl2, fll2_ef=ef2,
     fll2_cf=np.float64(cf2))

# ---------------------------------------------------------------------------
# G2: config 2 -- diff_method.py:165-307 flow on the sig_symb_x4_ncr capture.
f2 = copy_data("files/sig_symb_x4_ncr_77671952566_logon_id_1_tx_id_7616.pcm")
fp = copy_data("files/preambule_logon_id_2_tx_id_7616_float32.pcm")
from scipy.signal import firwin, lfilter  # noqa: E402

raw = np.fromfile("files/sig_symb_x4_ncr_77671952566_logon_id_1_tx_id_7616.pcm", dtype=np.float32)
sig = raw[::2] + 1j * raw[1::2]
pre = np.fromfile("files/preambule_logon_id_2_tx_id_7616_float32.pcm", dtype=np.float32)
pre = pre[::2] + 1j * pre[1::2]
sig = sig * np.exp(-1j * 2 * np.pi * 0.0105 * np.arange(len(sig)))
lpf = firwin(101, cutoff=0.14, window="hamming", fs=1.0)
sig = lfilter(lpf, 1.0, sig)
sig = sig / np.std(sig)
with contextlib.redirect_stdout(io.StringIO()):
    offset, aligned, conv_results, conv_max, phase_offset = ref_dm.find_preamble_offset(sig, pre, 4)
s1 = aligned[::4][:len(pre)]
phase_signal = s1 * np.conj(pre)
phases = np.angle(phase_signal)
f_m1 = np.mean(np.diff(np.unwrap(phases))) / (2 * np.pi * 4)
f_m2 = np.polyfit(np.arange(len(phases)), np.unwrap(phases), 1)[0] / (2 * np.pi * 4)
f_m3 = np.angle(np.mean(phase_signal[1:] * np.conj(phase_signal[:-1]))) / (2 * np.pi * 4)
out = dict(datafile=np.array(f2), offset=np.int64(offset), phase_offset=np.int64(phase_offset),
           aligned=aligned, f_rel=np.array([f_m1, f_m2, f_m3]))
s = aligned * np.exp(-1j * 2 * np.pi * f_m2 * np.arange(len(aligned)))
sfc = np.convolve(s, h, mode="same")
sf = sfc / np.std(sfc)
out["derot"] = s
out["conv"] = sfc
out["y"] = sf
a, b, c, d = ref_fll.fll_func(sf[0::4], 0.000001, 1)          # diff_method.py:304-307
out.update(fllA_out=a, fllA_cf=np.float64(b), fllA_freq=c, fllA_ef=d)
rec, te, mu = ref_g2.gardner_timing_recovery(sf, 4, alpha=0.007)
a, b, c, d = ref_fll.fll_func(rec, 0.03, 1)
out.update(rec=rec, terr=te, mu=mu, fllB_out=a, fllB_cf=np.float64(b), fllB_freq=c, fllB_ef=d,
           dec=quad(a))
save("g_config2", **out)

# ---------------------------------------------------------------------------
# G3: hw1/gardner.py script flow (:94-117): read_iq_int16 -> derotate -> gardner_ted(sps 2);
# plus read_file.gardner_ted with delta_mu 0.005 on a 12 000-sample prefix.
f3 = copy_data("iq_int16.pcm")
iq = ref_hw1g.read_iq_int16("iq_int16.pcm")
iq_raw = ref_hw1g.read_iq_int16("iq_int16.pcm", normalize=False)
n = np.arange(iq.size, dtype=np.float32)
bb = iq * np.exp(-1j * 2 * np.pi * (-0.00026) * n)
yt, et, mt = ref_hw1g.gardner_ted(bb, 2)
assert np.array_equal(yt, ref_hw1g.y_file)
yt2, et2, mt2 = ref_rf.gardner_ted(bb[:12000], 2, delta_mu=0.005)
# NOTE (numpy >= 2, NEP 50): bb is complex64 and gardner_ted's Python-float constants are
# "weak", so the run above silently computes in float32 (mu, the sample index and the
# interpolation all become float32 after the first iteration).  The same function on the
# complex128 view of the same samples is the float64 algorithm every other call site runs.
bb128 = bb.astype(np.complex128)
yd_, ed_, md_ = ref_hw1g.gardner_ted(bb128, 2)
yd2_, ed2_, md2_ = ref_rf.gardner_ted(bb128[:12000], 2, delta_mu=0.005)
save("g_hw1_gardner", datafile=np.array(f3), iq=iq, iq_raw=iq_raw, baseband=bb,
     y=yt, err=et, mu=mt, y2=yt2, err2=et2, mu2=mt2,
     y_f64=yd_, err_f64=ed_, mu_f64=md_, y2_f64=yd2_, err2_f64=ed2_, mu2_f64=md2_)

# ---------------------------------------------------------------------------
# G4: gardner2.py __main__ demo (:102-138): seed 1, sps 8, delay .095, alpha .07
np.random.seed(1)
Nsym, sps = 2000, 8
data = np.random.randint(0, 4, Nsym)
symbols = np.exp(1j * 2 * np.pi * data / 4)
samples = np.zeros(Nsym * sps, dtype=complex)
for i in range(Nsym):
    samples[i * sps] = symbols[i]
rrc8 = ref_rf.rrc_filter(sps, 10, 0.35)
samples = np.convolve(samples, rrc8, mode="same")
t = np.arange(len(samples))
shifted = np.interp(t, t - 0.095, samples)
rec, te, mu = ref_g2.gardner_timing_recovery(shifted, sps, alpha=0.07, mu_initial=0.0)
save("g_gardner2_demo", x=shifted, rec=rec, terr=te, mu=mu)

# ---------------------------------------------------------------------------
# G5: lagrange_resample (hw1/test_lagrange_resampling.py:116-168 scenario, seed 42)
np.random.seed(42)
with contextlib.redirect_stdout(io.StringIO()):
    base = ref_syn.generate_mpsk_signal(modulation_type="qpsk", Fs=100_000, Fc=1_000, Sps=100,
                                        num_symbols=200, use_pulse_shaping=True)
    rr = ref_syn.apply_rrc_filter(base["symbols_expanded"], Sps=100, alpha=0.35, filter_span=10,
                                  Fs=100_000)
clean = rr["signal"]
mis = ref_lag.apply_fractional_shift(clean, shift=0.35, half_len=6)
cor = ref_lag.apply_fractional_shift(mis, shift=-0.35, half_len=6)
rng = np.random.default_rng(7)
pos = rng.uniform(-3.0, len(clean) + 2.0, 2000)
pos[:8] = [0.0, 1.0, 5.0, len(clean) - 1.0, -0.5, len(clean) - 0.5, 17.25, 17.0 + 1e-13]
arb = ref_lag.lagrange_resample(clean, pos, half_len=4)
filter_delay = len(rr["filter"]) // 2
symbol_positions = filter_delay + np.arange(12, 200 - 12) * 100
save("g_lagrange", clean=clean, mis=mis, cor=cor, pos=pos, arb=arb,
     err_mis=np.float64(ref_lag.rms_error(clean[symbol_positions], mis[symbol_positions])),
     err_cor=np.float64(ref_lag.rms_error(clean[symbol_positions], cor[symbol_positions])))

# G10: RRC cascade (hw1/test_rrc_cascade.py:14-23,150-181 scenario, seed 42): the
# 'apply_filter' convention (full conv cropped by len(h)//2) with the 1002-tap
# symbol-time RRC of hw1/synthesys.py:263-291 (Sps 100).
np.random.seed(42)
with contextlib.redirect_stdout(io.StringIO()):
    sb = ref_syn.generate_mpsk_signal(modulation_type="qpsk", Fs=100000, Fc=1000, Sps=100,
                                      num_symbols=40, use_pulse_shaping=True)
    r1 = ref_syn.apply_rrc_filter(sb["symbols_expanded"], Sps=100, alpha=0.35, filter_span=10)
    r2 = ref_syn.apply_rrc_filter(r1["signal"], Sps=100, alpha=0.35, filter_span=10)
save("g_rrc_cascade", impulses=sb["symbols_expanded"], symbols=sb["symbols"], taps=r1["filter"],
     after1=r1["signal"], after2=r2["signal"])

# ---------------------------------------------------------------------------
# G7: seeded synthetic bursts (SURVEY 8d recipe; configs 3/4): worker.py:299-304 chain.
NB = 32
X = np.stack([synth_burst(b, h) for b in range(NB)])
assert X.dtype == np.complex64 and X.shape == (NB, 8192)
Y = np.empty((NB, 8192), np.complex128)
stds = np.empty(NB)
n_rec = np.empty(NB, np.int64)
REC = np.zeros((NB, 2048), np.complex128); TE = np.zeros((NB, 2048)); MU = np.zeros((NB, 2048))
SL = np.zeros((NB, 2048), np.complex128); FR = np.zeros((NB, 2048)); EF = np.zeros((NB, 2048))
DEC = np.zeros((NB, 2048), np.uint8)
# chain B: the same loops on the float32-rounded matched-filter output (the unit the
# device kernels consume): y32 = complex64(y) -> complex128 -> gardner -> fll
n_recB = np.empty(NB, np.int64)
RECB = np.zeros((NB, 2048), np.complex128); MUB = np.zeros((NB, 2048)); TEB = np.zeros((NB, 2048))
SLB = np.zeros((NB, 2048), np.complex128); FRB = np.zeros((NB, 2048)); EFB = np.zeros((NB, 2048))
DECB = np.zeros((NB, 2048), np.uint8)
for b in range(NB):
    yf = np.convolve(X[b], h, mode="same")
    stds[b] = np.std(yf)
    Y[b] = yf / stds[b]
    rec, te, mu = ref_g2.gardner_timing_recovery(Y[b], 4, alpha=0.007, mu_initial=0.0)
    sl, cf, cfl, ef = ref_fll.fll_func(rec, 0.03, 1)
    n = len(rec); n_rec[b] = n
    REC[b, :n] = rec; TE[b, :n - 1] = te; MU[b, :n - 1] = mu
    SL[b, :n] = sl; FR[b, :n] = cfl; EF[b, :n] = ef; DEC[b, :n] = quad(sl)
    y32 = Y[b].astype(np.complex64).astype(np.complex128)
    rec, te, mu = ref_g2.gardner_timing_recovery(y32, 4, alpha=0.007, mu_initial=0.0)
    sl, cf, cfl, ef = ref_fll.fll_func(rec, 0.03, 1)
    n = len(rec); n_recB[b] = n
    RECB[b, :n] = rec; TEB[b, :n - 1] = te; MUB[b, :n - 1] = mu
    SLB[b, :n] = sl; FRB[b, :n] = cfl; EFB[b, :n] = ef; DECB[b, :n] = quad(sl)
NF = 6   # full trajectories for the first NF bursts; decisions + final state for all
save("g_synth_bursts", x=X, std=stds, y=Y[:NF], n_rec=n_rec, rec=REC[:NF], terr=TE[:NF], mu=MU[:NF],
     fll_out=SL[:NF], fll_freq=FR[:NF], fll_ef=EF[:NF], dec=DEC,
     mu_last=MU[np.arange(NB), n_rec - 2], freq_last=FR[np.arange(NB), n_rec - 1],
     n_recB=n_recB, recB=RECB[:NF], terrB=TEB[:NF], muB=MUB[:NF], fllB_out=SLB[:NF],
     fllB_freq=FRB[:NF], fllB_ef=EFB[:NF], decB=DECB,
     muB_last=MUB[np.arange(NB), n_recB - 2], freqB_last=FRB[np.arange(NB), n_recB - 1])

# ---------------------------------------------------------------------------
# G9: edge cases of gardner_timing_recovery / gardner_ted / fll_func.
edge = {}
rng = np.random.default_rng(99)
z = (rng.standard_normal(64) + 1j * rng.standard_normal(64))
for L in (1, 3, 4, 7, 8, 9, 12, 13):
    r, e, m = ref_g2.gardner_timing_recovery(z[:L], 4, alpha=0.05)
    edge[f"g2_len{L}_rec"] = r; edge[f"g2_len{L}_mu"] = m
    r, e, m = ref_hw1g.gardner_ted(z[:L], 4) if L >= 1 else (None, None, None)
    edge[f"ted_len{L}_y"] = np.asarray(r, dtype=np.complex128); edge[f"ted_len{L}_mu"] = np.asarray(m)
# forced mu wraps (k slips both ways): strong gain on a noisy burst, long enough to wrap
zz = synth_burst(1000, h).astype(np.complex128)[:4096] * 3.0
import signal  # noqa: E402


class _Timeout(Exception):
    pass


def _alarm(*a):
    raise _Timeout()


signal.signal(signal.SIGALRM, _alarm)
# NOTE: the reference loop can run forever when mu keeps wrapping downwards (k -= sps
# at gardner2.py:68-70 undoes the k += sps at :98); such cases are recorded as
# "does not terminate" and left out of the vectors.
nonterm = []
for nm, al in (("pos", 0.9), ("neg", -0.9), ("big", 2.5), ("mid", 0.3), ("negmid", -0.3)):
    signal.alarm(20)
    try:
        r, e, m = ref_g2.gardner_timing_recovery(zz, 4, alpha=al, mu_initial=0.0)
        signal.alarm(0)
    except _Timeout:
        nonterm.append(nm)
        print("gardner2 wrap case", nm, "alpha", al, "does not terminate in 20 s -- skipped")
        continue
    edge[f"g2_wrap_{nm}_rec"] = r; edge[f"g2_wrap_{nm}_err"] = e; edge[f"g2_wrap_{nm}_mu"] = m
    edge[f"g2_wrap_{nm}_alpha"] = np.float64(al)
    kk = np.round((np.arange(1, len(m) + 1) * 0))  # placeholder to keep lint quiet
    print("wrap case", nm, "n", len(r), "mu range", float(np.min(m)), float(np.max(m)))
edge["g2_wrap_nonterminating"] = np.array(nonterm)
r, e, m = ref_g2.gardner_timing_recovery(zz, 4, alpha=0.007, mu_initial=3.7)
edge["g2_mu0_rec"] = r; edge["g2_mu0_mu"] = m
r, e, m = ref_rf.gardner_ted(zz, 4, delta_mu=0.8)       # clip at +-sps
edge["ted_clip_y"] = r; edge["ted_clip_err"] = e; edge["ted_clip_mu"] = m
edge["wrap_x"] = zz
# fll: exact zeros / pure-axis samples (np.sign(0) = 0 path), and an empty input
zf = np.array([0, 1, 1j, -1, -1j, 0.5 + 0.5j, 0, -0.3 + 0j, 0 - 0.2j, 1e-300 + 1j], dtype=np.complex128)
a, b, c, d = ref_fll.fll_func(zf, 50.0, 1)
edge["fll_axis_x"] = zf; edge["fll_axis_out"] = a; edge["fll_axis_freq"] = c; edge["fll_axis_ef"] = d
a, b, c, d = ref_fll.fll_func(np.zeros(0, complex), 0.03, 1)
edge["fll_empty_cf"] = np.float64(b)
a, b, c, d = ref_fll.fll_func(zz[::4], 0.5, 0.707)     # other loop constants
edge["fll_y707_out"] = a; edge["fll_y707_freq"] = c; edge["fll_y707_ef"] = d
save("g_edges", **edge)

# G8: read_iq_int16 edge cases (hw1/gardner.py:77-91) via temp files
import tempfile  # noqa: E402

cases = {"odd": np.array([100, -200, 300, 400, 7], np.int16),
         "zeros": np.zeros(8, np.int16),
         "minmax": np.array([-32768, 32767, 32767, -32768, 1, -1, 0, 5], np.int16),
         "rand": np.random.default_rng(5).integers(-32768, 32768, 4001).astype(np.int16)}
arrs = {}
with tempfile.TemporaryDirectory() as td:
    for k, v in cases.items():
        p = os.path.join(td, k + ".pcm")
        v.tofile(p)
        arrs[k + "_raw"] = v
        arrs[k + "_norm"] = ref_hw1g.read_iq_int16(p, True)
        arrs[k + "_nonorm"] = ref_hw1g.read_iq_int16(p, False)
    p = os.path.join(td, "short.pcm")
    np.array([5], np.int16).tofile(p)
    try:
        ref_hw1g.read_iq_int16(p)
        arrs["short_raises"] = np.array(False)
    except ValueError:
        arrs["short_raises"] = np.array(True)
save("g_read_iq", **arrs)
copy_data("preamb_symbols_float32.pcm")
print("done")

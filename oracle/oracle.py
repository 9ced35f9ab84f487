"""ctypes front end of the CPU oracle (oracle/dsp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- never by the product package dsp_amd/.
Function names and return tuples mirror the reference's Python functions so
the parity tests read like calls into the reference:

    rrc_filter                read_file.py:5-36
    convolve_same / np_std    worker.py:173-178
    gardner_timing_recovery   gardner2.py:6-100
    gardner_ted               hw1/gardner.py:22-74, read_file.py:53-104
    fll_func                  fll_func.py:3-29
    lagrange_resample         hw1/test_lagrange_resampling.py:34-72
    read_iq_int16             hw1/gardner.py:77-91
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "dsp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        i64, dbl, vp, ci = C.c_int64, C.c_double, C.c_void_p, C.c_int
        L.orc_sum.restype = dbl
        L.orc_sum.argtypes = [vp, i64]
        L.orc_rrc_taps.restype = i64
        L.orc_rrc_taps.argtypes = [ci, ci, dbl, vp, i64]
        L.orc_conv_same.restype = ci
        L.orc_conv_same.argtypes = [vp, i64, vp, i64, vp]
        L.orc_std_complex.restype = dbl
        L.orc_std_complex.argtypes = [vp, i64]
        L.orc_scale_complex.restype = None
        L.orc_scale_complex.argtypes = [vp, i64, dbl]
        L.orc_gardner2.restype = i64
        L.orc_gardner2.argtypes = [vp, i64, ci, dbl, dbl, vp, vp, vp, i64]
        L.orc_gardner_ted.restype = i64
        L.orc_gardner_ted.argtypes = [vp, i64, ci, dbl, vp, vp, vp, i64]
        L.orc_fll.restype = dbl
        L.orc_fll.argtypes = [vp, i64, dbl, dbl, vp, vp, vp]
        L.orc_fll_state.restype = dbl
        L.orc_fll_state.argtypes = [vp, i64, dbl, dbl, i64, vp, vp, vp, vp, vp]
        L.orc_quadrants.restype = None
        L.orc_quadrants.argtypes = [vp, i64, vp]
        L.orc_lagrange_resample.restype = ci
        L.orc_lagrange_resample.argtypes = [vp, i64, vp, i64, ci, vp]
        L.orc_read_iq_int16.restype = i64
        L.orc_read_iq_int16.argtypes = [vp, i64, ci, vp]
        L.orc_read_iq_f32.restype = i64
        L.orc_read_iq_f32.argtypes = [vp, i64, vp]
        L.orc_sync_chain_c64.restype = i64
        L.orc_sync_chain_c64.argtypes = [vp, i64, dbl, ci, dbl, dbl, dbl, dbl,
                                         vp, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c128(x):
    return np.ascontiguousarray(x, dtype=np.complex128)


# ---------------------------------------------------------------------------
def rrc_filter(sps, span, alpha):
    n = (span * sps) // 2 + 1 - ((-span * sps) // 2)
    h = np.zeros(max(n, 0), dtype=np.float64)
    got = lib().orc_rrc_taps(int(sps), int(span), float(alpha), _p(h), h.size)
    assert got == h.size, (got, h.size)
    return h


def convolve_same(x, h):
    x = _c128(x)
    h = np.ascontiguousarray(h, dtype=np.float64)
    y = np.empty_like(x)
    rc = lib().orc_conv_same(_p(x), x.size, _p(h), h.size, _p(y))
    if rc != 0:
        raise ValueError("orc_conv_same needs len(x) >= len(h) > 0")
    return y


def np_std(y):
    y = _c128(y)
    return lib().orc_std_complex(_p(y), y.size)


def filter_samples(x, h):
    """worker.py:173-178: convolve 'same' then / np.std."""
    y = convolve_same(x, h)
    s = np_std(y)
    lib().orc_scale_complex(_p(y), y.size, s)
    return y


def gardner_timing_recovery(signal, sps, alpha=0.007, mu_initial=0.0):
    x = _c128(signal)
    if x.size == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    cap = x.size // max(int(sps), 1) + 8
    for attempt in range(2):                # a loop that slips backwards can emit more than len/sps symbols;
        rec = np.empty(cap, np.complex128)  # one that never ends (gardner2.py can) still fails after the retry
        err = np.empty(cap, np.float64)
        mu = np.empty(cap, np.float64)
        n = lib().orc_gardner2(_p(x), x.size, int(sps), float(alpha), float(mu_initial),
                               _p(rec), _p(err), _p(mu), cap)
        if n != -2:
            break
        cap *= 4
    if n < 0:
        raise RuntimeError(f"orc_gardner2 failed ({n})")
    return rec[:n].copy(), err[:n - 1].copy(), mu[:n - 1].copy()


def gardner_ted(x, sps, delta_mu=0.002):
    x = _c128(x)
    cap = max(x.size // max(int(sps), 1) + 8, 8)
    y = np.empty(cap, np.complex128)
    err = np.empty(cap, np.float64)
    mu = np.empty(cap, np.float64)
    n = lib().orc_gardner_ted(_p(x), x.size, int(sps), float(delta_mu), _p(y), _p(err), _p(mu), cap)
    if n < 0:
        raise RuntimeError(f"orc_gardner_ted failed ({n})")
    return y[:n].copy(), err[:n].copy(), mu[:n].copy()


def fll_func(signal, Bn=0.001, y=1):
    x = _c128(signal)
    out = np.empty(x.size, np.complex128)
    freq = np.empty(x.size, np.float64)
    ef = np.empty(x.size, np.float64)
    cf = lib().orc_fll(_p(x), x.size, float(Bn), float(y), _p(out), _p(freq), _p(ef))
    return out, cf, freq, ef


def fll_state(signal, Bn, y, i0=0, state=None):
    """fll_func continued from (curr_freq, integrator) at absolute index i0."""
    x = _c128(signal)
    out = np.empty(x.size, np.complex128)
    freq = np.empty(x.size, np.float64)
    ef = np.empty(x.size, np.float64)
    st_in = np.asarray(state, np.float64) if state is not None else None
    st_out = np.zeros(2, np.float64)
    lib().orc_fll_state(_p(x), x.size, float(Bn), float(y), int(i0), _p(st_in), _p(st_out),
                        _p(out), _p(freq), _p(ef))
    return out, freq, ef, st_out


def quadrants(z):
    z = _c128(z)
    q = np.empty(z.size, np.uint8)
    lib().orc_quadrants(_p(z), z.size, _p(q))
    return q


def lagrange_resample(signal, positions, half_len=4):
    sig = np.asarray(signal)
    x = _c128(sig)
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    out = np.empty(pos.size, np.complex128)
    rc = lib().orc_lagrange_resample(_p(x), x.size, _p(pos), pos.size, int(half_len), _p(out))
    if rc != 0:
        raise ValueError("orc_lagrange_resample: bad arguments")
    return out.astype(sig.dtype) if np.iscomplexobj(sig) else out.real.astype(sig.dtype)


def read_iq_int16_words(raw, normalize=True):
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    out = np.empty(raw.size // 2, np.complex64)
    n = lib().orc_read_iq_int16(_p(raw), raw.size, int(bool(normalize)), _p(out))
    if n < 0:
        raise ValueError("need at least one I/Q pair")
    return out[:n]


def read_iq_int16(path, normalize=True):
    return read_iq_int16_words(np.fromfile(path, dtype=np.int16), normalize)


def read_iq_f32_words(raw):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    out = np.empty(raw.size // 2, np.complex64)
    lib().orc_read_iq_f32(_p(raw), raw.size, _p(out))
    return out


def sync_chain_c64(x, sps=4, alpha=0.007, mu0=0.0, Bn=0.03, y=1.0, scale=1.0, want="all"):
    """Gardner -> FLL -> quadrants on one burst of matched-filtered complex64."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    cap = x.size // sps + 8
    if want == "all":
        rec = np.empty(cap, np.complex128); err = np.empty(cap); mu = np.empty(cap)
        sig = np.empty(cap, np.complex128); freq = np.empty(cap); ef = np.empty(cap)
    else:
        rec = err = mu = sig = freq = ef = None
    dec = np.empty(cap, np.uint8)
    n = lib().orc_sync_chain_c64(_p(x), x.size, float(scale), int(sps), float(alpha), float(mu0),
                                 float(Bn), float(y), _p(rec), _p(err), _p(mu), _p(sig),
                                 _p(freq), _p(ef), _p(dec))
    if n < 0:
        raise RuntimeError(f"orc_sync_chain_c64 failed ({n})")
    if want == "all":
        return dict(n=n, rec=rec[:n], err=err[:n - 1], mu=mu[:n - 1], sig=sig[:n],
                    freq=freq[:n], ef=ef[:n], dec=dec[:n])
    return dict(n=n, dec=dec[:n])


# ---------------------------------------------------------------------------
# Segment-parallel mode (no reference counterpart: the definition of include/dsp_hip.h
# dsp_sync_capture_segmented restated with the sequential oracle loops above and numpy).
_TURN = np.array([1, 3, 0, 2], np.uint8)        # quadrant index after +90 degrees


def _positions(start, mu, n, sps):
    """Stream position of symbol rows 0..n-1 of a window: row 0 is the window's first sample
    (gardner2.py:59), row r >= 1 is k + mu of the iteration that produced it (gardner2.py:95)
    with k = sps*(r + slips) (gardner2.py:61,65-70,98; a slip shows as a jump of -+sps in the mu
    history), rounded to float32 as the device stores it."""
    p = np.empty(n)
    p[0] = start
    if n > 1:
        m = np.asarray(mu[:n - 1], dtype=np.float64)
        d = np.diff(m)
        step = np.zeros(n - 1, np.int64)
        step[1:][d < -0.5 * sps] = 1
        step[1:][d > 0.5 * sps] = -1
        slips = np.cumsum(step)
        r = np.arange(1, n)
        local = (sps * (r + slips).astype(np.float64) + m).astype(np.float32)
        p[1:] = start + local.astype(np.float64)
    return p


def acquire_phase(w, sps, n_sym=256):
    """Initial timing phase of a window (include/dsp_hip.h dsp_capture_args.acquire): the p in
    [0, sps) whose strobes w[p + k sps], k < n_sym, carry the most energy; float64 sums in k order,
    the first of equal maxima."""
    ns = min(n_sym, len(w) // sps)
    re = w.real.astype(np.float64)
    im = w.imag.astype(np.float64)
    best, arg = -1.0, 0
    for p in range(sps):
        acc = 0.0
        for k in range(ns):
            i = k * sps + p
            acc += float(re[i] * re[i] + im[i] * im[i])
        if acc > best:
            best, arg = acc, p
    return float(arg)


ACQ_FRAC_SYMBOLS = 512          # strobes per phase behind the fractional start (acquire = 2)
ACQ_FRAC_STEPS = 1024           # ... which is rounded to 1/1024 sample


def _cos_sin_turns(t):
    """cos, sin of t turns, exact at the quarter turns (what cospi / sinpi give on the device)."""
    t = t - np.floor(t)
    for k, v in ((0.0, (1.0, 0.0)), (0.25, (0.0, 1.0)), (0.5, (-1.0, 0.0)), (0.75, (0.0, -1.0))):
        if t == k:
            return v
    return float(np.cos(2.0 * np.pi * t)), float(np.sin(2.0 * np.pi * t))


def acquire_phase_frac(w, sps, n_sym=ACQ_FRAC_SYMBOLS):
    """Fractional initial timing phase of a window (dsp_capture_args.acquire = 2): the strobe energies
    E_p of acquire_phase are samples of A + B cos(2 pi (t - tau) / sps) -- the symbol-rate line of
    |x|^2 (Oerder-Meyr) -- so tau = atan2(sum E_p sin(2 pi p / sps), sum E_p cos(2 pi p / sps)) * sps
    / (2 pi), taken into [0, sps) and rounded to 1 / 1024 sample (the rounding makes the start the
    same number wherever atan2 and the trigonometric table differ in their last bits).  The strobes
    of a phase are added in `parts` contiguous parts of ceil(ns / parts) (4 parts up to sps 4, 2 up
    to 8, else 1; k order inside a part, then the parts first to last): the device gives each part
    a wave."""
    ns = min(n_sym, len(w) // sps)
    re = w.real.astype(np.float64)
    im = w.imag.astype(np.float64)
    parts = 4 if sps <= 4 else (2 if sps <= 8 else 1)
    plen = (ns + parts - 1) // parts
    c_sum, s_sum = 0.0, 0.0
    for p in range(sps):
        i = np.arange(ns) * sps + p
        en = (re[i] * re[i] + im[i] * im[i]).tolist()
        acc = None
        for q in range(parts):
            t = 0.0
            for v in en[q * plen:min(ns, (q + 1) * plen)]:          # float64 sums in k order
                t += v
            acc = t if acc is None else acc + t
        c, s = _cos_sin_turns(p / sps)
        c_sum += acc * c
        s_sum += acc * s
    tau = float(np.arctan2(s_sum, c_sum)) * sps / (2.0 * np.pi)
    if tau < 0.0:
        tau += sps
    q = float(np.floor(tau * ACQ_FRAC_STEPS + 0.5)) / ACQ_FRAC_STEPS
    return q - sps if q >= sps else q


def sync_capture_segmented(x, sps=4, alpha=0.007, mu0=0.0, Bn=0.003, y=1.0, seg_len=65536, warmup=2048,
                           vote=None, scale=1.0, acquire=2):
    """x: the whole capture, complex64, already matched-filtered; scale: factor applied to the
    samples in float64 (1/std of the whole filtered stream when the device filters); acquire: start
    every window at acquire_phase_frac(window) (2, the default), at acquire_phase(window) (1 / True:
    the sample grid) or at mu0 (0 / False)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n = x.size
    S = (n + seg_len - 1) // seg_len
    lead, tail = warmup * sps, 4 * sps + 8
    vote = min(256, warmup // 2) if vote is None else vote
    wins = []
    for s in range(S):
        st = max(0, s * seg_len - lead)
        en = n if s == S - 1 else min(n, (s + 1) * seg_len + tail)
        acq = int(acquire)
        m0 = acquire_phase_frac(x[st:en], sps) if acq == 2 else (acquire_phase(x[st:en], sps) if acq else mu0)
        o = sync_chain_c64(x[st:en], sps=sps, alpha=alpha, mu0=m0, Bn=Bn, y=y, scale=scale)
        pos = _positions(float(st), o["mu"], o["n"], sps)
        wins.append((o, pos))
    dec, mu, freq, info = [], [], [], np.zeros((S, 4), np.int32)
    p_last, j1_prev, rot_cum = None, None, 0
    for s, (o, pos) in enumerate(wins):
        seam_hi = np.inf if s == S - 1 else float((s + 1) * seg_len)
        keep = np.nonzero(pos < seam_hi)[0]
        j1 = int(keep[-1]) + 1 if keep.size else 0
        pl = pos[j1 - 1] if j1 else -np.inf
        j0, rot = 0, 0
        if s > 0:
            later = np.nonzero(pos[:j1] > p_last + 0.5 * sps)[0]
            j0 = int(later[0]) if later.size else j1
            V = min(vote, j0, j1_prev)
            prev = wins[s - 1][0]["dec"]
            best = -1
            for r4 in range(4):
                a = o["dec"][j0 - V:j0].copy()
                for _ in range(r4):
                    a = _TURN[a]
                agree = int(np.sum(a == prev[j1_prev - V:j1_prev]))
                if agree > best:
                    best, rot = agree, r4
        j0 = min(j0, j1)
        rot_cum = (rot_cum + rot) & 3
        d = o["dec"][j0:j1].copy()
        for _ in range(rot_cum):
            d = _TURN[d]
        dec.append(d)
        m_sym = np.concatenate(([0.0], o["mu"]))[:o["n"]]      # mu history row r-1 belongs to symbol r
        mu.append(m_sym[j0:j1].astype(np.float32))
        freq.append(o["freq"][j0:j1].astype(np.float32))
        info[s] = (j0, j1, rot_cum, o["n"])
        p_last, j1_prev = pl, j1
    dec = np.concatenate(dec) if dec else np.zeros(0, np.uint8)
    return dict(n=dec.size, dec=dec, mu=np.concatenate(mu), freq=np.concatenate(freq), seg_info=info)


def fll_class(x, detector="decision_directed", Kp=0.001, Ki=0.00005, freq_limit=0.05):
    """CPU restatement of the library's OWN definition of the class-FLL facade
    (dsp_amd/csrc/nco_kernels.hip).  Parity unpinned: the reference's fll.py is absent."""
    x = np.asarray(x, np.complex128)
    n = x.size
    y = np.zeros(n, complex); err = np.zeros(n); fr = np.zeros(n); ph = np.zeros(n)
    phase = freq = integ = 0.0
    prev = 0j
    for i in range(n):
        yi = x[i] * np.exp(-1j * phase)
        if detector == "decision_directed":
            d = np.sign(yi.real) + 1j * np.sign(yi.imag)
            e = float(np.angle(yi * np.conj(d)))
        else:
            z = yi * np.conj(prev)
            e = float(np.angle(z)) if detector == "atan2" else float(z.imag / (abs(yi) * abs(prev) + 1e-12))
        integ = min(max(integ + Ki * e, -freq_limit), freq_limit)
        freq = min(max(Kp * e + integ, -freq_limit), freq_limit)
        phase = phase + 2 * np.pi * freq
        if phase > np.pi:
            phase -= 2 * np.pi
        elif phase <= -np.pi:
            phase += 2 * np.pi
        prev = yi
        y[i] = yi; err[i] = e; fr[i] = freq; ph[i] = phase
    t0 = n - (n + 3) // 4
    return y, (float(np.mean(fr[t0:])) if n else 0.0), err, fr, ph


def find_preamble_offset(signal_iq, preamble_iq, sps):
    """diff_method.py:92-162 restated loop for loop (numpy; the per-window np.std and np.sum are
    the reference's own calls).  Returns (offset, signal_aligned, conv_results, conv_max, phase)."""
    signal_iq = np.asarray(signal_iq)
    preamble_iq = np.asarray(preamble_iq)
    d_pre = preamble_iq[1:] * np.conj(preamble_iq[:-1])                 # :118
    conv_results, conv_max = [], []
    for phase in range(sps):
        s = signal_iq[phase::sps]                                      # :124
        d = s[1:] * np.conj(s[:-1])                                    # :125
        conv = np.zeros(len(d), dtype=np.complex64)                    # :137
        for k in range(len(d) - len(d_pre) + 1):                       # :140
            cur = d[k:k + len(d_pre)]
            Y = np.std(cur)
            conv[k] = np.sum(cur * np.conj(d_pre)) / Y if Y > 0 else 0   # :143-148
        conv_results.append(conv)
        conv_max.append((np.argmax(np.abs(conv)), np.max(np.abs(conv))))   # :151
    values = [v for _, v in conv_max]
    idx = int(np.argmax(values))
    offset = conv_max[idx][0]
    return offset, signal_iq[offset * sps + idx:], conv_results, conv_max, idx


def unwrap(ph):
    """np.unwrap (period 2*pi, discont pi) restated as the loop it vectorises: the jump between
    neighbours is folded into [-pi, pi) (a fold of exactly -pi with a positive jump counts as +pi),
    jumps smaller than pi are left alone, and the corrections accumulate left to right."""
    ph = np.asarray(ph, dtype=np.float64)
    up = ph.copy()
    acc = 0.0
    for i in range(1, len(ph)):
        d = ph[i] - ph[i - 1]
        m = np.fmod(d + np.pi, 2 * np.pi)
        if m < 0:
            m += 2 * np.pi
        m -= np.pi
        if m == -np.pi and d > 0:
            m = np.pi
        acc += 0.0 if abs(d) < np.pi else m - d
        up[i] = ph[i] + acc
    return up


def preamble_cfo(signal_for_f_rel, preamble_iq, sps):
    """The three preamble-aided coarse frequency estimates of diff_method.py:267-292 (the first is
    also worker.py:288-294): phase of s*conj(preamble) -> (1) mean unwrapped phase step, (2) slope of
    the least-squares line through the unwrapped phases, (3) angle of the mean lag-1 product; each
    divided by 2*pi*sps (cycles per sample)."""
    s = np.asarray(signal_for_f_rel, dtype=np.complex128)[:len(preamble_iq)]
    p = s * np.conj(np.asarray(preamble_iq, dtype=np.complex128)[:len(s)])
    up = unwrap(np.arctan2(p.imag, p.real))
    n = np.arange(len(up), dtype=np.float64)
    m1 = float(np.mean(np.diff(up)))
    nc, uc = n - n.mean(), up - up.mean()
    m2 = float(np.sum(nc * uc) / np.sum(nc * nc))            # polyfit(n, up, 1)[0] in closed form
    q = np.mean(p[1:] * np.conj(p[:-1]))
    m3 = float(np.arctan2(q.imag, q.real))
    return np.array([m1, m2, m3]) / (2 * np.pi * sps)


def estimate_initial_frequency(signal, num_samples):
    """estimate_freq.py:4-58: z = s**4 (two squarings), R = mean z[i]*conj(z[i-1]), angle(R)/(8*pi);
    0.0 for fewer than two samples."""
    s = np.asarray(signal, dtype=np.complex128)
    n = min(int(num_samples), len(s))
    if n < 2:
        return 0.0
    z = s[:n] * s[:n]
    z = z * z
    r = np.sum(z[1:] * np.conj(z[:-1])) / (n - 1)
    return float(np.arctan2(r.imag, r.real) / (8 * np.pi))


def derotate(x, f_rel, offset=0):
    """x * exp(-1j*2*pi*f_rel*(arange(len(x)) + offset))  (worker.py:298, diff_method.py:296-298)."""
    x = np.asarray(x, dtype=np.complex128)
    th = (2 * np.pi * f_rel) * (np.arange(len(x)) + offset).astype(np.float64)
    return x * (np.cos(th) - 1j * np.sin(th))


def constellation_points(constellation):
    """The decision tables of worker.Processing.fll_func (worker.py:40,49,58-60); None for the
    sign slicers (bpsk :36, qpsk :38)."""
    if constellation == "8psk":
        return np.exp(1j * (np.pi / 4 * np.arange(8) - np.pi))
    if constellation == "16psk":
        return np.exp(1j * (np.pi / 8 * np.arange(16) - np.pi))
    if constellation == "16apsk":
        inner = 0.5 * np.exp(1j * (np.pi / 4 + np.arange(4) * np.pi / 2))
        outer = 1 * np.exp(1j * (np.pi / 12 + np.arange(12) * np.pi / 6))
        return np.concatenate([inner, outer])
    if constellation in ("bpsk", "qpsk"):
        return None
    raise ValueError(constellation)


def decide_symbol(sample, constellation, table=None):
    """worker.py:35-66 / :84-118 -> (decided point, index).  Index: table position (first of the
    nearest points wins, strict <); bpsk 0 / 1 / 2 for Re > 0 / < 0 / == 0; qpsk quadrant
    (Re<0) + 2 (Im<0) (the parity object of a7; zero components decide 0 there)."""
    if constellation == "bpsk":
        s = np.sign(sample.real)
        return complex(s, 0.0), (0 if s > 0 else (1 if s < 0 else 2))
    if constellation == "qpsk":
        d = (np.sign(sample.real) + 1j * np.sign(sample.imag)) / np.sqrt(2)
        return d, int(sample.real < 0) + 2 * int(sample.imag < 0)
    best, arg = np.inf, 0
    for j, d in enumerate(table):
        dist = np.abs(d - sample)
        if dist < best:
            best, arg = dist, j
    return table[arg], arg


def fll_func_constellation(signal, Bn=0.001, y=1, constellation="qpsk"):
    """worker.Processing.fll_func (worker.py:18-78) restated loop for loop (small inputs only).
    Returns (signal_list, curr_freq, curr_freq_list, ef_n_list, decision_index)."""
    signal = np.asarray(signal, dtype=np.complex128)
    table = constellation_points(constellation)
    kp = 4 * Bn / 1000 / (y + 1 / (4 * y))
    ki = 4 * (Bn / 1000 / (y + 1 / (4 * y))) ** 2
    n = len(signal)
    out = np.zeros(n, np.complex128); fr = np.zeros(n); ef = np.zeros(n); dec = np.zeros(n, np.uint8)
    curr_freq, integrator = 0.0, 0.0
    for i in range(n):
        phase = 2 * np.pi * curr_freq * i
        s = signal[i] * complex(np.cos(phase), -np.sin(phase))
        d, k = decide_symbol(s, constellation, table)
        z = s * np.conj(d)
        error = float(np.arctan2(z.imag, z.real))
        integrator += error * ki
        e = error * kp + integrator
        curr_freq += e
        out[i] = s; fr[i] = curr_freq; ef[i] = e; dec[i] = k
    return out, curr_freq, fr, ef, dec

"""Host-side mirror of the reference's Python functions, running on the MI355X.

Same names, positional order, defaults and return tuples as the reference, so a
caller switches with an import change:

    rrc_filter(sps, span, alpha)                                  read_file.py:5-36
    gardner_timing_recovery(signal, sps, alpha=0.007, mu_initial=0.0)   gardner2.py:6-100
    gardner_ted(x, sps, delta_mu=0.002)              hw1/gardner.py:22-74, read_file.py:53-104
    fll_func(signal, Bn=0.001, y=1)                               fll_func.py:3-29
    interpolate_sample(x, idx_float)                              hw1/gardner.py:8-19

The single-stream functions take any complex array, run it as complex128 (what
every reference call site passes) and return complex128 / float64 arrays.  The
``*_batch`` forms take ``[B, N]`` arrays of independent bursts (complex64 ->
float32 results, complex128 -> float64 results; loop arithmetic is float64 on
the device either way) and return padded ``[B, max_n]`` arrays plus ``n_out``.

All of these call libdsp_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import math

import numpy as np

from . import _lib
from ._lib import (DSP_C64, DSP_C128, DSP_I16, DSP_GARDNER2, DSP_GARDNER_TED, check,
                   default_context)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _as_bursts(x):
    """-> contiguous [B, N] complex64 or complex128, dtype code, real dtype."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("expected a [B, N] array of bursts")
    if x.dtype == np.complex64:
        return np.ascontiguousarray(x), DSP_C64, np.float32
    return np.ascontiguousarray(x, dtype=np.complex128), DSP_C128, np.float64


def _lens(lens, B, N):
    if lens is None:
        return None
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    if lens.shape != (B,):
        raise ValueError("lens must have shape [B]")
    if lens.min(initial=0) < 0 or lens.max(initial=0) > N:
        raise ValueError("lens out of range")
    return lens


# ---------------------------------------------------------------------------
# host helpers
def rrc_filter(sps, span, alpha):
    """Root-raised-cosine taps, unit energy (read_file.py:5-36)."""
    L = _lib.lib()
    n = L.dsp_rrc_taps(int(sps), int(span), float(alpha), None, 0)
    check(n)
    h = np.empty(n, dtype=np.float64)
    check(L.dsp_rrc_taps(int(sps), int(span), float(alpha), _vp(h), n))
    return h


def fll_gains(Bn, y=1):
    """(kp, ki) of fll_func (fll_func.py:5-6)."""
    kp, ki = C.c_double(), C.c_double()
    check(_lib.lib().dsp_fll_gains(float(Bn), float(y), C.byref(kp), C.byref(ki)))
    return kp.value, ki.value


def interpolate_sample(x, idx_float):
    """Linear interpolation at a fractional index (hw1/gardner.py:8-19)."""
    idx_low = int(np.floor(idx_float))
    idx_high = idx_low + 1
    if idx_high >= len(x):
        return x[idx_low]
    frac = idx_float - idx_low
    return x[idx_low] * (1 - frac) + x[idx_high] * frac


# ---------------------------------------------------------------------------
# batched forms
def gardner_batch(x, sps, alpha=0.007, mu_initial=0.0, *, variant=DSP_GARDNER2, lens=None,
                  interp_half_len=0, capacity=None, ctx=None):
    """Gardner timing recovery of B independent bursts.

    Returns (recovered [B, cap], timing_errors [B, cap], mu_history [B, cap], n_out [B]).
    variant DSP_GARDNER2: n_out = len(recovered) (errors/mu have n_out-1 entries);
    variant DSP_GARDNER_TED: alpha is delta_mu and all three have n_out entries.
    """
    ctx = ctx or default_context()
    x, dt, rdt = _as_bursts(x)
    B, N = x.shape
    sps = int(sps)
    if sps < 1:
        raise ValueError("sps must be >= 1")
    lens = _lens(lens, B, N)
    cap = int(capacity) if capacity else N // sps + 8
    L = _lib.lib()
    for attempt in range(2):
        rec = np.zeros((B, cap), dtype=x.dtype)
        err = np.zeros((B, cap), dtype=rdt)
        mu = np.zeros((B, cap), dtype=rdt)
        n_out = np.zeros(B, dtype=np.int32)
        rc = L.dsp_gardner(ctx.handle, _vp(x), dt, B, N, N, _vp(lens), sps, float(alpha),
                           float(mu_initial), int(variant), int(interp_half_len),
                           _vp(rec), _vp(err), _vp(mu), _vp(n_out), cap)
        if rc == _lib.DSP_E_CAPACITY and attempt == 0 and capacity is None:
            cap *= 4          # symbol slips backwards can lengthen the output
            continue
        check(rc)
        break
    return rec, err, mu, n_out


def fll_batch(x, Bn=0.001, y=1, *, lens=None, i0=None, state=None, want_state=False, ctx=None):
    """fll_func on B independent symbol streams [B, M].

    Returns (signal_list, curr_freq [B], curr_freq_list, ef_n_list, decisions);
    with want_state=True also the end state [B, 2] = (curr_freq, integrator).
    """
    ctx = ctx or default_context()
    x, dt, rdt = _as_bursts(x)
    B, M = x.shape
    lens = _lens(lens, B, M)
    out = np.zeros((B, max(M, 1)), dtype=x.dtype)
    freq = np.zeros((B, max(M, 1)), dtype=rdt)
    ef = np.zeros((B, max(M, 1)), dtype=rdt)
    dec = np.zeros((B, max(M, 1)), dtype=np.uint8)
    st_out = np.zeros((B, 2), dtype=np.float64)
    i0a = None if i0 is None else np.ascontiguousarray(np.broadcast_to(i0, (B,)), dtype=np.float64)
    sta = None if state is None else np.ascontiguousarray(state, dtype=np.float64).reshape(B, 2)
    if B and M:
        check(_lib.lib().dsp_fll(ctx.handle, _vp(x), dt, B, _vp(lens), M, M, float(Bn), float(y),
                                 _vp(i0a), _vp(sta), _vp(st_out), _vp(out), _vp(freq), _vp(ef),
                                 _vp(dec), max(M, 1)))
    elif sta is not None:
        st_out[:] = sta
    res = (out[:, :M], st_out[:, 0].copy(), freq[:, :M], ef[:, :M], dec[:, :M])
    return res + (st_out,) if want_state else res


def sync_chain_batch(x, sps=4, alpha=0.007, mu_initial=0.0, Bn=0.03, y=1, *, variant=DSP_GARDNER2,
                     lens=None, scale=None, interp_half_len=0, outputs="all", capacity=None, ctx=None):
    """Fused Gardner -> FLL -> hard decisions (worker.py:301-304) on matched-filtered bursts.

    outputs: "all" or an iterable out of {"rec","err","mu","sig","freq","ef","dec"}.
    Returns a dict with the requested [B, cap] arrays and "n_out" [B].
    """
    ctx = ctx or default_context()
    x, dt, rdt = _as_bursts(x)
    B, N = x.shape
    sps = int(sps)
    lens = _lens(lens, B, N)
    names = ("rec", "err", "mu", "sig", "freq", "ef", "dec") if outputs == "all" else tuple(outputs)
    cap = int(capacity) if capacity else N // sps + 8
    sc = None if scale is None else np.ascontiguousarray(np.broadcast_to(scale, (B,)), dtype=np.float64)
    L = _lib.lib()
    for attempt in range(2):
        bufs = {}
        for nm in names:
            dtp = x.dtype if nm in ("rec", "sig") else (np.uint8 if nm == "dec" else rdt)
            bufs[nm] = np.zeros((B, cap), dtype=dtp)
        n_out = np.zeros(B, dtype=np.int32)
        g = lambda nm: _vp(bufs.get(nm))  # noqa: E731
        rc = L.dsp_sync_chain(ctx.handle, _vp(x), dt, B, N, N, _vp(lens), _vp(sc), sps, float(alpha),
                              float(mu_initial), int(variant), int(interp_half_len), float(Bn),
                              float(y), g("rec"), g("err"), g("mu"), g("sig"), g("freq"), g("ef"),
                              g("dec"), _vp(n_out), cap)
        if rc == _lib.DSP_E_CAPACITY and attempt == 0 and capacity is None:
            cap *= 4
            continue
        check(rc)
        break
    bufs["n_out"] = n_out
    return bufs


# ---------------------------------------------------------------------------
# the reference's single-stream signatures
def gardner_timing_recovery(signal, sps, alpha=0.007, mu_initial=0.0):
    """Drop-in for gardner2.gardner_timing_recovery (gardner2.py:6-100)."""
    x = np.asarray(signal)
    if x.ndim != 1:
        raise ValueError("signal must be one-dimensional")
    if x.size == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")   # gardner2.py:59
    x = x.astype(np.complex128, copy=False)
    rec, err, mu, n = gardner_batch(x[None, :], sps, alpha, mu_initial, variant=DSP_GARDNER2)
    n = int(n[0])
    return rec[0, :n].copy(), err[0, :n - 1].copy(), mu[0, :n - 1].copy()


def gardner_ted(x, sps, delta_mu=0.002):
    """Drop-in for gardner_ted (hw1/gardner.py:22-74; delta_mu as in read_file.py:53)."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("x must be one-dimensional")
    if x.size == 0:
        return np.array([]), np.array([]), np.array([])
    x = x.astype(np.complex128, copy=False)
    y, err, mu, n = gardner_batch(x[None, :], sps, delta_mu, 0.0, variant=DSP_GARDNER_TED)
    n = int(n[0])
    return y[0, :n].copy(), err[0, :n].copy(), mu[0, :n].copy()


def fll_func(signal, Bn=0.001, y=1):
    """Drop-in for fll_func.fll_func (fll_func.py:3-29)."""
    x = np.asarray(signal)
    if x.ndim != 1:
        raise ValueError("signal must be one-dimensional")
    if x.size == 0:
        return np.zeros(0, dtype=complex), 0, np.zeros(0, dtype=float), np.zeros(0, dtype=float)
    x = x.astype(np.complex128, copy=False)
    out, cf, freq, ef, _ = fll_batch(x[None, :], Bn, y)
    return out[0].copy(), float(cf[0]), freq[0].copy(), ef[0].copy()


# ---------------------------------------------------------------------------
# loaders and matched filter
def read_iq_int16_words(raw, normalize=True, *, ctx=None):
    """read_iq_int16 on an in-memory int16 word buffer (hw1/gardner.py:79-91)."""
    ctx = ctx or default_context()
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    if raw.size < 2:
        raise ValueError("need at least one I/Q pair")                       # hw1/gardner.py:80-81
    out = np.empty(raw.size // 2, dtype=np.complex64)
    n = C.c_int64()
    check(_lib.lib().dsp_load_iq_i16(ctx.handle, _vp(raw), raw.size, int(bool(normalize)), _vp(out),
                                     C.byref(n)))
    return out[:n.value]


def read_iq_int16(path, normalize=True):
    """Drop-in for read_iq_int16(path, normalize=True) (hw1/gardner.py:77-91)."""
    return read_iq_int16_words(np.fromfile(path, dtype=np.int16), normalize)


def read_iq_float32(path):
    """The float32 loader idiom s[::2] + 1j*s[1::2] (read_file.py:108-109) -> complex64."""
    raw = np.ascontiguousarray(np.fromfile(path, dtype=np.float32))
    out = np.empty(raw.size // 2, dtype=np.complex64)
    n = C.c_int64()
    check(_lib.lib().dsp_load_iq_f32(None, _vp(raw), raw.size, _vp(out), C.byref(n)))
    return out[:n.value]


def fir_batch(x, taps, *, normalize=True, start=-1, lens=None, want_std=False, ctx=None):
    """y = np.convolve(x[b], taps, 'same') [/ np.std(y)] for every burst (worker.py:173-178).

    start=-1 is mode='same'; start=len(taps)//2 is apply_filter of hw1/test_rrc_cascade.py:103-108.
    """
    ctx = ctx or default_context()
    x, dt, rdt = _as_bursts(x)
    B, N = x.shape
    h = np.ascontiguousarray(taps, dtype=np.float64)
    if h.ndim != 1 or h.size < 1:
        raise ValueError("taps must be a non-empty vector")
    if h.size > N:
        raise ValueError("len(taps) > len(x): np.convolve would swap its operands")
    lens = _lens(lens, B, N)
    y = np.zeros_like(x)
    std = np.zeros(B, dtype=np.float64)
    check(_lib.lib().dsp_fir_same_std(ctx.handle, _vp(x), dt, B, N, N, _vp(lens), _vp(h), h.size,
                                      int(start), int(bool(normalize)), _vp(y), N, _vp(std)))
    return (y, std) if want_std else y


def filter_samples(samples, sps=4, span=10, alpha=0.35):
    """Drop-in for Processing.filter_samples (worker.py:173-178): RRC 'same' then / np.std."""
    x = np.asarray(samples)
    if x.ndim != 1:
        raise ValueError("samples must be one-dimensional")
    h = rrc_filter(sps, span, alpha)
    x = x.astype(np.complex128, copy=False)
    return fir_batch(x[None, :], h, normalize=True)[0]


def rx_chain_batch(x, taps, sps=4, alpha=0.007, mu_initial=0.0, Bn=0.03, y=1, *, variant=DSP_GARDNER2,
                   lens=None, normalize_peak=True, normalize_std=True, interp_half_len=0,
                   outputs="all", capacity=None, ctx=None):
    """Raw bursts -> RRC 'same' -> /std -> Gardner -> FLL -> decisions (worker.py:299-304).

    x: [B, N] int16 pairs ([B, N, 2] or [B, 2N] int16), complex64 or complex128.
    Returns a dict of [B, cap] arrays plus "n_out" and "std".
    """
    ctx = ctx or default_context()
    x = np.asarray(x)
    if x.dtype == np.int16:
        if x.ndim == 3 and x.shape[2] == 2:
            x = x.reshape(x.shape[0], -1)
        if x.ndim != 2 or x.shape[1] % 2:
            raise ValueError("int16 input must be [B, 2N] interleaved I,Q")
        x = np.ascontiguousarray(x)
        B, N = x.shape[0], x.shape[1] // 2
        kind, cdt, rdt = DSP_I16, np.complex64, np.float32
    else:
        x, kind, rdt = _as_bursts(x)
        B, N = x.shape
        cdt = x.dtype
    h = np.ascontiguousarray(taps, dtype=np.float64)
    sps = int(sps)
    lens = _lens(lens, B, N)
    names = ("rec", "err", "mu", "sig", "freq", "ef", "dec") if outputs == "all" else tuple(outputs)
    cap = int(capacity) if capacity else N // sps + 8
    L = _lib.lib()
    for attempt in range(2):
        bufs = {}
        for nm in names:
            dtp = cdt if nm in ("rec", "sig") else (np.uint8 if nm == "dec" else rdt)
            bufs[nm] = np.zeros((B, cap), dtype=dtp)
        n_out = np.zeros(B, dtype=np.int32)
        std = np.zeros(B, dtype=np.float64)
        g = lambda nm: _vp(bufs.get(nm))  # noqa: E731
        rc = L.dsp_rx_chain(ctx.handle, _vp(x), kind, B, N, N, _vp(lens), int(bool(normalize_peak)),
                            _vp(h), h.size, int(bool(normalize_std)), sps, float(alpha),
                            float(mu_initial), int(variant), int(interp_half_len), float(Bn), float(y),
                            g("rec"), g("err"), g("mu"), g("sig"), g("freq"), g("ef"), g("dec"),
                            _vp(n_out), cap, _vp(std))
        if rc == _lib.DSP_E_CAPACITY and attempt == 0 and capacity is None:
            cap *= 4
            continue
        check(rc)
        break
    bufs["n_out"] = n_out
    bufs["std"] = std
    return bufs


def auto_seg_len(n, sps, warmup):
    """Segment length (samples) for a capture of n samples.  The recurrence is latency-bound (its
    time follows the length of ONE window, about 0.7 us per symbol) until some 512 tiles of 64
    windows are in flight, so: aim at 512 tiles, at most twice the warm-up long (8192 symbols
    without one); a capture too short for that at one warm-up per segment keeps the segment at the
    warm-up length (a window's warm-up rows are copied from its predecessor only while they fit
    inside it), and a still shorter one goes down to a quarter of it, aiming at 128 tiles and
    paying the filter for the overlap.  DESIGN.md 3.5 has the sweep behind these numbers."""
    sps = int(sps)
    lead = int(warmup) * sps
    # rounded UP to a multiple of sps and of four samples: at most 512 tiles, all of them full (the last
    # segment is the short one) -- whole tiles of segments are filtered once into segment tiles, and one
    # tile more than 512 would be a third workgroup on some CUs
    q = sps * 4 // math.gcd(sps, 4)
    seg = -(-int(n) // (512 * 64 * q)) * q
    if seg >= lead:
        return int(max(min(seg, max(2 * lead, 8192 * sps)), 1024 * sps))
    seg = (int(n) // (128 * 64)) // sps * sps
    return int(min(max(seg, lead // 4 // sps * sps, 256 * sps), max(lead, 256 * sps)))


def sync_capture_segmented(x, sps=4, alpha=0.007, mu_initial=0.0, Bn=0.003, y=1, *, taps=None,
                           seg_len=None, warmup=2048, normalize=True, want=("dec", "mu", "freq"),
                           acquire=2, ctx=None):
    """Segment-parallel Gardner + FLL + decisions over ONE long capture (no reference
    counterpart; north-star (c), include/dsp_hip.h dsp_sync_capture_segmented).

    x: 1-D complex64 samples, or 1-D int16 interleaved I,Q words (read_iq_int16 semantics,
    hw1/gardner.py:77-91).  taps: matched filter applied as np.convolve(x, taps, 'same') /
    np.std over the whole capture (worker.py:173-178), None when x is already filtered.
    Returns a dict: dec uint8[n] quadrant indices, mu / freq float32[n] if wanted,
    seg_info int32[S, 4] = (first kept row, end row, rotation, rows produced) per segment;
    with freq also seam_dfreq float64[S-1] = |freq just after a seam - freq just before it|.
    freq is a PER-WINDOW quantity: fll_func's index restarts at every window start, so across a seam
    the estimate jumps by what two loops that saw different pasts differ by -- a few standard
    deviations of the loop's own estimate noise after a 4096-symbol warm-up at Bn = 0.003, more when
    the warm-up is shorter than the loop's settling time.  Decisions and mu carry over.
    seg_len: samples per segment, a multiple of sps (None: auto_seg_len).
    acquire: 2 (default): every window starts its timing loop at the fractional phase of the
    symbol-rate line of |x|^2 over its first 512 symbols, within some 0.01 sample rms of the lock point
    (a 2048-symbol warm-up then leaves < 3e-4 symbol against one sequential loop); 1 / True: at the
    sampling phase whose strobes carry the most energy over its first 256 symbols (up to half a sample
    off: 4096 warm-up symbols for the same); 0 / False: at mu_initial.
    """
    ctx = ctx or default_context()
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("one capture = one 1-D array")
    if x.dtype == np.int16:
        n = x.size // 2
        x = np.ascontiguousarray(x[:2 * n])
        kind = DSP_I16
    else:
        x = np.ascontiguousarray(x, dtype=np.complex64)
        n = x.size
        kind = DSP_C64
    if n == 0:
        raise IndexError("empty capture")                     # gardner2.py:59
    if seg_len is None:
        seg_len = auto_seg_len(n, sps, warmup)
    S = int(_lib.lib().dsp_capture_segments(n, int(seg_len)))
    cap = n // sps + 8 * max(S, 1) + 64
    dec = np.zeros(cap, np.uint8)
    mu = np.zeros(cap, np.float32) if "mu" in want else None
    freq = np.zeros(cap, np.float32) if "freq" in want else None
    info = np.zeros((max(S, 1), 4), np.int32)
    total = C.c_int64(0)
    a = _lib.CaptureArgs()
    a.x = x.ctypes.data; a.in_kind = kind; a.normalize = 1 if normalize else 0; a.n_samples = n
    h = None
    if taps is not None:
        h = np.ascontiguousarray(taps, dtype=np.float64)
        a.h = h.ctypes.data; a.n_taps = h.size
    a.sps = int(sps); a.alpha = float(alpha); a.mu0 = float(mu_initial); a.Bn = float(Bn); a.y = float(y)
    a.seg_len = int(seg_len); a.warmup = int(warmup); a.acquire = int(acquire)
    a.dec = dec.ctypes.data
    a.mu = mu.ctypes.data if mu is not None else None
    a.freq = freq.ctypes.data if freq is not None else None
    a.capacity = cap
    a.n_symbols = C.addressof(total)
    a.seg_info = info.ctypes.data
    check(_lib.lib().dsp_sync_capture_segmented(ctx.handle, C.byref(a)))
    nn = int(total.value)
    out = {"n": nn, "dec": dec[:nn], "seg_info": info[:S]}
    if mu is not None:
        out["mu"] = mu[:nn]
    if freq is not None:
        out["freq"] = freq[:nn]
        out["seam_dfreq"] = seam_frequency_steps(out["freq"], out["seg_info"])
    return out


def seam_frequency_steps(freq, seg_info):
    """|freq[first symbol of a share] - freq[last symbol of the share before it]| for every seam of a
    stitched capture (shares = seg_info[:, 1] - seg_info[:, 0] symbols each, empty ones skipped)."""
    share = (seg_info[:, 1] - seg_info[:, 0]).astype(np.int64)
    at = np.cumsum(share)[:-1]
    at = np.unique(at[(at > 0) & (at < len(freq))])
    f = np.asarray(freq, dtype=np.float64)
    return np.abs(f[at] - f[at - 1])


def lagrange_resample(signal, positions, half_len=4, *, ctx=None):
    """hw1/test_lagrange_resampling.py:34-72: resample `signal` at fractional `positions` with a
    (2*half_len+1)-point Lagrange interpolator; returns the dtype of `signal`."""
    ctx = ctx or default_context()
    sig = np.asarray(signal)
    pos = np.ascontiguousarray(positions, dtype=np.float64).ravel()
    if sig.ndim != 1 or sig.size == 0:
        raise ValueError("signal must be a non-empty 1-D array")
    cplx = np.iscomplexobj(sig)
    if sig.dtype == np.complex64:
        x, dt = np.ascontiguousarray(sig), DSP_C64
    else:
        x, dt = np.ascontiguousarray(sig, dtype=np.complex128), DSP_C128
    out = np.empty(pos.size, np.complex128)
    check(_lib.lib().dsp_lagrange_resample(ctx.handle, _vp(x), dt, x.size, _vp(pos), pos.size, int(half_len),
                                           _vp(out)))
    if cplx:
        return out.astype(sig.dtype)
    return out.real.astype(sig.dtype if sig.dtype.kind == "f" else np.float64)


def apply_fractional_shift(signal, shift, half_len=4, *, ctx=None):
    """hw1/test_lagrange_resampling.py:75-79."""
    positions = np.arange(len(signal), dtype=float) + shift
    return lagrange_resample(signal, positions, half_len, ctx=ctx)


class FrequencyLockedLoop:
    """API-shaped facade for the reference's ``fll.FrequencyLockedLoop`` (constructed at
    fll_real_signal.py:66-71, f_rel.py:123-129; ``process_signal`` at :74 / :132; ``history`` keys
    at fll_real_signal.py:107-125).  **Parity unpinned**: the class' source is absent from the
    reference; the loop is this library's definition (include/dsp_hip.h dsp_fll_class)."""

    _DET = {"cross_product": 0, "atan2": 1, "decision_directed": 2}

    def __init__(self, detector_method="decision_directed", Kp=0.001, Ki=0.00005, freq_limit=0.05, *, ctx=None):
        if detector_method not in self._DET:
            raise ValueError(f"unknown detector_method {detector_method!r}")
        self.detector_method, self.Kp, self.Ki, self.freq_limit = detector_method, Kp, Ki, freq_limit
        self.history = {"errors": [], "freq_estimates": [], "phases": []}
        self._ctx = ctx

    def process_signal(self, signal):
        ctx = self._ctx or default_context()
        x, dt, rdt = _as_bursts(np.asarray(signal).reshape(1, -1))
        N = x.shape[1]
        out = np.zeros((1, max(N, 1)), x.dtype)
        e = np.zeros((1, max(N, 1)), rdt); f = np.zeros_like(e); p = np.zeros_like(e)
        fe = np.zeros(1)
        if N:
            check(_lib.lib().dsp_fll_class(ctx.handle, _vp(x), dt, 1, N, N, None, self._DET[self.detector_method],
                                           float(self.Kp), float(self.Ki), float(self.freq_limit),
                                           _vp(out), _vp(e), _vp(f), _vp(p), max(N, 1), _vp(fe)))
        self.history = {"errors": e[0, :N].copy(), "freq_estimates": f[0, :N].copy(), "phases": p[0, :N].copy()}
        return out[0, :N].copy(), float(fe[0])


def diff_corr(signal_iq, preamble_iq, sps, *, ctx=None):
    """The sliding normalised differential correlation of find_preamble_offset
    (diff_method.py:118-148): a list of sps complex64 arrays, conv_results[phase] of length
    len(signal_iq[phase::sps]) - 1."""
    ctx = ctx or default_context()
    x = np.asarray(signal_iq)
    if x.dtype == np.complex64:
        x, dt = np.ascontiguousarray(x), DSP_C64
    else:
        x, dt = np.ascontiguousarray(x, dtype=np.complex128), DSP_C128
    pre = np.ascontiguousarray(preamble_iq, dtype=np.complex128)
    if pre.size < 2:
        raise ValueError("the preamble needs at least two symbols")
    sps = int(sps)
    lens = [max(len(range(p, x.size, sps)) - 1, 0) for p in range(sps)]
    ks = max(max(lens), 1)
    conv = np.zeros((sps, ks), np.complex64)
    if x.size:
        check(_lib.lib().dsp_diff_corr(ctx.handle, _vp(x), dt, x.size, _vp(pre), pre.size, sps, _vp(conv), ks))
    return [conv[p, :lens[p]].copy() for p in range(sps)]


def find_preamble_offset(signal_iq, preamble_iq, sps, *, ctx=None):
    """Same contract as diff_method.find_preamble_offset (diff_method.py:92-162):
    (offset in symbols, signal from the preamble on, per-phase correlations, per-phase
    (arg-max, peak height), winning phase).  The correlations come from the device
    (diff_corr); picking the peak is a handful of host operations."""
    sig = np.asarray(signal_iq)
    corr = diff_corr(sig, preamble_iq, sps, ctx=ctx)
    conv_max = []
    for c in corr:
        mag = np.abs(c)
        k = np.argmax(mag)                       # first maximum, as np.argmax in the reference
        conv_max.append((k, mag[k]))
    best_phase = int(np.argmax([height for _, height in conv_max]))     # ties: the lowest phase
    offset = conv_max[best_phase][0]
    return offset, sig[offset * sps + best_phase:], corr, conv_max, best_phase


def find_optimal_sampling_phase(signal_filtered, preamble_symbols, sps, preamble_length=None):
    """Same contract as diff_method.find_optimal_sampling_phase (diff_method.py:41-89): for each
    of the sps decimation phases, |sum(strobes * conj(preamble))| over the first preamble-length
    strobes (0 where the signal is too short); returns (best phase, the sps magnitudes).
    sps sums of len(preamble) products: host numpy, nothing worth a launch."""
    ref = np.asarray(preamble_symbols)
    n_ref = len(ref) if preamble_length is None else min(preamble_length, len(ref))
    ref_conj = np.conj(ref[:n_ref])
    sig = np.asarray(signal_filtered)
    score = np.zeros(sps)
    for p in range(sps):
        strobes = sig[p::sps][:n_ref]
        if len(strobes) == n_ref:
            score[p] = np.abs(np.sum(strobes * ref_conj))
    return np.argmax(score), score


# ---------------------------------------------------------------------------
# coarse frequency estimators and the pre-derotation
def _as_signal(x):
    x = np.asarray(x)
    if x.dtype == np.complex64:
        return np.ascontiguousarray(x), DSP_C64
    return np.ascontiguousarray(x, dtype=np.complex128), DSP_C128


def preamble_cfo(signal_iq, preamble_iq, sps, offsets=None, *, ctx=None):
    """The preamble-aided coarse frequency estimates of diff_method.py:267-292 (method 1 is also
    worker.py:288-294): with s = signal_iq[offset::sps][:len(preamble)], p = s * conj(preamble) and
    phases = np.unwrap(np.angle(p)), returns f_rel by (1) mean phase step, (2) least-squares slope,
    (3) angle of the mean lag-1 product, each / (2 pi sps): float64[3], or [B, 3] for a sequence of
    sample offsets into one signal (one burst each)."""
    ctx = ctx or default_context()
    x, dt = _as_signal(signal_iq)
    pre = np.ascontiguousarray(preamble_iq, dtype=np.complex128)
    if pre.size < 2:
        raise ValueError("the preamble needs at least two symbols")
    single = offsets is None
    off = np.zeros(1, np.int64) if single else np.ascontiguousarray(offsets, dtype=np.int64).ravel()
    out = np.zeros((off.size, 3), np.float64)
    check(_lib.lib().dsp_cfo_preamble(ctx.handle, _vp(x), dt, x.size, _vp(off), off.size, int(sps), _vp(pre),
                                     pre.size, _vp(out)))
    return out[0] if single else out


def estimate_initial_frequency_batch(signals, num_samples, lens=None, *, ctx=None):
    """estimate_initial_frequency for every row of a [B, N] block in one call -> float64[B]."""
    ctx = ctx or default_context()
    x = np.asarray(signals)
    x = np.ascontiguousarray(x) if x.dtype == np.complex64 else np.ascontiguousarray(x, dtype=np.complex128)
    if x.ndim != 2:
        raise ValueError("signals must be [B, N]")
    B, N = x.shape
    out = np.zeros(B, np.float64)
    check(_lib.lib().dsp_cfo_fourth_power(ctx.handle, _vp(x), DSP_C64 if x.dtype == np.complex64 else DSP_C128,
                                         B, N, N, _vp(_lens(lens, B, N)),
                                         int(num_samples), _vp(out)))
    return out


def estimate_initial_frequency(signal, num_samples, *, ctx=None):
    """estimate_freq.estimate_initial_frequency (estimate_freq.py:4-58): fourth-power lag-1
    estimate over the first num_samples symbols, cycles per symbol; 0.0 for fewer than two."""
    x = np.asarray(signal).reshape(1, -1)
    if x.shape[1] == 0:
        return 0.0
    return float(estimate_initial_frequency_batch(x, num_samples, ctx=ctx)[0])


def derotate(signal, f_rel, offset=0, *, ctx=None):
    """signal * np.exp(-1j * 2 * np.pi * f_rel * (np.arange(len(signal)) + offset))
    (diff_method.py:296-298, worker.py:298) -> complex128."""
    ctx = ctx or default_context()
    x, dt = _as_signal(signal)
    out = np.zeros(x.size, np.complex128)
    if x.size:
        f = np.array([f_rel], np.float64)
        i0 = np.array([offset], np.int64)
        check(_lib.lib().dsp_derotate(ctx.handle, _vp(x), dt, x.size, None, 1, x.size, _vp(f), _vp(i0), _vp(out),
                                     x.size))
    return out


def rx_bursts(signal_iq, offsets, n_pkt, preamble_iq, taps, sps=4, alpha=0.007, mu_initial=0.0, Bn=0.03, y=1, *,
              cfo_method=1, f_rel=None, variant=DSP_GARDNER2, normalize_std=True, interp_half_len=0,
              outputs="all", capacity=None, ctx=None):
    """The per-burst body of worker.Processing.process (worker.py:284-304) for every burst of one
    signal in one device call: coarse CFO from the preamble (cfo_method 1..3, or f_rel given), cut
    of n_pkt samples at each offset, derotation with the absolute sample index, RRC + /np.std,
    Gardner, FLL.  Returns a dict of [B, cap] complex128 / float64 arrays + "n_out", "std", "f_rel"."""
    ctx = ctx or default_context()
    x, dt = _as_signal(signal_iq)
    off = np.ascontiguousarray(offsets, dtype=np.int64).ravel()
    B = off.size
    pre = np.ascontiguousarray(preamble_iq, dtype=np.complex128) if preamble_iq is not None else None
    h = np.ascontiguousarray(taps, dtype=np.float64)
    fin = None
    if f_rel is not None:
        fin = np.ascontiguousarray(np.broadcast_to(np.asarray(f_rel, np.float64), (B,)))
        cfo_method = 0
    names = ("rec", "err", "mu", "sig", "freq", "ef", "dec") if outputs == "all" else tuple(outputs)
    n_pkt, sps = int(n_pkt), int(sps)
    cap = int(capacity) if capacity else n_pkt // sps + 8
    L = _lib.lib()
    for attempt in range(2):
        bufs = {}
        for nm in names:
            dtp = np.complex128 if nm in ("rec", "sig") else (np.uint8 if nm == "dec" else np.float64)
            bufs[nm] = np.zeros((B, cap), dtype=dtp)
        n_out = np.zeros(B, dtype=np.int32)
        std = np.zeros(B, dtype=np.float64)
        fout = np.zeros(B, dtype=np.float64)
        g = lambda nm: _vp(bufs.get(nm))  # noqa: E731
        rc = L.dsp_rx_bursts(ctx.handle, _vp(x), dt, x.size, _vp(off), B, n_pkt, _vp(pre),
                             pre.size if pre is not None else 0, int(cfo_method), _vp(fin), _vp(h), h.size,
                             int(bool(normalize_std)), sps, float(alpha), float(mu_initial), int(variant),
                             int(interp_half_len), float(Bn), float(y),
                             g("rec"), g("err"), g("mu"), g("sig"), g("freq"), g("ef"), g("dec"),
                             _vp(n_out), cap, _vp(fout), _vp(std))
        if rc == _lib.DSP_E_CAPACITY and attempt == 0 and capacity is None:
            cap *= 4
            continue
        check(rc)
        break
    bufs.update(n_out=n_out, std=std, f_rel=fout)
    return bufs


# ---------------------------------------------------------------------------
# the constellation switch of worker.Processing.fll_func / decision_direct
CONSTELLATIONS = ("bpsk", "qpsk", "8psk", "16psk", "16apsk")


def constellation_points(constellation):
    """Decision points in the reference's order: the tables of worker.py:40,49,58-60; for the sign
    slicers the values a decision index maps to (bpsk: +1, -1, 0; qpsk: quadrant order)."""
    if constellation == "bpsk":
        return np.array([1, -1, 0], np.complex128)
    if constellation == "qpsk":
        return np.array([1 + 1j, -1 + 1j, 1 - 1j, -1 - 1j]) / np.sqrt(2)
    if constellation == "8psk":
        return np.exp(1j * (np.pi / 4 * np.arange(8) - np.pi))
    if constellation == "16psk":
        return np.exp(1j * (np.pi / 8 * np.arange(16) - np.pi))
    if constellation == "16apsk":
        inner_ring = 0.5 * np.exp(1j * (np.pi / 4 + np.arange(4) * np.pi / 2))
        outer_ring = 1 * np.exp(1j * (np.pi / 12 + np.arange(12) * np.pi / 6))
        return np.concatenate([inner_ring, outer_ring])
    raise ValueError(f"unknown constellation {constellation!r}")


def fll_slicer_batch(x, Bn=0.001, y=1, constellation="qpsk", *, lens=None, outputs="all", ctx=None):
    """worker.Processing.fll_func (worker.py:18-78) on [B, M] bursts at one sample per symbol.
    Returns a dict: sig / freq / ef [B, M] (x's precision), dec uint8 decision indices into
    constellation_points(constellation), curr_freq float64[B]."""
    ctx = ctx or default_context()
    x, dt, rdt = _as_bursts(x)
    B, M = x.shape
    kind = 0 if constellation == "bpsk" else (1 if constellation == "qpsk" else 2)
    pts = np.ascontiguousarray(constellation_points(constellation), dtype=np.complex128) if kind == 2 else None
    names = ("sig", "freq", "ef", "dec") if outputs == "all" else tuple(outputs)
    bufs = {}
    for nm in names:
        bufs[nm] = np.zeros((B, M), dtype=x.dtype if nm == "sig" else (np.uint8 if nm == "dec" else rdt))
    cf = np.zeros(B, np.float64)
    g = lambda nm: _vp(bufs.get(nm))  # noqa: E731
    check(_lib.lib().dsp_fll_slicer(ctx.handle, _vp(x), dt, B, M, M, _vp(_lens(lens, B, M)), kind, _vp(pts),
                                   pts.size if pts is not None else 0, float(Bn), float(y),
                                   g("sig"), g("freq"), g("ef"), g("dec"), M, _vp(cf)))
    bufs["curr_freq"] = cf
    return bufs


def fll_func_constellation(signal, Bn=0.001, y=1, constellation="qpsk", *, ctx=None):
    """worker.Processing.fll_func(signal, Bn, y) for self.constellation (worker.py:18-78):
    (signal_list complex128, curr_freq float, curr_freq_list, ef_n_list)."""
    x = np.asarray(signal)
    n = x.size
    if n == 0:
        return np.zeros(0, complex), 0, np.zeros(0), np.zeros(0)
    o = fll_slicer_batch(x.reshape(1, -1).astype(np.complex128, copy=False), Bn, y, constellation,
                         outputs=("sig", "freq", "ef"), ctx=ctx)
    return o["sig"][0], float(o["curr_freq"][0]), o["freq"][0], o["ef"][0]


def decide_symbols(signal_list, constellation="qpsk", *, ctx=None):
    """The slicer of decision_direct (worker.py:84-123): (decided complex64 points, indices)."""
    x = np.asarray(signal_list)
    if x.size == 0:
        return np.zeros(0, np.complex64), np.zeros(0, np.uint8)
    o = fll_slicer_batch(x.reshape(1, -1).astype(np.complex128, copy=False), 0.0, 1, constellation,
                         outputs=("dec",), ctx=ctx)
    idx = o["dec"][0]
    pts = constellation_points(constellation)
    if constellation == "qpsk":
        # a zero component decides 0 in that component (np.sign(0) = 0), which no quadrant encodes
        d = (np.sign(x.real) + 1j * np.sign(x.imag)) / np.sqrt(2)
        zero = (x.real == 0) | (x.imag == 0)
        return np.where(zero, d, pts[idx]).astype(np.complex64), idx
    return pts[idx].astype(np.complex64), idx

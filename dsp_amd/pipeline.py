"""Device-resident burst pipeline: generate -> matched filter -> fused Gardner+FLL.

Everything stays in HBM in the tiled layout (include/dsp_hip.h); this is the
shape of the reference's receiver loop (worker.py:276-304: for every burst
filter_samples -> gardner_timing_recovery -> fll_func) batched over B bursts,
and it is what bench.py times.  Bursts are independent, so a multi-GPU job
simply gives every rank its own contiguous block of bursts (shard_range); no
collective is involved.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import CaptureArgs, FirArgs, SyncArgs, check, DSP_C64, DSP_I16
from .api import rrc_filter
from .sharding import shard_range  # noqa: F401  (the job partition lives in sharding.py)

_OUT_ELEM = {"rec": 8, "err": 4, "mu": 4, "sig": 8, "freq": 4, "ef": 4, "dec": 1}
_OUT_DTYPE = {"rec": np.complex64, "err": np.float32, "mu": np.float32, "sig": np.complex64,
              "freq": np.float32, "ef": np.float32, "dec": np.uint8}


class BurstPipeline:
    """B bursts of n_sym*sps complex64 samples on one GPU."""

    def __init__(self, ctx, B, n_sym=2048, sps=4, span=10, rolloff=0.35, alpha=0.007, mu0=0.0,
                 Bn=0.03, y=1.0, outputs="all", variant=0, interp_half_len=0):
        self.ctx, self.L = ctx, _lib.lib()
        self.B, self.n_sym, self.sps = int(B), int(n_sym), int(sps)
        self.N = self.n_sym * self.sps
        self.alpha, self.mu0, self.Bn, self.y, self.variant = alpha, mu0, Bn, y, variant
        self.taps = rrc_filter(sps, span, rolloff)
        self.ntiles = (self.B + 63) // 64
        self.rows = int(self.L.dsp_tile_rows(self.N))
        self.out_rows = self.n_sym + 8
        self.names = tuple(_OUT_ELEM) if outputs == "all" else tuple(outputs)
        self._bufs = []
        self.d_taps = self._alloc(self.taps.nbytes)
        ctx.h2d(self.d_taps, self.taps)
        self.d_raw = None
        self.d_tiles = self._alloc(self.ntiles * self.rows * 64 * 8)
        self.d_std = self._alloc(self.B * 8)
        self.d_inv = self._alloc(self.B * 8)
        self.d_out = {k: self._alloc(self.ntiles * self.out_rows * 64 * _OUT_ELEM[k]) for k in self.names}
        self.d_nout = self._alloc(self.B * 4)
        self.d_status = self._alloc(self.B * 4)
        ws = int(self.L.dsp_fir_workspace_bytes(0, self.B, self.N, self.taps.size))
        self.d_ws, self.ws_bytes = self._alloc(max(ws, 16)), ws
        a = SyncArgs()
        a.x = self.d_tiles; a.dtype = 0; a.variant = variant; a.B = self.B; a.N = self.N
        a.in_rows = self.rows; a.sps = self.sps; a.interp_half_len = int(interp_half_len)
        a.alpha = alpha; a.mu0 = mu0; a.do_fll = 1; a.Bn = Bn; a.y = y
        for k in self.names:
            setattr(a, k, self.d_out[k])
        a.out_rows = self.out_rows; a.n_out = self.d_nout; a.status = self.d_status
        self._sync_args = a

    def _alloc(self, n):
        p = self.ctx.malloc(n)
        self._bufs.append(p)
        return p

    def close(self):
        for p in self._bufs:
            self.ctx.free(p)
        self._bufs = []
        if self.d_raw:
            self.ctx.free(self.d_raw)
            self.d_raw = None

    # -- stages -------------------------------------------------------------
    def generate(self, seed=20251114, b0=0, esn0_db=20.0):
        """Synthesize the bursts on the device (raw, before the RX matched filter)."""
        if self.d_raw is None:
            self.d_raw = self.ctx.malloc(self.B * self.N * 8)
        step = 32768
        for s0 in range(0, self.B, step):
            nb = min(step, self.B - s0)
            check(self.L.dsp_synth_bursts_dev(self.ctx.handle, C.c_uint64(seed), b0 + s0, nb, self.n_sym,
                                              self.sps, C.c_void_p(self.d_taps), self.taps.size,
                                              float(esn0_db), C.c_void_p(self.d_raw + s0 * self.N * 8)))

    def upload(self, X):
        """Raw bursts from the host ([B, N] complex64)."""
        X = np.ascontiguousarray(X, dtype=np.complex64)
        assert X.shape == (self.B, self.N)
        if self.d_raw is None:
            self.d_raw = self.ctx.malloc(self.B * self.N * 8)
        self.ctx.h2d(self.d_raw, X)

    def matched_filter(self, free_raw=False):
        """RX RRC 'same' + np.std (worker.py:173-178) in ONE filter pass: the tiles receive
        fl32(y64), the pass's own per-chunk statistics give std and 1/std per burst, and the
        recurrence kernel multiplies by 1/std in float64 as it loads (what dsp_rx_chain does)."""
        f = FirArgs()
        f.src = self.d_raw; f.in_kind = 0; f.start = -1; f.B = self.B; f.N = self.N; f.src_stride = self.N
        f.h = self.d_taps; f.n_taps = self.taps.size; f.rows = self.rows
        f.workspace = self.d_ws; f.workspace_bytes = self.ws_bytes
        f.tiles = self.d_tiles; f.std_out = self.d_std; f.inv_std_out = self.d_inv
        check(self.L.dsp_fir_tiles_dev(self.ctx.handle, C.byref(f)))
        self._sync_args.scale = self.d_inv
        if free_raw:
            self.free_raw()

    def free_raw(self):
        """Release the raw (unfiltered) bursts once the filter has consumed them."""
        if self.d_raw:
            self.ctx.synchronize()
            self.ctx.free(self.d_raw)
            self.d_raw = None

    def run(self):
        """One pass of the fused Gardner+FLL+decision kernel over all bursts (asynchronous)."""
        check(self.L.dsp_sync_tiles_dev(self.ctx.handle, C.byref(self._sync_args)))

    # -- read-back ------------------------------------------------------------
    def n_out(self):
        n = np.empty(self.B, np.int32)
        self.ctx.d2h(n, self.d_nout)
        st = np.empty(self.B, np.int32)
        self.ctx.d2h(st, self.d_status)
        if st.any():
            raise RuntimeError("output capacity reached for some bursts")
        return n

    def fetch_tiles(self, d_tiles, elem, rows, dtype, b_lo, b_hi, ncopy):
        """Burst-major copy of bursts [b_lo, b_hi) (b_lo a multiple of 64) of a tiled array."""
        assert b_lo % 64 == 0
        nb = b_hi - b_lo
        d_tmp = self.ctx.malloc(nb * ncopy * elem)
        try:
            src = d_tiles + (b_lo // 64) * rows * 64 * elem
            check(self.L.dsp_untile_dev(self.ctx.handle, C.c_void_p(src), elem, rows, nb,
                                        C.c_void_p(d_tmp), ncopy, ncopy))
            out = np.empty((nb, ncopy), dtype)
            self.ctx.d2h(out, d_tmp)
        finally:
            self.ctx.free(d_tmp)
        return out

    def fetch(self, name, b_lo=0, b_hi=None):
        """Output `name` of bursts [b_lo, b_hi), burst-major [.., out_rows - 1].  Row r of burst b is valid
        for r < n_out()[b] (mu / err: n_out - 1); the device tiles are not cleared between runs, so what
        lies beyond is scratch (the host-pointer entry points return zeros there)."""
        b_hi = self.B if b_hi is None else b_hi
        return self.fetch_tiles(self.d_out[name], _OUT_ELEM[name], self.out_rows, _OUT_DTYPE[name],
                                b_lo, b_hi, self.out_rows - 1)

    def fetch_input(self, b_lo=0, b_hi=None):
        """The matched-filtered complex64 bursts fl32(y64) the recurrence kernel loaded; it multiplies
        them by fetch_inv_std() in float64 (x' = float64(y32) * (1/std))."""
        b_hi = self.B if b_hi is None else b_hi
        return self.fetch_tiles(self.d_tiles, 8, self.rows, np.complex64, b_lo, b_hi, self.N)

    def fetch_inv_std(self, b_lo=0, b_hi=None):
        """1/np.std of the filtered bursts [b_lo, b_hi) (float64), the scale of the recurrence loads."""
        b_hi = self.B if b_hi is None else b_hi
        inv = np.empty(b_hi - b_lo, np.float64)
        self.ctx.d2h(inv, self.d_inv + 8 * b_lo)
        return inv


class CapturePipeline:
    """One long capture resident in HBM -> segment-parallel receiver (north-star (c)).

    generate() synthesises a continuous QPSK stream on the device (int16 I,Q by default, the
    wire format of hw1/gardner.py:77-91); run() = peak normalisation, RRC 'same' / std of the
    whole stream, batched Gardner + FLL over overlapping segment windows, stitching; the
    decisions (and mu / frequency per symbol) stay on the device until fetch()."""

    def __init__(self, ctx, n_samples, sps=4, span=10, rolloff=0.35, alpha=0.007, mu0=0.0, Bn=0.003,
                 y=1.0, seg_len=None, warmup=2048, int16=True, matched_filter=True, want_mu=True,
                 want_freq=False, acquire=2):
        self.ctx, self.L = ctx, _lib.lib()
        self.n, self.sps = int(n_samples), int(sps)
        self.kind = DSP_I16 if int16 else DSP_C64
        self.taps = rrc_filter(sps, span, rolloff)
        if seg_len is None:
            from .api import auto_seg_len
            seg_len = auto_seg_len(self.n, sps, warmup)
        self.seg_len = int(seg_len)
        self.S = int(self.L.dsp_capture_segments(self.n, int(seg_len)))
        self.cap = self.n // self.sps + 8 * self.S + 64
        self._bufs = []
        self.d_taps = self._alloc(self.taps.nbytes)
        ctx.h2d(self.d_taps, self.taps)
        self.d_x = self._alloc(self.n * (4 if int16 else 8))
        self.d_dec = self._alloc(self.cap)
        self.d_mu = self._alloc(self.cap * 4) if want_mu else None
        self.d_freq = self._alloc(self.cap * 4) if want_freq else None
        self.total = C.c_int64(0)
        self.info = np.zeros((self.S, 4), np.int32)
        a = CaptureArgs()
        a.x = self.d_x; a.in_kind = self.kind; a.normalize = 1; a.n_samples = self.n
        if matched_filter:
            a.h = self.d_taps; a.n_taps = self.taps.size
        a.sps = self.sps; a.alpha = alpha; a.mu0 = mu0; a.Bn = Bn; a.y = y
        a.seg_len = int(seg_len); a.warmup = int(warmup); a.acquire = int(acquire)
        a.dec = self.d_dec; a.mu = self.d_mu; a.freq = self.d_freq; a.capacity = self.cap
        a.n_symbols = C.addressof(self.total); a.seg_info = self.info.ctypes.data
        self._args = a

    def _alloc(self, n):
        p = self.ctx.malloc(n)
        self._bufs.append(p)
        return p

    def close(self):
        for p in self._bufs:
            self.ctx.free(p)
        self._bufs = []

    def generate(self, seed=20251114, esn0_db=20.0, cfo=1.5e-4, tau0=0.3, drift=0.0, amplitude=6000.0):
        check(self.L.dsp_synth_stream_dev(self.ctx.handle, C.c_uint64(seed), self.n, self.sps,
                                          C.c_void_p(self.d_taps), self.taps.size, float(esn0_db), float(cfo),
                                          float(tau0), float(drift), self.kind, float(amplitude),
                                          C.c_void_p(self.d_x)))

    def upload(self, x):
        x = np.ascontiguousarray(x, dtype=np.int16 if self.kind == DSP_I16 else np.complex64)
        assert x.nbytes == self.n * (4 if self.kind == DSP_I16 else 8)
        self.ctx.h2d(self.d_x, x)

    def run(self):
        """The whole segmented receiver; synchronises; returns the number of symbols."""
        check(self.L.dsp_sync_capture_segmented_dev(self.ctx.handle, C.byref(self._args)))
        return int(self.total.value)

    def fetch(self, name="dec", lo=0, hi=None):
        hi = int(self.total.value) if hi is None else hi
        d, dt = {"dec": (self.d_dec, np.uint8), "mu": (self.d_mu, np.float32),
                 "freq": (self.d_freq, np.float32)}[name]
        out = np.empty(hi - lo, dt)
        self.ctx.d2h(out, d + lo * out.itemsize)
        return out

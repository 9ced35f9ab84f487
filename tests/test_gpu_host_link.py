"""The host link of the host-pointer entry points (dsp_api.hip: link_h2d / link_d2h): transfers of 4 MiB or
more from / to pageable arrays are staged through a page-locked ring by host threads; page-locked arrays
(dsp_host_alloc -> dsp_amd.pinned_empty) go over the link directly.  Same bytes either way, and the same
results from an entry point whichever way its arrays travel."""
import numpy as np
import pytest

import dsp_amd

pytestmark = pytest.mark.gpu

MiB = 1 << 20


@pytest.fixture(scope="module")
def ctx():
    c = dsp_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("nbytes", [1, 4 * MiB - 1, 4 * MiB, 16 * MiB, 16 * MiB + 5, 64 * MiB, 64 * MiB + 123, 150 * MiB + 7])
@pytest.mark.parametrize("threads", [4, 1, 0, 3])
def test_round_trip_bytes(ctx, nbytes, threads):
    """Every slot boundary: below the staging threshold, one slot, the ring's four slots exactly, more than
    the ring (slots reused), odd tails; 0 threads = the runtime's own pageable copy."""
    ctx.set_option("link_threads", threads)
    try:
        rng = np.random.default_rng(nbytes % 9973 + threads)
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)
        back = np.zeros(nbytes + 64, dtype=np.uint8)            # 64 guard bytes behind the destination
        d = ctx.malloc(nbytes)
        try:
            ctx.h2d(d, src)
            ctx.d2h(back[:nbytes], d)
        finally:
            ctx.free(d)
        assert np.array_equal(back[:nbytes], src)
        assert not back[nbytes:].any()
    finally:
        ctx.set_option("link_threads", 4)


def test_page_locked_arrays_round_trip(ctx):
    n = 40 * MiB + 3
    src = dsp_amd.pinned_empty(n, np.uint8)
    back = dsp_amd.pinned_empty(n, np.uint8)
    src[:] = np.random.default_rng(5).integers(0, 256, n, dtype=np.uint8)
    back[:] = 0
    d = ctx.malloc(n)
    try:
        ctx.h2d(d, src)
        ctx.d2h(back, d)
    finally:
        ctx.free(d)
    assert np.array_equal(back, src)
    shaped = dsp_amd.pinned_empty((3, 5), np.complex64)
    assert shaped.shape == (3, 5) and shaped.dtype == np.complex64
    del src, back, shaped                                       # blocks go back with their last view


def test_chain_results_do_not_depend_on_the_link(ctx):
    """sync_chain_batch on 1200 x 8192 complex64 bursts (79 MB in: the ring wraps; 20 MB per complex output):
    staged, runtime-copied and page-locked inputs give the same bits."""
    B, N, sps = 1200, 8192, 4
    rng = np.random.default_rng(11)
    sym = ((rng.integers(0, 2, (B, N // sps)) * 2 - 1) + 1j * (rng.integers(0, 2, (B, N // sps)) * 2 - 1)) / np.sqrt(2)
    x = np.repeat(sym, sps, axis=1).astype(np.complex64)
    x += (0.05 * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))).astype(np.complex64)
    got = {}
    for mode in ("staged", "runtime", "page_locked"):
        ctx.set_option("link_threads", 0 if mode == "runtime" else 4)
        xin = x
        if mode == "page_locked":
            xin = dsp_amd.pinned_empty(x.shape, x.dtype)
            xin[...] = x
        try:
            got[mode] = dsp_amd.sync_chain_batch(xin, sps, 0.007, 0.0, 0.03, 1, ctx=ctx)
        finally:
            ctx.set_option("link_threads", 4)
    assert int(got["staged"]["n_out"].min()) > 2000
    for mode in ("runtime", "page_locked"):
        for k, v in got["staged"].items():
            assert np.array_equal(v, got[mode][k]), (mode, k)

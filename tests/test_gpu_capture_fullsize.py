"""BASELINE config 5 at full size: ONE int16 capture of 2^30 samples through the segment-parallel
receiver.  The sequential oracle would need minutes for this, so the checks are the
size-independent properties the domain offers: the symbol count of a drifting clock, decisions equal
to the transmitted symbols (up to ONE global QPSK rotation and a few symbol slips) at the start,
three interior places and the very end of the stitched stream, every segment's share accounted
for, and idempotence of a second run.  Small captures are compared with the oracle bit for bit in
tests/test_segmented.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20251114
TURN = np.array([1, 3, 0, 2], np.uint8)       # quadrant index after a +90 degree turn


def _best_alignment(dec, centre, span=24):
    """Best match of a run of decisions against the transmitted symbols around index `centre`:
    -> (match rate, symbol offset, rotation)."""
    from dsp_amd.synth import stream_symbols
    n = dec.size
    tx = stream_symbols(SEED, centre - span, n + 2 * span)
    best = (0.0, 0, 0)
    d = dec.copy()
    for rot in range(4):
        for off in range(2 * span + 1):
            r = float(np.mean(d == tx[off:off + n]))
            if r > best[0]:
                best = (r, off - span, rot)
        d = TURN[d]
    return best


# (65536, 2048) is SURVEY.md 8(d)'s config 5 as written; None = the automatic segment length; acquire 1 = windows
# started on the sample grid (round 2's definition), 2 = at the fractional phase (the default; bench.py's line
# is (None, 2048, 2))
@pytest.mark.parametrize("seg_len,warmup,acquire", [(32768, 4096, 1), (65536, 4096, 2), (65536, 2048, 2), (None, 2048, 2),
                                                    (None, 1536, 2)])
def test_capture_2p30(ctx, seg_len, warmup, acquire):
    from dsp_amd.pipeline import CapturePipeline
    n, sps, tau0, drift = 1 << 30, 4, 0.3, 2.0e-8
    pipe = CapturePipeline(ctx, n, sps, seg_len=seg_len, warmup=warmup, acquire=acquire)
    try:
        pipe.generate(SEED, tau0=tau0, drift=drift)
        nsym = pipe.run()
        assert abs(nsym - int(round(n * (1.0 + drift) / sps))) <= 4
        assert pipe.S == n // pipe.seg_len
        info = pipe.info.copy()
        # seg_info rows (first kept row, end row, rotation, rows produced): every window contributed a
        # non-empty share inside what it produced, and the shares add up to the stitched stream
        assert (info[:, 1] > info[:, 0]).all() and (info[:, 1] <= info[:, 3]).all() and (info[:, 0] >= 0).all()
        assert int((info[:, 1].astype(np.int64) - info[:, 0]).sum()) == nsym
        run = 8192
        rots = []
        for lo in (64, nsym // 4, nsym // 2 + 12345, 3 * (nsym // 4) + 777, nsym - run - 64):
            dec = pipe.fetch("dec", lo, lo + run)
            rate, off, rot = _best_alignment(dec, lo)
            assert rate > 0.999, (lo, rate, off, rot)
            assert abs(off) <= 8, (lo, off)                 # the drift moves the count by ~5 symbols end to end
            rots.append(rot)
        assert len(set(rots)) == 1, rots                    # ONE rotation ambiguity for the whole stream
        d0 = pipe.fetch("dec", nsym // 2, nsym // 2 + (1 << 20))
        assert pipe.run() == nsym
        assert np.array_equal(pipe.fetch("dec", nsym // 2, nsym // 2 + (1 << 20)), d0)
    finally:
        pipe.close()


@pytest.mark.parametrize("warmup", [2048])
def test_capture_2p30_windows_against_the_oracle(ctx, oracle, warmup):
    """Oracle parity INSIDE the full-size capture.  A complex64 capture of 2^30 samples that needs no filter
    (so a window depends on nothing but its own samples: no whole-stream peak or std) goes through the
    segment-parallel receiver; runs of 192 consecutive segments -- the stream's first, a run from its middle
    and its last -- are then handed to the CPU oracle's segmented definition as streams of their own.
    Away from an excerpt's artificial ends (two windows) its windows see the same samples, warm-up and
    neighbours as in the capture, so their shares (first / end row, rows produced), their decisions (up to
    the ONE rotation the two chains of votes differ by) and their mu values (float32) must be the device's,
    bit for bit; at the capture's true start and end the excerpt's end windows are compared too."""
    from dsp_amd.pipeline import CapturePipeline
    n, sps, K = 1 << 30, 4, 192
    pipe = CapturePipeline(ctx, n, sps, warmup=warmup, int16=False, matched_filter=False)
    try:
        pipe.generate(SEED, tau0=0.3, drift=2.0e-8)
        nsym = pipe.run()
        info = pipe.info.astype(np.int64)
        L, S = pipe.seg_len, pipe.S
        assert S * L == n and S > 3 * K
        share = info[:, 1] - info[:, 0]
        off = np.concatenate(([0], np.cumsum(share)))           # where window w's share starts in the stitched stream
        assert off[-1] == nsym
        for w0, j_lo, j_hi in ((0, 0, K - 2), (S // 2 - 7, 2, K - 2), (S - K, 2, K)):
            ex = np.empty(K * L, np.complex64)
            ctx.d2h(ex, pipe.d_x + w0 * L * 8)
            ref = oracle.sync_capture_segmented(ex, sps, 0.007, 0.0, 0.003, 1.0, seg_len=L, warmup=warmup, acquire=2)
            rinfo = ref["seg_info"].astype(np.int64)
            roff = np.concatenate(([0], np.cumsum(rinfo[:, 1] - rinfo[:, 0])))
            turns = set()
            for j in range(j_lo, j_hi):
                w = w0 + j
                assert tuple(info[w, [0, 1, 3]]) == tuple(rinfo[j, [0, 1, 3]]), (w0, j, info[w], rinfo[j])
                turn = int(info[w, 2] - rinfo[j, 2]) & 3
                turns.add(turn)
                r = ref["dec"][roff[j]:roff[j + 1]].copy()
                for _ in range(turn):
                    r = TURN[r]
                assert np.array_equal(pipe.fetch("dec", int(off[w]), int(off[w + 1])), r), (w0, j)
                assert np.array_equal(pipe.fetch("mu", int(off[w]), int(off[w + 1])), ref["mu"][roff[j]:roff[j + 1]]), (w0, j)
            assert len(turns) == 1, (w0, turns)
            if w0 == 0:
                assert turns == {0}                              # the same chain of votes from the same first window
    finally:
        pipe.close()

import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import dsp_amd
rng = np.random.default_rng(1)
nsym, sps = 50000, 4
sym = ((rng.integers(0, 2, nsym) * 2 - 1) + 1j * (rng.integers(0, 2, nsym) * 2 - 1)) / np.sqrt(2)
x = np.repeat(sym, sps) + 0.05 * (rng.standard_normal(nsym * sps) + 1j * rng.standard_normal(nsym * sps))
h = dsp_amd.rrc_filter(4, 10, 0.35)
x = np.convolve(x, h, "same"); x /= x.std()
def t(fn, n=3):
    fn(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r
for name, fn in (("gardner_timing_recovery", lambda: dsp_amd.gardner_timing_recovery(x, 4, 0.007)),
                 ("gardner_ted", lambda: dsp_amd.gardner_ted(x, 4)),
                 ("gardner_ted sps2", lambda: dsp_amd.gardner_ted(x[::2].copy(), 2))):
    ms, r = t(fn)
    print(name, "%.2f ms" % ms, len(r[0]), "symbols -> %.3f us/symbol" % (ms * 1e3 / len(r[0])), flush=True)
rec = dsp_amd.gardner_timing_recovery(x, 4, 0.007)[0]
ms, r = t(lambda: dsp_amd.fll_func(rec, 0.03, 1))
print("fll_func", "%.2f ms" % ms, len(rec), "symbols -> %.3f us/symbol" % (ms * 1e3 / len(rec)))
ms, r = t(lambda: dsp_amd.sync_chain_batch(x[None, :], 4, 0.007, 0.0, 0.03, 1))
print("sync_chain_batch B=1", "%.2f ms" % ms, "-> %.3f us/symbol" % (ms * 1e3 / nsym))
for cst in ("bpsk", "8psk", "16apsk"):
    ms, r = t(lambda: dsp_amd.fll_func_constellation(rec, 0.03, 1, cst))
    print("fll_func_constellation", cst, "%.2f ms" % ms, "-> %.3f us/symbol" % (ms * 1e3 / len(rec)))

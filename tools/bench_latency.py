#!/usr/bin/env python3
"""Single-burst latency of the reference's two real-file flows (BASELINE configs 1 and 2) through the
host-pointer entry points (upload, kernels, download included):

  config 1  qpsk_high_snr_sps_4_float32.pcm -> RRC 'same' + /std -> gardner_timing_recovery(alpha .007)
            -> fll_func(Bn .03)                                          (gardner_real_file_2.py:9-29)
  config 2  sig_symb_x4_ncr capture -> find_preamble_offset -> f_rel (method 2) -> derotate -> RRC + /std
            -> Gardner -> FLL                                            (diff_method.py:164-307)

One burst is ONE lane: these are latency numbers (sequential recurrence, ~0.6 us per symbol), not
throughput.  Prints one JSON line; the reference's Python path takes ~0.2 s (config 1) and ~1.8 s
(config 2, dominated by the preamble search) on one core of the build container.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsp_amd  # noqa: E402

D = os.path.join(ROOT, "tests", "golden", "data")


def best(f, n=5):
    f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


def main():
    for kv in os.environ.get("DSP_LATENCY_OPTS", "").split(","):        # e.g. split=0,queue_block=8
        if kv:
            dsp_amd.default_context().set_option(kv.split("=")[0], int(kv.split("=")[1]))
    h = dsp_amd.rrc_filter(4, 10, 0.35)
    x1 = dsp_amd.read_iq_float32(os.path.join(D, "qpsk_high_snr_sps_4_float32.pcm")).astype(np.complex128)
    res = {}

    def cfg1():
        return dsp_amd.rx_chain_batch(x1[None, :], h, 4, 0.007, 0.0, 0.03, 1, outputs=("sig", "dec", "mu", "freq"))
    o = cfg1()
    res["config1"] = {"samples": int(x1.size), "symbols": int(o["n_out"][0]), "ms": best(cfg1)}

    raw = dsp_amd.read_iq_float32(os.path.join(D, "sig_symb_x4_ncr_77671952566_logon_id_1_tx_id_7616.pcm"))
    pre = dsp_amd.read_iq_float32(os.path.join(D, "preambule_logon_id_2_tx_id_7616_float32.pcm"))
    from scipy.signal import firwin, lfilter
    sig = raw * np.exp(-1j * 2 * np.pi * 0.0105 * np.arange(len(raw)))          # diff_method.py:172-186
    sig = lfilter(firwin(101, cutoff=0.14, window="hamming", fs=1.0), 1.0, sig)
    sig = sig / np.std(sig)

    def search():
        return dsp_amd.find_preamble_offset(sig, pre, 4)

    def cfg2():
        offset, aligned, _, _, phase = search()
        return dsp_amd.rx_bursts(sig, [offset * 4 + phase], len(aligned), pre, h, 4, 0.007, 0.0, 0.03, 1,
                                 cfo_method=2, outputs=("sig", "dec"))
    o = cfg2()
    res["config2"] = {"samples": int(sig.size), "symbols": int(o["n_out"][0]), "f_rel": float(o["f_rel"][0]),
                      "preamble_search_ms": best(search), "ms": best(cfg2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

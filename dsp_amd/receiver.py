"""End-to-end burst receiver with the reference's ``worker.Processing`` interface
(worker.py:8-337, QPSK): preamble search -> correlation peaks -> per burst: coarse CFO from the
preamble, derotation, RRC matched filter + /np.std, Gardner, FLL.

The loops that cost the reference its run time are device kernels here: the sliding normalised
differential correlation (``dsp_diff_corr``, worker.py:183-245) and the whole per-burst body --
phase-slope CFO estimate over the preamble symbols, cut + derotation, filter -> Gardner -> FLL
(worker.py:284-304) -- which runs for ALL found bursts in one device call (``dsp_rx_bursts``).
Peak picking on the correlation magnitudes stays scipy's ``find_peaks`` as in the reference, and
the quirks are kept (the cut offset is ``peak // 2`` in both modes, worker.py:276-283).
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import shift as array_shift          # worker.py:3
from scipy.signal import find_peaks                     # worker.py:4

from .api import (CONSTELLATIONS, decide_symbols, diff_corr, fll_func_constellation, rrc_filter, rx_bursts)


class Processing:
    def __init__(self, signal=None, preamble=None, packet_symb=None, sps=4, constellation="qpsk", interp=False,
                 *, ctx=None):
        if constellation not in CONSTELLATIONS:
            raise ValueError(f"constellation must be one of {CONSTELLATIONS}")
        self.signal = signal
        self.preamble = preamble
        self.packet_symb = packet_symb      # preamble + payload symbols of one packet
        self.sps = sps
        self.constellation = constellation
        self.interp = interp
        self._ctx = ctx

    # worker.py:18-78
    def fll_func(self, signal, Bn=0.001, y=1):
        return fll_func_constellation(signal, Bn, y, self.constellation, ctx=self._ctx)

    # worker.py:80-170.  The reference prints the match percentages and returns nothing; here the
    # decided symbols are returned and the percentages kept in self.match_rates:
    # [(None, pct without rotation), (degrees, pct) per trial rotation].
    def decision_direct(self, signal_list):
        decision_direct, _ = decide_symbols(signal_list, self.constellation, ctx=self._ctx)     # :120-123
        self.match_rates = []
        if self.packet_symb is not None and len(self.packet_symb) == len(decision_direct):     # :129-130
            ref = np.asarray(self.packet_symb)
            n = len(decision_direct)
            self.match_rates.append((None, np.count_nonzero(
                np.isclose(decision_direct, ref, rtol=1e-5, atol=1e-5)) / n * 100))             # :137-142
            num_phases, phase_step = {"qpsk": (4, np.pi / 2), "8psk": (8, np.pi / 4),
                                      "16psk": (16, np.pi / 8)}.get(self.constellation, (16, 2 * np.pi / 16))
            for phase in range(num_phases):                                                     # :158-170
                shifted = decision_direct * np.exp(1j * phase_step * phase)
                cnt = np.count_nonzero(np.isclose(shifted, ref, rtol=1e-5, atol=1e-5))
                self.match_rates.append((float(np.degrees(phase_step * phase)), cnt / n * 100))
        return decision_direct

    # worker.py:180-245
    def find_preamble_offset_with_interpolate_dots(self, signal_iq, preamble_iq, sps, interp=False):
        conv_results_interp = []
        if interp:
            shifted = array_shift(signal_iq, shift=0.5, mode="nearest")          # :189
            conv_results_interp = diff_corr(shifted, preamble_iq, sps, ctx=self._ctx)
        conv_results = diff_corr(signal_iq, preamble_iq, sps, ctx=self._ctx)
        return conv_results, conv_results_interp

    # worker.py:247-273
    def find_conv_peaks(self, conv_results, conv_results_interp, interp=False):
        n0 = len(conv_results[0])
        if conv_results_interp:
            corr_abs = np.zeros(self.sps * 2 * n0)
            for i, c in enumerate(conv_results):
                corr_abs[i * 2:i * 2 + len(c) * self.sps * 2:self.sps * 2] = np.abs(c)
            for i, c in enumerate(conv_results_interp):
                corr_abs[i * 2 + 1:i * 2 + 1 + len(c) * self.sps * 2:self.sps * 2] = np.abs(c)
        else:
            corr_abs = np.zeros(self.sps * n0)
            for i, c in enumerate(conv_results):
                corr_abs[i:i + len(c) * self.sps:self.sps] = np.abs(c)
        max_corr = np.max(corr_abs)
        peaks, _ = find_peaks(corr_abs, height=max_corr * 0.7, distance=len(self.packet_symb) * 0.1,
                              prominence=max_corr * 0.1)
        return peaks

    # worker.py:276-337
    def process(self):
        conv_results, conv_results_interp = self.find_preamble_offset_with_interpolate_dots(
            self.signal, self.preamble, self.sps, self.interp)
        peaks = self.find_conv_peaks(conv_results, conv_results_interp, self.interp)
        n_pkt = len(self.packet_symb) * self.sps
        signal_iq = array_shift(self.signal, shift=0.5, mode="nearest") if conv_results_interp else self.signal
        offsets = [(peak - 1) // 2 if conv_results_interp else peak // 2 for peak in peaks]      # :279-283
        offsets = [o for o in offsets if o + n_pkt <= len(signal_iq)]                            # :284-285
        found = []
        self.f_rel = np.zeros(0)
        if offsets:
            h = rrc_filter(self.sps, 10, 0.35)                                                   # :174
            qpsk = self.constellation == "qpsk"        # the fused Gardner+FLL kernel slices QPSK
            out = rx_bursts(signal_iq, offsets, n_pkt, self.preamble, h, self.sps, 0.007, 0.0, 0.03, 1,
                            cfo_method=1, outputs=("sig",) if qpsk else ("rec",), ctx=self._ctx)   # :287-304
            self.f_rel = out["f_rel"]                                                            # :294
            key = "sig" if qpsk else "rec"
            found = [out[key][b, :int(out["n_out"][b])].copy() for b in range(len(offsets))]
            if not qpsk:
                found = [self.fll_func(rec, Bn=0.03)[0] for rec in found]                        # :304
            self.decisions = [self.decision_direct(s) for s in found]                            # :332
        return peaks, found

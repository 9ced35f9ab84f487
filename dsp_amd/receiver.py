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

from .api import diff_corr, rrc_filter, rx_bursts


class Processing:
    def __init__(self, signal=None, preamble=None, packet_symb=None, sps=4, constellation="qpsk", interp=False,
                 *, ctx=None):
        if constellation != "qpsk":
            raise ValueError("only the QPSK receiver is built (SURVEY.md 8f rank 4)")
        self.signal = signal
        self.preamble = preamble
        self.packet_symb = packet_symb      # preamble + payload symbols of one packet
        self.sps = sps
        self.constellation = constellation
        self.interp = interp
        self._ctx = ctx

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
            out = rx_bursts(signal_iq, offsets, n_pkt, self.preamble, h, self.sps, 0.007, 0.0, 0.03, 1,
                            cfo_method=1, outputs=("sig",), ctx=self._ctx)                       # :287-304
            self.f_rel = out["f_rel"]                                                            # :294
            found = [out["sig"][b, :int(out["n_out"][b])].copy() for b in range(len(offsets))]
        return peaks, found

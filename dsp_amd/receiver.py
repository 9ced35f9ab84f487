"""End-to-end burst receiver with the reference's ``worker.Processing`` interface
(worker.py:8-337, QPSK): preamble search -> correlation peaks -> per burst: coarse CFO from the
preamble, derotation, RRC matched filter + /np.std, Gardner, FLL.

The two loops that cost the reference its run time are device kernels here: the sliding
normalised differential correlation (``dsp_diff_corr``, worker.py:183-245) and the per-burst
filter -> Gardner -> FLL chain, which runs for ALL found bursts in one batched call
(``dsp_rx_chain``, worker.py:299-304).  The glue in between -- peak picking on the correlation
magnitudes, the phase-slope estimate over the preamble symbols, the derotation -- is a few hundred
numpy operations per burst and follows the reference line by line, quirks included (the cut offset
is ``peak // 2`` in both modes, worker.py:276-283).
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import shift as array_shift          # worker.py:3
from scipy.signal import find_peaks                     # worker.py:4

from .api import diff_corr, rrc_filter, rx_chain_batch


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
        bursts = []
        signal_iq = array_shift(self.signal, shift=0.5, mode="nearest") if conv_results_interp else self.signal
        for peak in peaks:
            offset = (peak - 1) // 2 if conv_results_interp else peak // 2          # :279-283
            if offset + n_pkt > len(signal_iq):
                continue
            cut = signal_iq[offset:offset + n_pkt]
            s = cut[::self.sps][:len(self.preamble)]
            phases = np.angle(s * np.conj(self.preamble))                              # :289-291
            f_rel = np.mean(np.diff(np.unwrap(phases))) / (2 * np.pi * self.sps)        # :292-294
            bursts.append(cut * np.exp(-1j * 2 * np.pi * f_rel * (np.arange(len(cut)) + offset)))   # :298
        found = []
        if bursts:
            X = np.stack(bursts).astype(np.complex128)
            h = rrc_filter(self.sps, 10, 0.35)                                         # :174
            out = rx_chain_batch(X, h, self.sps, 0.007, 0.0, 0.03, 1, outputs=("sig",), ctx=self._ctx)   # :299-304
            found = [out["sig"][b, :int(out["n_out"][b])].copy() for b in range(len(bursts))]
        return peaks, found

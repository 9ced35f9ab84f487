"""Test-signal generator with the reference's ``worker.Generation`` interface (worker.py:339-427) and
the raw-PCM writer of test_w/1.py:183-195.

The random draws are numpy's legacy global generator in the reference's order (preamble, payload,
noise I, noise Q), so a seeded run reproduces the reference's symbols bit for bit; the pulse shaping
(RRC 'same' convolution + / np.std, worker.py:415-419) runs on the device through
``dsp_amd.filter_samples``.  Throughput benchmarks do not use this class: they generate bursts on
the device with a counter-based generator (``dsp_amd.synth``, ``dsp_synth_bursts_dev``).
"""
from __future__ import annotations

import numpy as np

from .api import filter_samples


class Generation:
    def __init__(self, constellation, preamble_length, signal_length, sps, Frel, Noise_power, N, Delay):
        self.constellation = constellation
        self.preamble_length = preamble_length
        self.signal_length = signal_length
        self.sps = sps
        self.Frel = Frel
        self.Noise_power = Noise_power
        self.N = N
        self.Delay = Delay

    def generate(self):                                     # worker.py:350-358
        preamble = self.generate_preamble()
        samples, symbols = self.generate_signal(preamble)
        samples = self.upsample(samples)
        samples = self.filter_samples(samples)
        samples = self.frequency_offset(samples)
        samples = self.add_noise(samples)
        return preamble, samples, symbols

    def _alphabet(self, payload):
        """The symbol alphabets of worker.py:361-376 (preamble) and :382-397 (payload); they differ
        for qpsk (0.707 vs 1/sqrt(2)), 8psk (rounded vs exact points) and 16apsk (outer-ring phase)."""
        c = self.constellation
        if c == "bpsk":
            return [-1 + 0j, 1 + 0j]
        if c == "qpsk":
            if payload:
                return [0.707 + 1j * 0.707, 0.707 - 1j * 0.707, -0.707 + 1j * 0.707, -0.707 - 1j * 0.707]
            return [(1 + 1j) / np.sqrt(2), (1 - 1j) / np.sqrt(2), (-1 + 1j) / np.sqrt(2), (-1 - 1j) / np.sqrt(2)]
        if c == "8psk":
            if payload:
                return np.exp(1j * (np.pi / 4 * np.arange(8) - np.pi))
            return [0.707 + 1j * 0.707, 0.707 - 1j * 0.707, -0.707 + 1j * 0.707, -0.707 - 1j * 0.707,
                    0 + 1j, 0 - 1j, 1 + 0j, -1 + 0j]
        if c == "16psk":
            return [np.exp(k * 1j * np.pi / 8) for k in range(1, 17)]
        if c == "16apsk":
            inner_ring = 0.5 * np.exp(1j * (np.pi / 4 + np.arange(4) * np.pi / 2))
            outer_ring = 1 * np.exp(1j * ((np.pi / 12 if payload else np.pi / 4) + np.arange(12) * np.pi / 6))
            return np.concatenate([inner_ring, outer_ring])
        raise ValueError(f"unsupported constellation: {c}")

    def generate_preamble(self):                            # :360-379
        return np.random.choice(self._alphabet(False), size=self.preamble_length)

    def generate_signal(self, preamble):                    # :381-407
        inform = np.random.choice(self._alphabet(True), size=self.signal_length - self.preamble_length)
        symbols = np.concatenate([preamble, inform])
        gap = np.zeros(self.Delay, dtype=np.complex64)
        signal = np.concatenate([gap] + [np.concatenate([symbols, gap]) for _ in range(self.N)])
        return signal, symbols

    def upsample(self, symbols):                            # :409-413
        samples = np.zeros(len(symbols) * self.sps, dtype=np.complex64)
        samples[::self.sps] = symbols
        return samples

    def filter_samples(self, samples):                      # :415-419 (device)
        return filter_samples(samples, self.sps, 10, 0.35)

    def frequency_offset(self, samples):                    # :421-423
        return samples * np.exp(1j * 2 * np.pi * self.Frel * np.arange(len(samples)))

    def add_noise(self, samples):                           # :425-427
        noise = (np.random.normal(0, self.Noise_power, len(samples))
                 + 1j * np.random.normal(0, self.Noise_power, len(samples))) / np.sqrt(2)
        return samples + noise


def save_as_pcm_float32_iq(x, filename):
    """test_w/1.py:183-195: raw little-endian float32 [I0, Q0, I1, Q1, ...], no header -- the format
    read_iq_float32 (and the reference's np.fromfile idiom) loads."""
    x = np.asarray(x)
    assert np.iscomplexobj(x), "expected a complex array"
    iq = np.empty(2 * len(x), dtype="<f4")
    iq[0::2] = x.real
    iq[1::2] = x.imag
    with open(filename, "wb") as f:
        iq.tofile(f)

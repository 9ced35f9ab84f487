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


_PARAMS = ("constellation", "preamble_length", "signal_length", "sps", "Frel", "Noise_power", "N", "Delay")


class Generation:
    def __init__(self, *args, **kwargs):
        """Generation(constellation, preamble_length, signal_length, sps, Frel, Noise_power, N, Delay):
        the parameter names and order of worker.Generation (worker.py:340), kept as attributes."""
        if len(args) > len(_PARAMS):
            raise TypeError(f"Generation takes {len(_PARAMS)} parameters: {', '.join(_PARAMS)}")
        given = dict(zip(_PARAMS, args))
        for name, value in kwargs.items():
            if name not in _PARAMS or name in given:
                raise TypeError(f"unexpected or repeated parameter {name!r}")
            given[name] = value
        missing = [n for n in _PARAMS if n not in given]
        if missing:
            raise TypeError(f"missing parameters: {', '.join(missing)}")
        for name in _PARAMS:
            setattr(self, name, given[name])

    def generate(self):
        """(preamble symbols, noisy baseband samples, symbols of one burst) -- the stages in the
        reference's order (worker.py:350-358); the draws happen in generate_preamble,
        generate_signal and add_noise, in that order."""
        preamble = self.generate_preamble()
        stream, symbols = self.generate_signal(preamble)
        for stage in (self.upsample, self.filter_samples, self.frequency_offset, self.add_noise):
            stream = stage(stream)
        return preamble, stream, symbols

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
        # the product is formed left to right exactly as the reference writes it: a seeded run is
        # compared bit for bit and any other association rounds differently
        rotation = np.exp(1j * 2 * np.pi * self.Frel * np.arange(len(samples)))
        return samples * rotation

    def add_noise(self, samples):                           # :425-427
        # two draws of len(samples) normals each, in-phase first: the reference's order of
        # consumption from numpy's global generator, which a seeded comparison depends on
        n, sigma = len(samples), self.Noise_power
        in_phase = np.random.normal(0, sigma, n)
        quadrature = np.random.normal(0, sigma, n)
        return samples + (in_phase + 1j * quadrature) / np.sqrt(2)


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

"""Seeded synthetic QPSK bursts (host side) -- the recipe of SURVEY.md section 8(d).

Per burst b, ``rng = np.random.default_rng([20251114, b])``:
2048 QPSK symbols uniform over (+-1 +-1j)/sqrt(2) -> zero-stuff x sps -> TX RRC
``np.convolve(..., 'same')`` (the recipe of the reference's generator,
worker.py:409-419) -> fractional delay tau ~ U(-0.5, 0.5) samples as
``np.interp(t, t - tau, x)`` (gardner2.py:131-133) -> CFO
``exp(j(2 pi f n + phi))``, f ~ U(-5e-4, 5e-4) cycles/sample, phi ~ U(0, 2 pi)
-> complex AWGN at Es/N0 = 20 dB in the form of worker.py:425-427 -> complex64.

This module is plain numpy and needs neither the GPU nor the oracle; the taps
are passed in by the caller (``dsp_amd.rrc_filter`` in product code).
"""
from __future__ import annotations

import numpy as np

SEED = 20251114
N_SYM = 2048
SPS = 4
ESN0_DB = 20.0


def burst_params(b: int):
    """The per-burst impairments, without generating samples."""
    rng = np.random.default_rng([SEED, int(b)])
    bits = rng.integers(0, 4, N_SYM)
    tau = rng.uniform(-0.5, 0.5)
    f = rng.uniform(-5e-4, 5e-4)
    phi = rng.uniform(0.0, 2 * np.pi)
    return bits, tau, f, phi, rng


def synth_burst(b: int, taps: np.ndarray, n_sym: int = N_SYM, sps: int = SPS,
                esn0_db: float = ESN0_DB, return_info: bool = False):
    """One burst of n_sym*sps complex64 samples at the TX output + channel."""
    rng = np.random.default_rng([SEED, int(b)])
    bits = rng.integers(0, 4, n_sym)
    tau = rng.uniform(-0.5, 0.5)
    f = rng.uniform(-5e-4, 5e-4)
    phi = rng.uniform(0.0, 2 * np.pi)
    const = np.array([1 + 1j, -1 + 1j, 1 - 1j, -1 - 1j]) / np.sqrt(2)  # index = (re<0)+2*(im<0)
    symbols = const[bits]
    x = np.zeros(n_sym * sps, dtype=np.complex128)
    x[::sps] = symbols
    x = np.convolve(x, taps, mode="same")
    t = np.arange(x.size, dtype=np.float64)
    x = np.interp(t, t - tau, x)
    x = x * np.exp(1j * (2 * np.pi * f * t + phi))
    sigma = np.sqrt(10.0 ** (-esn0_db / 10.0))
    noise = (rng.normal(0, sigma, x.size) + 1j * rng.normal(0, sigma, x.size)) / np.sqrt(2)
    x = (x + noise).astype(np.complex64)
    if return_info:
        return x, dict(bits=bits, tau=tau, f=f, phi=phi)
    return x


def synth_bursts(b0: int, count: int, taps: np.ndarray, **kw) -> np.ndarray:
    """[count][n_sym*sps] complex64, bursts b0 .. b0+count-1."""
    rows = [synth_burst(b0 + i, taps, **kw) for i in range(count)]
    return np.stack(rows, axis=0)


# ---------------------------------------------------------------------------
# host mirror of the device generator's counter-based RNG (dsp_amd/csrc/synth_kernels.hip)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def device_symbols(seed: int, b: int, n_sym: int) -> np.ndarray:
    """Quadrant indices (Re<0) + 2*(Im<0) of the symbols dsp_synth_bursts_dev transmits in burst b."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) + np.uint64(b))
        idx = np.arange(n_sym, dtype=np.uint64)
        h = _splitmix64(key + (np.uint64(1) << np.uint64(40)) + idx)
    return (h & np.uint64(3)).astype(np.uint8)


def device_params(seed: int, b: int):
    """(tau, f, phi) of burst b of the device generator."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) + np.uint64(b))
        h = _splitmix64(key + (np.uint64(2) << np.uint64(40)) + np.arange(3, dtype=np.uint64))
    u = ((h >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740992.0
    return u[0] - 0.5, (u[1] - 0.5) * 1e-3, u[2] * 2 * np.pi


def stream_symbols(seed: int, i0: int, count: int) -> np.ndarray:
    """Quadrant indices of symbols i0 .. i0+count-1 of the continuous stream dsp_synth_stream_dev
    generates (symbol i sits on sample i*sps of the transmit grid)."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed))
        idx = np.arange(i0, i0 + count, dtype=np.uint64)
        h = _splitmix64(key + (np.uint64(1) << np.uint64(40)) + idx)
    return (h & np.uint64(3)).astype(np.uint8)

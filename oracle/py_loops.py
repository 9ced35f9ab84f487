"""Pure-Python restatement of the two recurrences (test infrastructure, like the rest of oracle/):
the interpreter-bound proxy for the reference's own CPU path that SURVEY.md 8(d) asks to time on
the GPU box, where the reference itself cannot go.  Same arithmetic as oracle/dsp_oracle.c
(gardner2.py:6-100 with its nested interpolate :38-50; fll_func.py:3-29), written as plain scalar
loops over Python complex / float -- small inputs only.  tests/test_oracle_golden.py pins it to
the C oracle.
"""
import cmath
import math


def _interp(sig, idx):
    i = int(idx)
    frac = idx - i
    if i < 0 or i >= len(sig):
        return 0.0
    if i + 1 < len(sig):
        return sig[i] * (1 - frac) + sig[i + 1] * frac
    return sig[i]


def gardner(sig, sps, alpha=0.007, mu0=0.0):
    sig = list(sig)
    mu = mu0
    rec, errs, mus = [sig[0]], [], []
    half = sps // 2
    k = sps
    n = len(sig)
    while k <= n - sps:
        while mu >= sps:
            mu -= sps; k += sps
        while mu < -sps:
            mu += sps; k -= sps
        early, mid, late = k - sps + mu, k - half + mu, k + mu
        if early < 0 or late + 1 >= n:
            break
        xe, xm, xl = _interp(sig, early), _interp(sig, mid), _interp(sig, late)
        e = (xm * (xl - xe).conjugate()).real
        mu -= alpha * e
        errs.append(e); mus.append(mu)
        rec.append(_interp(sig, k + mu))
        k += sps
    return rec, errs, mus


def fll(sig, Bn=0.001, y=1):
    d = y + 1 / (4 * y)
    kp = 4 * Bn / 1000 / d
    ki = 4 * (Bn / 1000 / d) ** 2
    f = 0.0
    integ = 0.0
    out, freqs, efs = [], [], []
    r2 = math.sqrt(2)
    sgn = lambda v: (v > 0) - (v < 0)  # noqa: E731
    for i, x in enumerate(sig):
        s = x * cmath.exp(-1j * (2 * math.pi * f * i))
        dec = complex(sgn(s.real), sgn(s.imag)) / r2
        z = s * dec.conjugate()
        err = math.atan2(z.imag, z.real)
        integ += err * ki
        ef = err * kp + integ
        f += ef
        out.append(s); freqs.append(f); efs.append(ef)
    return out, f, freqs, efs


# ---- the same loops in the reference's idiom: numpy scalars and numpy ufunc calls per sample --------
# This is what makes the reference slow (each np.exp / np.angle / np.sign on a scalar costs ~1 us),
# so THIS variant is the speed proxy bench.py times next to the GPU number; the plain-Python
# variant above is the interpreter's best case.
import numpy as np  # noqa: E402


def _interp_np(sig, idx):
    i = int(idx)
    frac = idx - i
    if i < 0 or i >= len(sig):
        return 0.0
    if i + 1 < len(sig):
        return sig[i] * (1 - frac) + sig[i + 1] * frac
    return sig[i]


def gardner_np(sig, sps, alpha=0.007, mu0=0.0):
    sig = np.asarray(sig)
    mu = mu0
    rec, errs, mus = [sig[0]], [], []
    k = sps
    while k <= len(sig) - sps:
        while mu >= sps:
            mu -= sps; k += sps
        while mu < -sps:
            mu += sps; k -= sps
        early, mid, late = k - sps + mu, k - sps // 2 + mu, k + mu
        if early < 0 or late + 1 >= len(sig):
            break
        xe, xm, xl = _interp_np(sig, early), _interp_np(sig, mid), _interp_np(sig, late)
        e = np.real(xm * np.conj(xl - xe))
        mu -= alpha * e
        errs.append(e); mus.append(mu)
        rec.append(_interp_np(sig, k + mu))
        k += sps
    return np.array(rec), np.array(errs), np.array(mus)


def fll_np(sig, Bn=0.001, y=1):
    sig = np.asarray(sig)
    d = y + 1 / (4 * y)
    kp = 4 * Bn / 1000 / d
    ki = 4 * (Bn / 1000 / d) ** 2
    f = 0
    integ = 0
    out = np.zeros(len(sig), dtype=complex)
    freqs = np.zeros(len(sig))
    efs = np.zeros(len(sig))
    for i in range(len(sig)):
        s = sig[i] * np.exp(-1j * (2 * np.pi * f * i))
        out[i] = s
        dec = (np.sign(np.real(s)) + 1j * np.sign(np.imag(s))) / np.sqrt(2)
        err = np.angle(s * np.conj(dec))
        integ += err * ki
        ef = err * kp + integ
        f += ef
        freqs[i] = f; efs[i] = ef
    return out, f, freqs, efs

#!/usr/bin/env python3
"""The reference's end-to-end script test_worker.py (generate five noise-free QPSK packets, find them,
receive them) on the device path -- the same calls with `worker` replaced by dsp_amd's mirrors:

    python tools/demo_worker.py [interp=0|1]

Prints the correlation peaks, the coarse frequency estimate per packet, the match percentage the
reference's decision_direct computes (np.isclose against packet_symb: the payload alphabet is written
0.707 there while decisions are 1/sqrt(2), so only the preamble symbols can ever "match"), and the
share of decisions that fall in the transmitted symbol's quadrant under the best rotation.
With interp=0 the reference cuts every packet at peak // 2 (its own "fix this condition" remark,
worker.py:277-283) and receives noise; the mirror keeps that behaviour.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.generation import Generation            # noqa: E402   (worker.Generation)
from dsp_amd.receiver import Processing              # noqa: E402   (worker.Processing)


def main(interp=True, seed=1):
    np.random.seed(seed)
    gen = Generation(constellation="qpsk", preamble_length=300, signal_length=2000, sps=4, Frel=0.000,
                     Noise_power=0.00, N=5, Delay=30)                              # test_worker.py:4-12
    preamble, samples, symbols = gen.generate()                                    # :14
    process = Processing(signal=samples, preamble=preamble, packet_symb=symbols, sps=4, constellation="qpsk",
                         interp=interp)                                            # :23-29
    peaks, found = process.process()                                               # :31
    best = max((r for _, r in process.match_rates), default=0.0) if found else 0.0
    quad = 0.0
    if found:                                   # quadrant agreement of the last packet, best of four rotations
        tx = (symbols.real < 0) + 2 * (symbols.imag < 0)
        for k in range(4):
            z = found[-1] * np.exp(1j * np.pi / 2 * k)
            quad = max(quad, float(np.mean(((z.real < 0) + 2 * (z.imag < 0))[50:] == tx[50:len(z)])))
    return peaks, found, process.f_rel, best, quad


if __name__ == "__main__":
    interp = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True                 # test_worker.py:29
    peaks, found, f_rel, best, quad = main(interp)
    print("len(samples)", (30 + 5 * 2030) * 4)
    print("peaks", list(map(int, peaks)))
    print("packets received", len(found), "f_rel", np.array2string(np.asarray(f_rel), precision=3))
    print("last packet: decision_direct's best match %.2f %%, quadrant agreement %.2f %%" % (best, 100 * quad))

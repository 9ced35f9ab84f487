"""dsp_amd -- MI355X (gfx950) implementation of the QPSK receiver synchronisation
chain of boryaska/dsp (RRC matched filter -> Gardner timing recovery -> FLL ->
hard decisions), behind the reference's own Python function signatures.

    from dsp_amd import rrc_filter, gardner_timing_recovery, gardner_ted, fll_func

Everything numeric runs in libdsp_hip.so (hand-written HIP, see dsp_amd/csrc);
importing the package does not need a GPU, calling a device function does.
"""
from ._lib import (Context, DspError, default_context, build, lib, LIB_PATH, pinned_empty,  # noqa: F401
                   DSP_C64, DSP_C128, DSP_I16, DSP_GARDNER2, DSP_GARDNER_TED, DSP_FLL_ONLY)
from .api import (rrc_filter, fll_gains, interpolate_sample, gardner_timing_recovery,  # noqa: F401
                  gardner_ted, fll_func, gardner_batch, fll_batch, sync_chain_batch,
                  read_iq_int16, read_iq_int16_words, read_iq_float32, fir_batch, filter_samples,
                  rx_chain_batch, sync_capture_segmented, lagrange_resample,
                  apply_fractional_shift, FrequencyLockedLoop, diff_corr, find_preamble_offset,
                  find_optimal_sampling_phase, preamble_cfo, estimate_initial_frequency,
                  estimate_initial_frequency_batch, derotate, rx_bursts, constellation_points,
                  fll_slicer_batch, fll_func_constellation, decide_symbols)

__all__ = [
    "Context", "DspError", "default_context", "build", "lib", "pinned_empty",
    "rrc_filter", "fll_gains", "interpolate_sample",
    "gardner_timing_recovery", "gardner_ted", "fll_func",
    "gardner_batch", "fll_batch", "sync_chain_batch",
    "read_iq_int16", "read_iq_int16_words", "read_iq_float32", "fir_batch", "filter_samples",
    "rx_chain_batch", "sync_capture_segmented", "lagrange_resample", "apply_fractional_shift", "FrequencyLockedLoop", "diff_corr", "find_preamble_offset", "find_optimal_sampling_phase",
    "preamble_cfo", "estimate_initial_frequency", "estimate_initial_frequency_batch", "derotate", "rx_bursts",
    "constellation_points", "fll_slicer_batch", "fll_func_constellation", "decide_symbols",
]

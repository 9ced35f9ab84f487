/*
 * dsp_hip.h -- C ABI of libdsp_hip.so: the MI355X (gfx950) implementation of the
 * QPSK receiver synchronisation chain of boryaska/dsp (RRC matched filter ->
 * Gardner timing recovery -> decision-directed FLL -> hard decisions, plus the
 * raw IQ loaders that feed it).
 *
 * The reference has no FFI: its boundary is a set of Python functions imported
 * by name.  Each entry point below names the reference function
 * (file:line) whose arithmetic it reproduces; dsp_amd/api.py binds them with
 * ctypes behind the reference's own function signatures.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (DSP_E_*); the message is
 *     available from dsp_last_error() (thread-local);
 *   - complex64 is interleaved float[2], complex128 interleaved double[2];
 *     `dtype` selects the sample type of the call: DSP_C64 -> float32 outputs,
 *     DSP_C128 -> float64 outputs (what the reference's numpy code returns);
 *   - all loop arithmetic is float64 on the device in both cases;
 *   - host-pointer entry points take burst-major arrays [B][stride] (one burst
 *     per row); the library never keeps a host pointer after it returns;
 *   - *_dev entry points take device pointers in the tiled layout: element
 *     (burst b, row n) at ((b/64)*rows + n)*64 + b%64 -- 64 bursts per tile, one
 *     burst per wavefront lane.  Rows >= n_out[b] of an output hold zeros;
 *   - one dsp_ctx = one HIP device + one stream; calls on a context are
 *     serialised by the caller; use one context (one process) per GPU.
 */
#ifndef DSP_HIP_H
#define DSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsp_ctx dsp_ctx;

enum { DSP_OK = 0, DSP_E_ARG = -1, DSP_E_HIP = -2, DSP_E_CAPACITY = -3, DSP_E_NOMEM = -4 };
enum { DSP_C64 = 0, DSP_C128 = 1, DSP_I16 = 2 /* interleaved int16 I,Q: loaders / dsp_rx_chain only */ };
enum { DSP_GARDNER2 = 0,   /* gardner_timing_recovery, gardner2.py:6-100          */
       DSP_GARDNER_TED = 1 /* gardner_ted, hw1/gardner.py:22-74, read_file.py:53-104 */ };

/* ---- context ------------------------------------------------------------- */
int dsp_version(void);
int dsp_device_count(int *count);
int dsp_create(int device_id, dsp_ctx **out);
void dsp_destroy(dsp_ctx *ctx);
const char *dsp_last_error(void);
int dsp_set_stream(dsp_ctx *ctx, void *hip_stream);      /* adopt an external hipStream_t */
int dsp_synchronize(dsp_ctx *ctx);
/* what the context read from hipGetDeviceProperties when it was created: compute units and the LDS
 * bytes of one CU (= the most one workgroup can be given); either pointer may be NULL.  Utility. */
int dsp_device_info(dsp_ctx *ctx, int32_t *n_cu, int64_t *lds_per_cu);
int dsp_set_option(dsp_ctx *ctx, const char *name, int64_t value); /* tuning knobs: "ring_rows" (LDS ring rows, 0 auto),
   "lds_pad", "split" (-1 auto / 0 one wave per tile / 1 two), "queue_block" (hand-off block, 0 auto),
   "prio_period" (rotate wave priorities every N DMA blocks; -1 auto = 8 from three tiles per CU on, 0 off), "spare_wave" (two-wave kernel
   launched with idle waves so that two workgroups on a CU use four SIMDs: -1 auto = at two workgroups
   per CU, 0 never, 1 one idle wave, 2 two idle waves chosen by a per-CU ticket, 3 three LIVE waves -- timing, FLL and one
   that only feeds the LDS ring -- automatic at up to one workgroup per CU), "big_ring" (1 = default: a tile that has its CU's
   LDS to itself may use a ring of up to 128 KiB, i.e. 128 rows of complex128; 0 = 64 KiB at most), "link_threads"
   (host threads that stage transfers of >= 4 MiB from / to pageable host arrays through the context's page-locked ring,
   default 4, 0 = the runtime's own pageable copy), "spec" (1 = default: speculative branch-free segments / blocks in
   the recurrence kernels; 0 = the per-symbol paths, for A/B runs; same Gardner bits, FLL to ~1e-16), "lead_copy" (segment-parallel mode, a window's warm-up rows: 0 = default: the stream is filtered once
   into lane-major segment tiles when it is a whole number of 64-segment tiles, else the warm-up rows are stored
   by the predecessor's filter workgroup; 1 = copied by a kernel after the filter; 2 = filtered again),
   "place_ptr" (diagnostic: a device buffer of 2 x tiles uint32 that the four-wave launch fills with the HW_ID /
   XCC_ID of each tile's two live waves, 0 = off; tests/test_gpu_fullsize.py checks the SIMD placement with it),
   "fir_stream" (matched filter of <= 41 taps on complex64 / int16 input: 2 = default, persistent fir_strip_kernel;
   1 = fir_stream_kernel, one chunk per workgroup; 0 = round 1's tap-blocked fir_tile_kernel; same bits),
   "corr_window" (preamble search: 1 = default, each thread's outputs share their LDS reads through a register
   window; 0 = round 1's diff_corr_kernel; same bits),
   "batch_split" (1 = default: a recurrence batch of more than six tiles per CU is issued as equal launches
   of whole tiles; 0 = always one launch, the surplus tiles queue on the CUs; same bits) */

/* ---- device memory and timing (for *_dev pipelines and bench.py) ---------- */
int dsp_malloc(dsp_ctx *ctx, size_t bytes, void **dptr);
int dsp_free(dsp_ctx *ctx, void *dptr);
int dsp_memset(dsp_ctx *ctx, void *dptr, int value, size_t bytes);
/* Page-locked host memory for the arrays a caller hands to the host-pointer entry points below.  No
 * counterpart in the reference (its arrays are numpy's).  Those entry points take any host array; a
 * pageable one of 4 MiB or more is staged through a page-locked ring of the context by
 * "link_threads" host threads (dsp_set_option, default 4, 0 = the runtime's own pageable copy), an
 * array from here goes over the link as it is (57 GB/s against 52 staged and 17-18 GB/s for the
 * runtime's pageable copy on this pool's boxes, tools/ubench/host_link.hip).  Utility.            */
int dsp_host_alloc(size_t bytes, void **hptr);
int dsp_host_free(void *hptr);
int dsp_h2d(dsp_ctx *ctx, void *dptr, const void *hptr, size_t bytes);
int dsp_d2h(dsp_ctx *ctx, void *hptr, const void *dptr, size_t bytes);
/* plain float4 streaming copy kernel (asynchronous; bytes rounded down to 16): bench.py times it
 * beside the kernels as the same-box reference for a 1:1 read/write stream.  Utility. */
int dsp_stream_copy_dev(dsp_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int dsp_timer_start(dsp_ctx *ctx);                       /* hipEvent on the context stream */
int dsp_timer_stop(dsp_ctx *ctx, float *elapsed_ms);     /* records, synchronises, returns */

/* ---- host-side helpers (no GPU needed) ------------------------------------ */
/* rrc_filter(sps, span, alpha) -- read_file.py:5-36 (= diff_method.py:8-39).
 * Writes n_taps = span*sps//2 + 1 - (-span*sps//2) taps; returns the count when
 * h is NULL.                                                                   */
int dsp_rrc_taps(int sps, int span, double alpha, double *h, int64_t n_taps);
/* kp, ki of fll_func -- fll_func.py:5-6.                                       */
int dsp_fll_gains(double Bn, double y, double *kp, double *ki);
/* rows a tiled input buffer needs for bursts of N samples (even, >= N).        */
int64_t dsp_tile_rows(int64_t N);

/* ---- host-pointer, burst-major entry points ------------------------------- */
/* Gardner timing recovery for B independent bursts (a3-a5, a9).
 *   variant DSP_GARDNER2:    gardner_timing_recovery(x, sps, alpha, mu0)  gardner2.py:6-100
 *           DSP_GARDNER_TED: gardner_ted(x, sps, delta_mu = alpha)        hw1/gardner.py:22-74
 *   interp_half_len 0 = the reference's linear interpolation; h > 0 = (2h+1)-point
 *   Lagrange (hw1/test_lagrange_resampling.py:34-72) inside the loop (extension).
 *   lens: per-burst sample counts (NULL = all N).  Outputs [B][out_stride]
 *   (NULL = not wanted): rec has n_out[b] entries, err and mu n_out[b]-1
 *   (GARDNER2) or n_out[b] (TED).  Returns DSP_E_CAPACITY if out_stride was
 *   too small for some burst (outputs up to the capacity are valid).           */
int dsp_gardner(dsp_ctx *ctx, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                const int32_t *lens, int sps, double alpha, double mu0, int variant,
                int interp_half_len, void *rec, void *err, void *mu, int32_t *n_out,
                int64_t out_stride);

/* fll_func(signal, Bn, y) for B independent symbol streams (a6, a7) --
 * fll_func.py:3-29.  M: per-stream symbol counts (NULL = all M_max).
 * i0 / state_in / state_out (NULL = reference behaviour: index 0, zero state)
 * continue a loop at absolute index i0[b] from {curr_freq, integrator}.  A continued
 * run decides as one run over the whole stream does and follows its frequency to the
 * FLL's bar (1e-12), not to the last bit: whole blocks of symbols go through a restated
 * form of the step (same values to ~1e-16, dsp_math.h fll_step_fast), the symbols
 * around a burst's ends through the reference-order one, and a cut moves the blocks.
 * Outputs [B][out_stride]: out = signal_list, freq = curr_freq_list,
 * ef = ef_n_list, dec = quadrant (Re<0) + 2*(Im<0) of signal_list.             */
int dsp_fll(dsp_ctx *ctx, const void *x, int dtype, int64_t B, const int32_t *M, int64_t M_max,
            int64_t x_stride, double Bn, double y, const double *i0, const double *state_in,
            double *state_out, void *out, void *freq, void *ef, uint8_t *dec, int64_t out_stride);

/* The fused chain of worker.py:301-304 on matched-filtered bursts: Gardner
 * (variant) -> fll_func -> decisions in ONE kernel; the recovered symbols never
 * leave the chip.  scale: optional per-burst factor applied to every input
 * sample first (the 1/np.std of worker.py:177), NULL = 1.                      */
int dsp_sync_chain(dsp_ctx *ctx, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                   const int32_t *lens, const double *scale, int sps, double alpha, double mu0,
                   int variant, int interp_half_len, double Bn, double y,
                   void *rec, void *err, void *mu, void *sig, void *freq, void *ef, uint8_t *dec,
                   int32_t *n_out, int64_t out_stride);

/* IQ PCM loaders (a8).  dsp_load_iq_i16: read_iq_int16(path, normalize) of
 * hw1/gardner.py:77-91 on an in-memory word buffer: drops an odd trailing word,
 * int16 -> float32 I,Q -> complex64, divides by max|iq| (float32 arithmetic, as
 * numpy does for complex64) when normalize != 0 and the peak is > 0.  n_words < 2
 * -> DSP_E_ARG (the reference raises ValueError).  dsp_load_iq_f32: the idiom
 * s[::2] + 1j*s[1::2] of read_file.py:108-109 (no scaling; a host-side copy).   */
int dsp_load_iq_i16(dsp_ctx *ctx, const int16_t *iq, int64_t n_words, int normalize, float *out_c64,
                    int64_t *n_out);
int dsp_load_iq_f32(dsp_ctx *ctx, const float *iq, int64_t n_words, float *out_c64, int64_t *n_out);

/* Matched filter + normalisation (a2): y = convolve(x, h) cropped at `start`
 * (start < 0: mode='same', i.e. (n_taps-1)/2 -- worker.py:176; n_taps/2 gives
 * apply_filter of hw1/test_rrc_cascade.py:103-108), then y / np.std(y) per burst
 * when normalize_std != 0 (worker.py:177).  float64 accumulation; y has the
 * dtype of x.  std_out [B] (optional) receives np.std of the unnormalised y.   */
int dsp_fir_same_std(dsp_ctx *ctx, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                     const int32_t *lens, const double *h, int n_taps, int start, int normalize_std,
                     void *y, int64_t y_stride, double *std_out);

/* The whole receive chain of worker.py:299-304 on raw bursts: [int16 peak
 * normalisation ->] RRC 'same' -> / np.std -> Gardner -> fll_func -> decisions.
 * in_kind DSP_I16 / DSP_C64: the matched-filter output is held as complex64 in
 * HBM (float32 outputs); DSP_C128: everything float64.  The 1/std factor is
 * folded into the recurrence kernel's loads.                                   */
int dsp_rx_chain(dsp_ctx *ctx, const void *x, int in_kind, int64_t B, int64_t N, int64_t x_stride,
                 const int32_t *lens, int normalize_peak, const double *h, int n_taps,
                 int normalize_std, int sps, double alpha, double mu0, int variant,
                 int interp_half_len, double Bn, double y,
                 void *rec, void *err, void *mu, void *sig, void *freq, void *ef, uint8_t *dec,
                 int32_t *n_out, int64_t out_stride, double *std_out);

/* ---- device-resident tiled entry points ------------------------------------ */
/* burst-major device array -> tiles (zero padded); rows = dsp_tile_rows(N).     */
int dsp_tile_bursts_dev(dsp_ctx *ctx, const void *d_src, int dtype, int64_t B, int64_t N,
                        int64_t src_stride, const int32_t *d_lens, void *d_tiles, int64_t rows);
/* tiles -> burst-major device array, first ncopy rows; elem_bytes in {1,4,8,16} */
int dsp_untile_dev(dsp_ctx *ctx, const void *d_tiles, int elem_bytes, int64_t rows, int64_t B,
                   void *d_dst, int64_t dst_stride, int64_t ncopy);

typedef struct dsp_sync_args {
    const void *x;          /* tiles [ceil(B/64)][in_rows][64], dtype samples        */
    int32_t dtype;          /* DSP_C64 / DSP_C128                                    */
    int32_t variant;        /* DSP_GARDNER2 / DSP_GARDNER_TED / 2 = FLL only (1 sps) */
    int64_t B, N, in_rows;  /* bursts, samples per burst (max), rows of x            */
    const int32_t *lens;    /* [B] or NULL                                           */
    const double *scale;    /* [B] or NULL                                           */
    int32_t sps, interp_half_len;
    double alpha, mu0;
    int32_t do_fll, reserved;
    double Bn, y;
    const double *i0;       /* [B] or NULL                                           */
    const double *fll_state_in;   /* [B][2] or NULL                                  */
    double *fll_state_out;        /* [B][2] or NULL                                  */
    void *rec, *err, *mu, *sig, *freq, *ef;   /* tiles [..][out_rows][64] or NULL    */
    uint8_t *dec;
    int64_t out_rows;       /* rows of every output; the LAST row is scratch: capacity = out_rows-1 */
    int32_t *n_out;         /* [B] or NULL                                           */
    int32_t *status;        /* [B] or NULL; bit0 = capacity reached                  */
    void *pos;              /* tiles or NULL: sample position k+mu of recovered symbol r+1 in row r
                               (gardner2.py:95; float32 / float64 like mu)              */
    const double *mu0_per;  /* [B] or NULL: mu_initial per burst (DSP_GARDNER2), overrides mu0 */
} dsp_sync_args;

typedef struct dsp_fir_args {
    const void *src;        /* device, burst-major [B][src_stride] samples of in_kind      */
    int32_t in_kind;        /* DSP_C64 / DSP_C128 / DSP_I16                                 */
    int32_t start;          /* < 0: 'same'                                                  */
    int64_t B, N, src_stride;
    const int32_t *lens;    /* device [B] or NULL                                           */
    const double *h;        /* device taps [n_taps]                                         */
    int32_t n_taps, reserved;
    const float *in_scale;  /* device scalar multiplied into int16 samples (float32) or NULL */
    void *tiles;            /* out: tiles [..][rows][64] complex64 (complex128 for C128), or NULL */
    int64_t rows;
    const double *scale;    /* device [B] factor applied to the outputs, or NULL            */
    double *std_out;        /* device [B] np.std of the unscaled output, or NULL            */
    double *inv_std_out;    /* device [B] 1/std, or NULL                                    */
    void *workspace;        /* device, dsp_fir_workspace_bytes() bytes (needed for std)     */
    int64_t workspace_bytes;
} dsp_fir_args;

int64_t dsp_fir_workspace_bytes(int in_kind, int64_t B, int64_t N, int n_taps);
/* FIR from burst-major samples straight into tiles (+ per-burst std); asynchronous.
 * Returns DSP_E_ARG if the filter is too long for the LDS-tiled kernel (n_taps > ~80). */
int dsp_fir_tiles_dev(dsp_ctx *ctx, const dsp_fir_args *args);

/* dsp_fir_tiles_dev without any allocation -- NO COUNTERPART IN THE REFERENCE.  For callers that cannot
 * call hipMalloc or bring a workspace: tight serving loops, harnesses that forbid allocation, code
 * that must not change the device's memory map (free memory is the same before and after a call:
 * tests/test_zz_no_alloc.py).  Not exercised under hipStreamBeginCapture: besides its kernel launches
 * the call makes hipSetDevice and hipFuncSetAttribute calls.  Same arguments, same kernels and bit-identical results, but args->workspace
 * / workspace_bytes are ignored: the partial statistics and the padded taps live in a static device
 * array of dsp_static_scratch_bytes() = 16 MiB that every GPU the library runs on holds until the
 * process ends.  A call that needs more -- 512 + B * (ceil(N / chunk) + 8) * 32 bytes when std_out /
 * inv_std_out is wanted; B = 4096 bursts of 8192 samples need 10.7 MB -- returns DSP_E_ARG and launches
 * nothing.  The call makes no hipMalloc / hipFree (nor does anything it calls) and does not
 * synchronise.  All calls on one GPU share the array: calls that overlap in time there (two contexts,
 * two streams) race; calls on one context's stream are ordered and safe.  Nothing needs clearing: a
 * call writes every record it later reads.                                                        */
int dsp_fir_tiles_static_dev(dsp_ctx *ctx, const dsp_fir_args *args);
int64_t dsp_static_scratch_bytes(void);

/* Differential-correlation preamble search -- the sliding loop of
 * find_preamble_offset, diff_method.py:118-148 (= worker.py:183-245): for phase in 0..sps-1,
 * s = x[phase::sps], d = s[1:]*conj(s[:-1]), conv[phase][k] = sum(d[k:k+L]*conj(d_pre)) /
 * np.std(d[k:k+L]) (0 where the window's std is 0), L = n_pre - 1, complex64 like :137.
 * x: n samples of dtype (host); preamble: n_pre complex128 symbols; conv: [sps][k_stride]
 * complex64, zero beyond the last full window of each phase (k_stride >= len(x[phase::sps]) - 1
 * holds every phase).  The arg-max / peak picking of :151-160 stays with the caller.        */
int dsp_diff_corr(dsp_ctx *ctx, const void *x, int dtype, int64_t n, const double *preamble_c128, int n_pre,
                  int sps, float *conv_c64, int64_t k_stride);

/* fll_func with the constellation switch of worker.Processing.fll_func (worker.py:18-78; SURVEY.md 8f
 * rank 4): the loop of fll_func.py with the slicer chosen by kind -- 0 bpsk sign(Re) + 0j (:36),
 * 1 qpsk (:38), 2 nearest point of points_c128[n_points] by np.abs distance, first of equally near
 * points wins (:40-66; 8psk / 16psk / 16apsk tables are built by the caller exactly as the reference
 * does).  x: [B][x_stride] host bursts at 1 sample/symbol.  Outputs [B][out_stride] in x's precision
 * (out complex, freq / ef real; any may be NULL), dec uint8 decision index (table position; qpsk
 * quadrant (Re<0)+2(Im<0); bpsk 0 / 1 / 2 for Re > 0 / < 0 / == 0), curr_freq [B] the final frequency.
 * Bn = 0 freezes the loop: dec is then decision_direct's slicer alone (worker.py:80-123).           */
int dsp_fll_slicer(dsp_ctx *ctx, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                   const int32_t *lens, int kind, const double *points_c128, int n_points, double Bn, double y,
                   void *out, void *freq, void *ef, uint8_t *dec, int64_t out_stride, double *curr_freq);

/* Coarse carrier-frequency estimators.
 *
 * dsp_cfo_preamble -- the three preamble-aided estimates of diff_method.py:267-292 (the first is what
 * worker.Processing.process uses, worker.py:288-294).  For burst b: s = x[offsets[b] :: sps][:n_pre],
 * p = s * conj(preamble), phases = np.unwrap(np.angle(p));
 *   f_rel[b][0] = mean(diff(phases))                     / (2 pi sps)
 *   f_rel[b][1] = polyfit(arange, phases, 1)[0]          / (2 pi sps)      (closed-form LSQ slope)
 *   f_rel[b][2] = angle(mean(p[1:] * conj(p[:-1])))      / (2 pi sps)
 * x: n host samples of dtype (DSP_C64 is computed in float64 on the upcast samples); every burst must
 * hold (n_pre-1)*sps+1 samples from its offset.  All three are float64, cycles per sample.          */
int dsp_cfo_preamble(dsp_ctx *ctx, const void *x, int dtype, int64_t n, const int64_t *offsets, int64_t B, int sps,
                     const double *preamble_c128, int n_pre, double *f_rel /* [B][3] */);

/* estimate_initial_frequency(signal, num_samples), estimate_freq.py:4-58, one call per row of a
 * [B][x_stride] host block: n = min(num_samples, len); z = s[:n]**4; freq = angle(mean(z[1:] *
 * conj(z[:-1]))) / (8 pi); 0.0 where n < 2.  lens (or NULL = N) gives each row's length.            */
int dsp_cfo_fourth_power(dsp_ctx *ctx, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                         const int32_t *lens, int64_t num_samples, double *freq /* [B] */);

/* Cut + derotate -- worker.py:287,298 / diff_method.py:296-298: out[b][i] = x[offsets[b] + i] *
 * exp(-1j * 2 pi * f_rel[b] * (i + index0[b])), i < n_cut, complex128 [B][out_stride].
 * offsets NULL = every cut starts at 0; index0 NULL = offsets (the worker's absolute-index phase).  */
int dsp_derotate(dsp_ctx *ctx, const void *x, int dtype, int64_t n, const int64_t *offsets, int64_t B,
                 int64_t n_cut, const double *f_rel, const int64_t *index0, double *out_c128, int64_t out_stride);

/* Device-resident front end of dsp_rx_bursts (asynchronous on the context's stream; every pointer is
 * device memory): frequency per burst estimated from the preamble (cfo_method 1..3: d_f is [B][3] and
 * receives all three estimates) or taken from d_f [B] (cfo_method 0), then out[b][i] = x[off_b + i] *
 * exp(-1j 2 pi f_b (i + off_b)), complex128 [B][n_pkt] -- ready for dsp_fir_tiles_dev (DSP_C128).
 * The offsets are the caller's responsibility here: 0 <= off_b and off_b + n_pkt <= n.            */
int dsp_cfo_derotate_dev(dsp_ctx *ctx, const void *d_x, int dtype, int64_t n, const int64_t *d_offsets, int64_t B,
                         int64_t n_pkt, int sps, const double *d_preamble_c128, int n_pre, int cfo_method,
                         double *d_f, double *d_out_c128);

/* The same front end with the matched filter behind it, as ONE pass over the signal (worker.py:287-299:
 * cut, f_rel, derotation, filter_samples): the estimator runs as above, then every sample is cut out of the
 * signal, derotated and filtered on its way through LDS -- the derotated bursts never exist in memory.
 * Results are the bits of dsp_cfo_derotate_dev followed by dsp_fir_tiles_dev (DSP_C128) on its output.
 * d_h: n_taps <= 41 float64 taps on the device ("same" alignment, start = (n_taps - 1) / 2).
 * d_tiles_c128: complex128 tiles, `rows` >= n_pkt rows per tile (dsp_tile_rows), or NULL (statistics only).
 * d_std / d_inv_std: [B] np.std of each filtered burst and its reciprocal, or NULL; they need `workspace`
 * of dsp_rx_front_workspace_bytes(B, n_pkt) bytes.  Longer filters: DSP_E_ARG (use the two calls above).
 * Asynchronous on the context's stream.                                                              */
int64_t dsp_rx_front_workspace_bytes(int64_t B, int64_t n_pkt);
int dsp_rx_front_dev(dsp_ctx *ctx, const void *d_x, int dtype, int64_t n, const int64_t *d_offsets, int64_t B,
                     int64_t n_pkt, int sps, const double *d_preamble_c128, int n_pre, int cfo_method, double *d_f,
                     const double *d_h, int n_taps, void *d_tiles_c128, int64_t rows, double *d_std,
                     double *d_inv_std, void *workspace, int64_t workspace_bytes);

/* The per-burst body of worker.Processing.process (worker.py:284-304) for B bursts of one signal in
 * one call: coarse CFO from the preamble (cfo_method 1..3 as above; 0 = take f_rel_in), cut of n_pkt
 * samples at offsets[b], derotation with the absolute sample index, RRC 'same' convolution + /np.std
 * (normalize_std), gardner_timing_recovery, fll_func.  The signal is uploaded once and everything
 * between stays on the device.  Outputs as dsp_rx_chain (complex128 / float64, any may be NULL);
 * f_rel_out [B] receives the estimate that was applied.                                              */
int dsp_rx_bursts(dsp_ctx *ctx, const void *x, int dtype, int64_t n, const int64_t *offsets, int64_t B, int64_t n_pkt,
                  const double *preamble_c128, int n_pre, int cfo_method, const double *f_rel_in,
                  const double *h, int n_taps, int normalize_std, int sps, double alpha, double mu0, int variant,
                  int interp_half_len, double Bn, double y,
                  void *rec, void *err, void *mu, void *sig, void *freq, void *ef, uint8_t *dec,
                  int32_t *n_out, int64_t out_stride, double *f_rel_out, double *std_out);

/* Facade for the reference's class FrequencyLockedLoop(detector_method, Kp, Ki, freq_limit)
 * .process_signal(x) (used by fll_real_signal.py:66-74,209-216, f_rel.py:123-132).  PARITY
 * UNPINNED: fll.py itself is absent from the reference, so the loop is this library's own
 * definition (accumulating-phase NCO, detector 0 cross_product / 1 atan2 / 2 decision_directed,
 * PI filter with anti-windup, frequency clamp; dsp_amd/csrc/nco_kernels.hip).  B bursts [B][x_stride]
 * of dtype; outputs [B][out_stride]: corrected samples (dtype), history errors / freq_estimates /
 * phases (real of the same precision), any may be NULL; freq_estimate [B] float64.        */
int dsp_fll_class(dsp_ctx *ctx, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                  const int32_t *lens, int detector, double Kp, double Ki, double freq_limit,
                  void *out, void *errors, void *freqs, void *phases, int64_t out_stride, double *freq_estimate);

/* lagrange_resample(signal, positions, half_len=4) -- hw1/test_lagrange_resampling.py:34-72
 * ((2h+1)-point barycentric Lagrange, neighbour indices clipped to [0, n-1] :58-59, exact-node
 * shortcut :62-66).  x: n samples of dtype DSP_C64 / DSP_C128 (a real signal is passed with zero
 * imaginary parts); out: P complex128 values.  Host pointers.                              */
int dsp_lagrange_resample(dsp_ctx *ctx, const void *x, int dtype, int64_t n, const double *positions, int64_t P,
                          int half_len, double *out_c128);

/* ---- segment-parallel mode for ONE long capture (north-star (c)) ----------
 * No counterpart in the reference, whose loops are strictly sequential (gardner2.py:63-98,
 * fll_func.py:14-26).  The capture is cut into segments of seg_len samples; each segment is one
 * lane of the batched recurrence kernel, started `warmup` symbols before its seam so that the
 * loops have converged when the segment's own share begins, and the shares are stitched back:
 * positions k+mu decide which symbols belong to which segment (no gap, no duplicate), the QPSK
 * quadrant ambiguity between neighbouring segments is resolved over the symbols both decoded.
 * The FLL's sample index (fll_func.py:16) restarts at every window start: with the reference's
 * 2*pi*f*i phase the loop gain grows with i and the loop is unstable for i > 1/(pi*kp), so one
 * run over a long capture is not a usable definition.
 *   x        DSP_I16 interleaved int16 I,Q (peak-normalised like read_iq_int16 when normalize)
 *            or DSP_C64 complex64 samples
 *   h        matched-filter taps: np.convolve(x, h, 'same') / np.std over the WHOLE capture
 *            (worker.py:173-178), or NULL = x is already filtered.  The filter output is held as
 *            complex64 and 1/std is applied by the recurrence kernel (like dsp_rx_chain)
 *   outputs  one entry per symbol of the stitched stream (capacity entries available):
 *            dec quadrant index, mu (float32, may be NULL), freq (float32, may be NULL);
 *            *n_symbols receives the count; seg_info (may be NULL) [S][4] int32:
 *            first kept row, end row, rotation applied, rows produced, per segment.
 * dsp_sync_capture_segmented takes host pointers, ..._dev device pointers (n_symbols and
 * seg_info stay host pointers; the call synchronises).                                        */
typedef struct dsp_capture_args {
    const void *x;
    int32_t in_kind, normalize;
    int64_t n_samples;
    const double *h;
    int32_t n_taps, sps;
    double alpha, mu0, Bn, y;
    int64_t seg_len, warmup;
    uint8_t *dec;
    float *mu, *freq;
    int64_t capacity;
    int64_t *n_symbols;
    int32_t *seg_info;
    int32_t acquire;        /* 1: every window starts its timing loop at the sampling phase (0..sps-1)
                               whose strobes carry the most energy over its first DSP_ACQUIRE_SYMBOLS
                               symbols (a cold start half a symbol off can hang on the Gardner S-curve's
                               unstable zero for longer than any warm-up); 2: at the fractional phase of
                               the same energies' symbol-rate line over DSP_ACQUIRE_FRAC_SYMBOLS symbols,
                               atan2(sum E_p sin(2 pi p / sps), sum E_p cos(2 pi p / sps)) sps / (2 pi) in
                               [0, sps), rounded to 1/1024 sample -- some 0.01 sample rms from the lock
                               point instead of up to half a sample, so a 2048-symbol warm-up leaves what
                               a 4096-symbol one leaves after 1; 0: every window starts at mu0          */
    int32_t reserved;
} dsp_capture_args;
#define DSP_ACQUIRE_SYMBOLS 256
#define DSP_ACQUIRE_FRAC_SYMBOLS 512
/* The segment-parallel calls keep their large device work buffers (window tiles, per-window output
 * tiles) in the context between calls; dsp_trim releases them (dsp_destroy does too).  The host-pointer
 * entry points likewise keep their device blocks (<= 1 GiB per process) for the next call: dsp_trim returns
 * those of the context's device as well.                                                            */
int dsp_trim(dsp_ctx *ctx);
/* The argument checks of the two calls below as a function of its own (no device needed): DSP_OK, or
 * DSP_E_ARG with the reason in dsp_last_error().  Limits: sps 1..64; seg_len a multiple of sps and
 * >= 16 symbols; at most 2^24 segments; a window (seg_len + warmup*sps + 4*sps + 8 samples) of at
 * most 2^23 samples -- symbol positions inside a window are float32 and the stitch decides on half a
 * symbol.  x / dec are only checked for NULL.                                                     */
int dsp_capture_validate(const dsp_capture_args *args);
int dsp_sync_capture_segmented(dsp_ctx *ctx, const dsp_capture_args *args);
int dsp_sync_capture_segmented_dev(dsp_ctx *ctx, const dsp_capture_args *args);
int64_t dsp_capture_segments(int64_t n_samples, int64_t seg_len);   /* S = ceil(n / seg_len) */

/* Synthetic QPSK bursts generated on the device (benchmarks / full-size tests; recipe of
 * SURVEY.md 8d after worker.py:339-427, counter-based RNG): bursts b0 .. b0+B-1 of n_sym*sps
 * complex64 samples, burst-major, at the TX output + channel (before the RX matched filter).
 * d_taps: device TX RRC taps.  No counterpart function in the reference (its generator is
 * numpy-random based); asynchronous; B <= 65535 per call.                          */
int dsp_synth_bursts_dev(dsp_ctx *ctx, uint64_t seed, int64_t b0, int64_t B, int n_sym, int sps,
                         const double *d_taps, int n_taps, double esn0_db, void *d_out_c64);

/* One continuous synthetic QPSK stream on the device (config 5: a long capture for the
 * segment-parallel mode): same recipe, timing offset tau0 + drift*n samples, carrier offset cfo
 * cycles/sample; out_kind DSP_I16 (interleaved int16 I,Q, `amplitude` counts per unit) or
 * DSP_C64.  Symbol i is splitmix-derived from seed (dsp_amd/synth.py stream_symbols).
 * No counterpart in the reference; asynchronous.                                  */
int dsp_synth_stream_dev(dsp_ctx *ctx, uint64_t seed, int64_t n_samples, int sps, const double *d_taps,
                         int n_taps, double esn0_db, double cfo, double tau0, double drift, int out_kind,
                         double amplitude, void *d_out);

/* Launches the fused recurrence kernel on the context stream (asynchronous).    */
int dsp_sync_tiles_dev(dsp_ctx *ctx, const dsp_sync_args *args);

#ifdef __cplusplus
}
#endif
#endif /* DSP_HIP_H */

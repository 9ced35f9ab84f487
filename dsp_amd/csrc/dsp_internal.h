// dsp_internal.h -- declarations shared between the C-ABI layer (dsp_api.hip)
// and the kernel translation units.  Not installed; the public surface is
// include/dsp_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// ---------------------------------------------------------------------------
// Device data layout ("tiles"): bursts are grouped 64 to a tile, one burst per
// wavefront lane.  Element (burst b, row n) of a tiled array lives at
//     ((b / 64) * rows + n) * 64 + (b % 64)
// so a wavefront that walks its tile row by row reads/writes one contiguous
// 64-element row per step (512 B for complex64) and a tile is one contiguous
// stream of rows * 64 elements.  Input tiles have `rows` even, padded with zeros.
// ---------------------------------------------------------------------------

enum { DSP_VARIANT_GARDNER2 = 0, DSP_VARIANT_TED = 1, DSP_VARIANT_NONE = 2 };

struct SyncParams {
    // input
    const void *x;            // tiled [ntiles][in_rows][64] complex64 or complex128
    const int *lens;          // [B] valid samples per burst, or NULL -> len_all
    const double *scale;      // [B] per-burst input scale (1/std), or NULL -> 1.0
    int B, in_rows, len_all, sps;
    int x_tile_rows;          // rows between row 0 of consecutive tiles (== in_rows unless the tiles' windows overlap)
    const void *x_tile0;      // tile 0's rows if they live elsewhere, or NULL
    int idle_pass;            // bursts of length 0 may run as passenger lanes: nothing reads their output columns
    // Gardner
    double alpha;             // gardner2: alpha; ted: delta_mu
    double mu0;
    const double *mu0_per;    // [B] initial mu per burst (gardner2 variant), or NULL -> mu0
    int half_len;             // 0 = linear interpolation
    // FLL
    double kp, ki;
    const double *i0;         // [B] absolute index of the first FLL symbol, or NULL -> 0
    const double *fll_in;     // [B][2] {curr_freq, integrator} start state, or NULL -> 0
    double *fll_out;          // [B][2] end state, or NULL
    // outputs, tiled [ntiles][out_rows][64]; any may be NULL
    void *rec, *terr, *mu, *sig, *freq, *ef;
    uint8_t *dec;
    void *pos;                // sample position k + mu of every recovered symbol (row r = symbol r + 1), or NULL
    int out_rows;
    int *n_out;               // [B] number of symbols produced
    int *status;              // [B] bit0: output capacity reached before the loop ended
    // LDS ring schedule (host-computed, see sync_plan())
    int use_ring, ring_rows, T, C, nwait;
    int TQ;                   // > 0: split kernel, queue block length (iterations)
    int spare_wave;           // split kernel: 1 = 3 waves per tile, the middle one idle; 2 = 4 waves, two chosen by CU ticket
    unsigned *cu_tickets;     // [4096] per-CU workgroup counters (spare_wave == 2)
    unsigned *place;          // diagnostic, or NULL: [ntiles][2] where the two live waves of the four-wave launch
                              // run: bit 31 set, XCC_ID in bits 19:16, HW_ID[15:0] (wave, SIMD 5:4, CU 11:8, SH 12, SE 15:13)
    int guard_rows;           // LDS rows behind the ring that mirror its first rows (0 or 8): a lane's rows of one symbol
                              // are then contiguous in LDS even across the ring's wrap (speculative segments)
    int prio_role;            // split kernel: fixed wave priority by role, timing wave (the longer chain) above the FLL wave
    int dma_fll;              // split kernel: the FLL wave (which idles a third of its time) issues the ring's LDS-DMA,
                              // waits for it and mirrors the guard rows; needs T % TQ == 0 (sync_kernels.hip)
    int spec;                 // split kernel, gardner2 with the linear interpolator: speculative segments (sync_kernels.hip)
    int prio_period;          // split kernel: rotate the wave priority every this many DMA blocks (power of 2, 0 = off)
#ifdef DSP_TRACE
    unsigned long long *trace; // diagnostic build only: [tile][role][20] = start, end (s_memtime), HW_ID, XCC_ID, 16 segment sums
    int trace_stamps;          // also stamp the loop segments (slows the kernel: read shares, not totals)
#endif
};

struct SyncPlan { int use_ring, ring_rows, T, C, nwait, lds_bytes; };

// host helpers implemented next to the kernels
SyncPlan sync_plan(int elem_bytes, int sps, int half_len, int variant, int n_store, int ring_rows_req);
hipError_t launch_sync(const SyncParams &P, bool c128, int variant, bool fll, size_t lds_bytes,
                       hipStream_t stream);
int sync_split_tiles_per_cu(bool c128, int variant, bool scaled);

// tiling: burst-major [B][ld] <-> tiles
hipError_t launch_tile_from_bursts(const void *src, int64_t ld, const int *lens, int len_all, int B,
                                   void *dst, int rows, int elem_bytes, hipStream_t stream,
                                   const int64_t *starts = nullptr);
hipError_t launch_bursts_from_tiles(const void *src, int rows, int B, void *dst, int64_t ld,
                                    int ncopy, int elem_bytes, hipStream_t stream);

// ---------------------------------------------------------------------------
// FIR / loaders (fir_kernels.hip)
enum { DSP_IN_C64 = 0, DSP_IN_C128 = 1, DSP_IN_I16 = 2 };

struct FirParams {
    const void *src;          // burst-major [B][ld] samples: int16 pairs, complex64 or complex128
    int64_t ld;               // samples between consecutive bursts (may be < N: overlapping views)
    const int *lens;          // [B] or NULL -> len_all
    int len_all, B;
    // bursts cut out of one signal: burst b, sample n is src[src_off[b] + n] (NULL -> src[b * ld + n]); outside
    // [0, len) the filter sees zeros (np.convolve on the cut burst), not the signal's neighbours
    const int64_t *src_off;
    // cut + derotation fused into the staging loads (worker.py:298-299): sample n of burst b is multiplied by
    // exp(-1j * fl(fl(2 pi * f_b) * (n + i0_b))) in float64 as it goes to LDS, f_b = derot_f[b * derot_fstride],
    // i0_b = derot_i0 ? derot_i0[b] : src_off[b]; the filter then runs on complex128 staging whatever the
    // input type.  NULL -> no derotation.  fir_stream_kernel / fir_strip_kernel only (<= 41 taps).
    const double *derot_f;
    int derot_fstride;
    const int64_t *derot_i0;
    // windows of ONE long stream (segment-parallel mode): burst b, sample n is stream sample
    // starts[b] + n, present iff 0 <= starts[b] + n < n_total (so the filter sees the true
    // neighbours across window edges and zeros only beyond the stream's own ends)
    const int64_t *starts;    // [B] or NULL -> src + b * ld, zeros outside [0, len)
    int64_t n_total;
    const int *stat_lo, *stat_hi;   // [B] window-local range that counts in the statistics, or NULL -> [0, len)
    const double *h;          // device taps [M]
    int M, start;             // y[n] = sum_j h[j] x[n + start - j]
    const float *in_scale;    // device scalar applied to int16 samples in float32, or NULL
    void *dst;                // tiles [ntiles][rows][64] (complex64 / complex128), or NULL: stats only
    int rows;
    const double *scale;      // [B] factor applied to the outputs (1/std), or NULL
    double *stats;            // [B][nchunk][4] partial statistics, or NULL
    int nchunk, CH, W, stride;
    int skip_rows;            // tiles > 0 skip the chunks that end at or before this output row (0 = none)
    int waves;                // waves per workgroup: 0 = auto (8 where the chunk divides), 4 = always 4
    int stream;               // set by fir_plan: 1 = fir_stream_kernel (<= 41 taps, float32 staging)
    int strip_ok;             // in: launch_fir_tile may use the persistent fir_strip_kernel
    int nrec_extra;           // caller's note: records appended behind the B * nrec of this pass (capture mode)
    int stream_mode;          // in: the bursts are the consecutive segments of one stream of n_total samples,
                              //     lane l of tile t = segment l * ntiles + t, rows == len_all == ld
                              //     (fir_strip_kernel STREAM; cleared by fir_stream_mode_plan if it cannot serve)
    int n_cu;                 // in: compute units of the device (strip planning)
    // set by launch_fir_tile:
    int cps;                  // chunks per strip of fir_strip_kernel, 0 = chunk-per-workgroup kernels only
    int c_int0, c_int1;       // interior chunks [c_int0, c_int1) the strips cover (of the first strip_tiles tiles)
    int strip_tiles;          // full tiles
    int nrec;                 // statistics records per burst in the workspace (stride)
    int rec_full;             // records written for the bursts of the first strip_tiles tiles (the others: nchunk)
    int edge, rec0, tile0;    // fir_stream_kernel beside the strips: edge-chunk mapping, first record, first tile
    double *h_pad;            // device scratch for the stream kernel's zero-padded taps (>= 48 doubles)
    int feed_lead;            // 1: every window also stores its rows [shift, shift + skip_rows) as rows [0, skip_rows)
                              //    of its successor (shift = starts[b+1] - starts[b]; successors in tiles > 0 only)
};

bool fir_plan(int in_kind, int M, int n_out, FirParams &P, bool allow_stream = true, bool c128_stream = false);
bool fir_try_c128_strips(FirParams &P);       // complex128: switch a filled P to the strip kernel's plan if it applies
#define DSP_FIR_MT 41          // taps the stream / strip kernels are unrolled for (fir_kernels.hip)
hipError_t launch_fir_tile(FirParams &P, int in_kind, hipStream_t s);
// the fused form (P.derot_f, P.src_off set; in_kind DSP_IN_C64 or DSP_IN_C128; output complex128 tiles): plans
// the stream / strip kernels over float64 staging; false -> the shape needs the unfused path
bool fir_plan_derot(FirParams &P, int in_kind);
hipError_t launch_fir_derot(FirParams &P, int in_kind, hipStream_t s);
// stream_mode requested: can fir_strip_kernel serve these parameters?  (false: clears stream_mode; the
// caller falls back to the overlapping-window pass)
bool fir_stream_mode_plan(FirParams &P, int in_kind);
hipError_t launch_fir_stats_finish(const FirParams &P, double *std_out, double *inv_out, hipStream_t s);
hipError_t launch_fir_stats_global(const double *stats, int64_t count, int B, double *std_out, double *inv_fill,
                                   double *partial, int n_partial, hipStream_t s);
hipError_t launch_fir_direct(const FirParams &P, int in_kind, void *dst, int64_t dst_ld, int n_out, hipStream_t s);
hipError_t launch_burst_std(const void *y, int c128, int64_t ld, const int *lens, int len_all, int B,
                            double *std_out, double *inv_out, hipStream_t s);
hipError_t launch_burst_scale(void *y, int c128, int64_t ld, const int *lens, int len_all, int B, int n_max,
                              const double *scale, hipStream_t s);
hipError_t launch_burst_cast(const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int B, int n, hipStream_t s);
hipError_t launch_iq_i16(const void *raw, int64_t n, int normalize, unsigned *peak2, float *iscale, void *out,
                         hipStream_t s);

// synthetic burst generator (synth_kernels.hip)
hipError_t launch_synth(uint64_t seed, int64_t b0, int B, int n_sym, int sps, const double *taps, int M,
                        float sigma, void *out, hipStream_t s);

hipError_t launch_synth_stream(uint64_t seed, int64_t n_total, int sps, const double *taps, int M, float sigma,
                               double cfo, double tau0, double drift, int i16, float amp, void *out, hipStream_t s);

// ---------------------------------------------------------------------------
// segment-parallel mode for one long capture (segment_kernels.hip)
struct SegParams {
    int S;                    // segments (= lanes of the recurrence kernel)
    int sps, out_rows;
    int64_t n_total, seg_len; // stream samples, samples per segment
    const int64_t *starts;    // [S] stream sample of each window's first sample
    const int *n_out;         // [S] symbols produced per window (rows of the tiled outputs)
    const void *mu;           // tiled float32 mu history of the windows
    const void *pos;          // tiled float32 window-local position of every symbol (row r = symbol r + 1)
    const uint8_t *dec;       // tiled decisions
    const void *freq;         // tiled float32 frequency (may be NULL)
    int *j0, *j1;             // [S] kept rows [j0, j1) of each window
    double *p_last;           // [S] stream position of row j1 - 1
    int *rot;                 // [S] quadrant rotation of each window relative to its predecessor, then cumulative
    int64_t *off;             // [S + 1] output offset of each window's first kept symbol
    // Every per-window array above is indexed by the window's SLOT = tile * 64 + lane in the recurrence
    // kernel.  map_tiles == 0: slot == window; > 0 (lane-major, S == 64 * map_tiles): window s sits in
    // lane s / map_tiles of tile s % map_tiles, so that a lane's windows in consecutive tiles are
    // consecutive segments of the stream.
    int map_tiles;
    int n_slots;              // slots the arrays hold (>= S; 64 * map_tiles when mapped: the slots behind the
                              // last window are empty windows, lens == 0); off[n_slots] receives the total
};
static inline __host__ __device__ int seg_slot(int s, int map_tiles)
{
    return map_tiles > 0 ? (s % map_tiles) * 64 + s / map_tiles : s;
}
static inline __host__ __device__ int seg_of_slot(int slot, int map_tiles)
{
    return map_tiles > 0 ? (slot & 63) * map_tiles + (slot >> 6) : slot;
}
// (map_tiles > 0: the arrays hold 64 * map_tiles slots; those of the windows >= S get length 0)
hipError_t launch_seg_windows(int S, int64_t n_total, int64_t seg_len, int64_t lead, int64_t tail,
                              int64_t *starts, int *lens, int *own_lo, int *own_hi, hipStream_t s, int map_tiles = 0);
// Lane-major segment tiles (fir_strip_kernel STREAM: [map_tiles][seg_len][64] at `tiles`): the windows of
// tile 0 gathered into a tile of their own (`tile0`, win_rows rows: their warm-up rows are the ends of
// other lanes' last segments, and the stream's first window has none) and the pad_rows rows behind the
// last tile (the starts of the next lanes' first segments; zeros past the stream's end)
hipError_t launch_seg_edge_tiles(void *tiles, int map_tiles, int64_t seg_len, int64_t n_total, const int64_t *starts,
                                 const int *lens, void *tile0, int win_rows, int pad_rows, hipStream_t s);
hipError_t launch_seg_lead_copy(void *tiles, int rows, const int64_t *starts, int S, int lead, hipStream_t s);
hipError_t launch_seg_phase(const void *tiles, int rows, const int *lens, int S, int sps, int n_sym, double *mu0,
                            hipStream_t s, int x_tile_rows = 0, const void *x_tile0 = nullptr, int frac = 0);
hipError_t launch_seg_stitch(const SegParams &P, int overlap_vote, hipStream_t s);
hipError_t launch_seg_gather(const SegParams &P, uint8_t *dec_out, float *mu_out, float *freq_out,
                             int64_t capacity, hipStream_t s);
hipError_t launch_lagrange_resample(const void *x, int c128, int len, const double *pos, int64_t P, int half_len,
                                    void *out, hipStream_t s);
hipError_t launch_stream_copy(const void *src, void *dst, size_t bytes, hipStream_t s);
hipError_t launch_fll_class(const void *x, int c128, const int *lens, int len_all, int B, int rows, int detector,
                            double Kp, double Ki, double flim, void *out, void *errs, void *freqs, void *phases,
                            double *freq_final, hipStream_t s);
hipError_t launch_fll_slicer(const void *x, int c128, const int *lens, int len_all, int B, int rows, int kind,
                             const void *points, int n_points, double kp, double ki, void *out, void *freqs,
                             void *efs, unsigned char *dec, double *freq_final, hipStream_t s);
// cfo_kernels.hip
hipError_t launch_cfo_preamble(const void *x, bool c128, const int64_t *offsets, int B, int sps, const void *pre,
                               int L, double *out /* [B][3] */, hipStream_t s);
hipError_t launch_cfo_fourth_power(const void *x, bool c128, int64_t stride, const int *lens, int len_all, int B,
                                   int64_t num_samples, double *out, hipStream_t s);
hipError_t launch_derotate_cut(const void *x, bool c128, const int64_t *offsets, const int64_t *index0,
                               const double *f, int f_stride, int B, int64_t n_cut, void *out, int64_t out_stride,
                               hipStream_t s);
hipError_t launch_diff_corr(const void *x, int c128, int64_t n, int sps, const void *dpre, int L, void *conv,
                            int64_t kstride, int window, hipStream_t s);

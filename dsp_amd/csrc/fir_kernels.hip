// fir_kernels.hip -- matched filter (real taps x complex samples, 'same' /
// cropped full convolution), np.std normalisation, and the raw IQ loaders.
//
//   y[n] = sum_j h[j] x[n + start - j]     np.convolve(x, h, 'same'): start = (M-1)//2
//                                           (worker.py:176); apply_filter: start = M//2
//                                           (hw1/test_rrc_cascade.py:103-108)
//   y / np.std(y)                           worker.py:177 (population std of a complex vector)
//
// fir_tile_kernel: one workgroup = one tile (64 bursts) x CH output samples.
// The input window of every burst is staged in LDS with coalesced loads along
// the burst (complex64 HBM loads, int16 pairs or complex128), taps are read
// from LDS as broadcasts, each lane (= burst) then slides an OG-output register
// block over the taps in float64.  Outputs leave as full 64-lane rows of the
// tiled layout (lane = burst), so no output transpose is needed.  Per (burst,
// chunk) partial statistics (count, mean, M2) are merged by fir_stats_finish
// with Chan's formula -> 1/std per burst, applied by the consumer kernels.
#include "dsp_internal.h"

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));
typedef short s2v __attribute__((ext_vector_type(2)));

template <typename IN> struct Stage;                     // LDS staging type of an input type
template <> struct Stage<s2v> { typedef f2v T; };
template <> struct Stage<f2v> { typedef f2v T; };
template <> struct Stage<d2v> { typedef d2v T; };

__device__ __forceinline__ f2v to_stage(s2v v, float isc) { return f2v{(float)v.x * isc, (float)v.y * isc}; }
__device__ __forceinline__ f2v to_stage(f2v v, float) { return v; }
__device__ __forceinline__ d2v to_stage(d2v v, float) { return v; }

// XF = 1: cut + derotation fused into the staging (FirParams::derot_f): float64 staging whatever the input
// type, every sample multiplied by exp(-1j theta) on its way to LDS -- the arithmetic of derotate_cut_kernel
// (cfo_kernels.hip) expression for expression, so the filtered bursts are the same bits as filtering its output.
template <typename IN, int XF> struct StageT { typedef typename Stage<IN>::T T; };
template <typename IN> struct StageT<IN, 1> { typedef d2v T; };
template <typename IN, int XF> struct RowX {                 // per-burst input transform
    float isc;
    __device__ __forceinline__ void init(const FirParams &, int, float isc_) { isc = isc_; }
    __device__ __forceinline__ typename Stage<IN>::T operator()(IN v, int) const { return to_stage(v, isc); }
};
template <typename IN> struct RowX<IN, 1> {
    double w;
    int64_t i0;
    __device__ __forceinline__ void init(const FirParams &P, int b, float)
    {
        b = b < P.B ? b : P.B - 1;
        w = 6.283185307179586 * P.derot_f[(size_t)b * P.derot_fstride];          // (2 pi) * f_rel, then * n
        i0 = P.derot_i0 ? P.derot_i0[b] : P.src_off[b];
    }
    __device__ __forceinline__ d2v operator()(IN v, int n) const
    {
        const double th = w * (double)((int64_t)n + i0);
        double sn, cs;
        sincos(th, &sn, &cs);
        const double xr = (double)v.x, xi = (double)v.y;
        return d2v{xr * cs + xi * sn, xi * cs - xr * sn};                        // x * (cos - j sin)
    }
};
// first sample of burst b in the source
__device__ __forceinline__ int64_t burst_origin(const FirParams &P, int b)
{
    return P.src_off ? P.src_off[b < P.B ? b : P.B - 1] : (int64_t)b * P.ld;
}

// (nontemporal: the tiles are read next by another kernel, long after they left the caches; 2.17 -> 2.14 ms per pass)
__device__ __forceinline__ void put(f2v *p, double re, double im) { __builtin_nontemporal_store(f2v{(float)re, (float)im}, p); }
__device__ __forceinline__ void put(d2v *p, double re, double im) { *p = d2v{re, im}; }

// Stage the input windows of one tile: xs[bb][i] = x[b][x0 + i], i < W, zero outside the burst (or the
// stream).  W <= 64 * KS, so a wave covers a row in KS steps; all loads of a wave are issued before
// the first LDS write (each wave would otherwise pay the HBM latency once per load).
template <typename IN, int KS, int NW, int XF = 0>
__device__ __forceinline__ void stage_windows(const FirParams &P, typename StageT<IN, XF>::T *xs, int tile, int x0, int W,
                                              int stride, float isc, int lane, int w)
{
    typedef typename StageT<IN, XF>::T ST;
    constexpr int RQ = 64 / NW;                        // rows a wave stages
    // Interior chunks of full tiles need no bounds logic at all (it is half of the kernel's instructions
    // otherwise): every staged index of all 64 rows lies inside its burst, or inside the stream (window
    // starts do not decrease with the window number).
    bool fast = false;
    if (tile * 64 + 63 < P.B) {
        if (P.starts) fast = P.starts[tile * 64] + x0 >= 0 && P.starts[tile * 64 + 63] + x0 + W <= P.n_total;
        else fast = !P.lens && x0 >= 0 && x0 + W <= P.len_all;
    }
    if (fast) {
        IN v[RQ][KS];
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            const int b = tile * 64 + w * RQ + q;
            const IN *src = P.starts ? (const IN *)P.src + (P.starts[b] + x0) : (const IN *)P.src + (burst_origin(P, b) + x0);
#pragma unroll
            for (int k = 0; k < KS; k++) {
                int i = lane + 64 * k;
                if (i > W - 1) i = W - 1;              // idle lanes of the last step re-read the row's end
                v[q][k] = src[i];
            }
        }
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            ST *row = xs + (size_t)(w * RQ + q) * stride;
            RowX<IN, XF> rx;
            rx.init(P, tile * 64 + w * RQ + q, isc);
#pragma unroll
            for (int k = 0; k < KS; k++) {
                const int i = lane + 64 * k;
                if (64 * (k + 1) <= W) row[i] = rx(v[q][k], x0 + i);
                else if (i < W) row[i] = rx(v[q][k], x0 + i);
            }
        }
    } else {
        IN v[RQ][KS];
        bool ok[RQ][KS];
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            const int b = tile * 64 + w * RQ + q;
            const bool live = b < P.B;
            if (P.starts) {         // window of one long stream: true neighbours across the window edges
                const int64_t st = live ? P.starts[b] : 0;
                const IN *src = (const IN *)P.src;
#pragma unroll
                for (int k = 0; k < KS; k++) {
                    const int i = lane + 64 * k;
                    const int64_t g = st + x0 + i;
                    ok[q][k] = live && i < W && g >= 0 && g < P.n_total;
                    v[q][k] = src[ok[q][k] ? g : 0];
                }
            } else {
                const int len = live ? (P.lens ? P.lens[b] : P.len_all) : 0;
                const IN *src = (const IN *)P.src + burst_origin(P, live ? b : 0);
#pragma unroll
                for (int k = 0; k < KS; k++) {
                    const int i = lane + 64 * k, n = x0 + i;
                    ok[q][k] = i < W && n >= 0 && n < len;
                    v[q][k] = src[ok[q][k] ? n : 0];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            RowX<IN, XF> rx;
            rx.init(P, tile * 64 + w * RQ + q, isc);
#pragma unroll
            for (int k = 0; k < KS; k++) {
                const int i = lane + 64 * k;
                if (i < W) xs[(size_t)(w * RQ + q) * stride + i] = ok[q][k] ? rx(v[q][k], x0 + i) : ST{0, 0};
            }
        }
    }
}

// What a lane (= burst) does with the outputs it has computed: statistics in one pass, the tile store,
// and in the segment-parallel mode the second store into the successor window's warm-up rows.
template <typename OUT>
struct LaneOut {
    int len, st_lo, st_hi, rows, feed_lo, feed_hi, feed_lead;
    double scale;
    OUT *dst, *feed;
    // statistics of this lane's counted outputs: sums of (y - c) and |y - c|^2 around the first counted
    // output c (a sample of the data, so the usual cancellation of the sum-of-squares form cannot
    // occur, whatever the signal's offset)
    double s_re, s_im, s_sq, c_re, c_im;
    int cnt;

    __device__ __forceinline__ void init(const FirParams &P, int tile, int lane)
    {
        const int b = tile * 64 + lane;
        len = b < P.B ? (P.lens ? P.lens[b] : P.len_all) : 0;
        scale = (P.scale && b < P.B) ? P.scale[b] : 1.0;
        st_lo = (P.stat_lo && b < P.B) ? P.stat_lo[b] : 0;
        st_hi = (P.stat_hi && b < P.B) ? P.stat_hi[b] : len;
        rows = P.rows;
        dst = P.dst ? (OUT *)P.dst + (size_t)tile * P.rows * 64 + lane : nullptr;
        // segment-parallel mode: this window's rows [shift, shift + skip_rows) are the warm-up rows
        // [0, skip_rows) of its successor (the same samples, hence the same sums): store them there too
        feed = nullptr;
        feed_lo = 0;
        feed_lead = P.feed_lead;
        if (P.feed_lead && P.dst && b + 1 < P.B && b + 1 >= 64) {
            feed_lo = (int)(P.starts[b + 1] - P.starts[b]);
            feed = (OUT *)P.dst + (size_t)((b + 1) >> 6) * P.rows * 64 + ((b + 1) & 63) - (ptrdiff_t)feed_lo * 64;
        }
        feed_hi = feed_lo + P.skip_rows;
        s_re = 0.0; s_im = 0.0; s_sq = 0.0; c_re = 0.0; c_im = 0.0;
        cnt = 0;
    }

    // outputs n_a .. n_a + OG - 1 of this lane's burst
    template <int OG>
    __device__ __forceinline__ void block(int n_a, const double (&are)[OG], const double (&aim)[OG])
    {
#pragma unroll
        for (int o = 0; o < OG; o++) {
            const int n = n_a + o;
            const bool valid = n < len;
            const bool counts = valid && n >= st_lo && n < st_hi;
            if (counts) {
                if (cnt == 0) { c_re = are[o]; c_im = aim[o]; }
                const double dr = are[o] - c_re, di = aim[o] - c_im;
                s_re += dr; s_im += di; s_sq += dr * dr + di * di;
                cnt++;
            }
            if (dst && n < rows) put(dst + (size_t)n * 64, valid ? are[o] * scale : 0.0, valid ? aim[o] * scale : 0.0);
        }
        if (feed_lead) {
            // a branch per output costs more than the store: groups wholly inside every lane's
            // range (the usual case, ranges and groups are multiples of OG) store unconditionally
            const int n_z = n_a + OG - 1;
            const bool in_all = feed && n_a >= feed_lo && n_z < feed_hi;
            const bool in_any = feed && n_z >= feed_lo && n_a < feed_hi;
            if (__all(in_all)) {
#pragma unroll
                for (int o = 0; o < OG; o++) {
                    const bool valid = n_a + o < len;
                    put(feed + (ptrdiff_t)(n_a + o) * 64, valid ? are[o] * scale : 0.0, valid ? aim[o] * scale : 0.0);
                }
            } else if (__any(in_any)) {
#pragma unroll
                for (int o = 0; o < OG; o++) {
                    const int n = n_a + o;
                    const bool valid = n < len;
                    if (feed && n >= feed_lo && n < feed_hi)
                        put(feed + (ptrdiff_t)n * 64, valid ? are[o] * scale : 0.0, valid ? aim[o] * scale : 0.0);
                }
            }
        }
    }

    // The same for a block every lane of the wave keeps whole: all OG outputs valid, counted and inside
    // the tile's rows, no second store (interior chunks of equal-length bursts) -- no predicates at all.
    template <int OG, bool SCALED>
    __device__ __forceinline__ void block_plain(int n_a, const double (&are)[OG], const double (&aim)[OG])
    {
        if (cnt == 0) { c_re = are[0]; c_im = aim[0]; }
#pragma unroll
        for (int o = 0; o < OG; o++) {
            const double dr = are[o] - c_re, di = aim[o] - c_im;
            s_re += dr; s_im += di; s_sq += dr * dr + di * di;
            if (dst) put(dst + (size_t)(n_a + o) * 64, SCALED ? are[o] * scale : are[o], SCALED ? aim[o] * scale : aim[o]);
        }
        cnt += OG;
    }

    // per (burst, chunk) statistics: each wave's (n, mean, M2), merged over the NW waves (Chan)
    template <int NW>
    __device__ __forceinline__ void reduce_stats(const FirParams &P, double *red, int tile, int chunk, int lane, int w)
    {
        __syncthreads();                    // every wave is done with xs: reuse it as scratch
        {
            const double n = (double)cnt;
            const double mr = cnt ? s_re / n : 0.0, mi = cnt ? s_im / n : 0.0;     // mean of (y - c)
            red[(w * 64 + lane) * 4 + 0] = n;
            red[(w * 64 + lane) * 4 + 1] = c_re + mr;
            red[(w * 64 + lane) * 4 + 2] = c_im + mi;
            red[(w * 64 + lane) * 4 + 3] = cnt ? s_sq - n * (mr * mr + mi * mi) : 0.0;
        }
        __syncthreads();
        const int b = tile * 64 + lane;
        if (w == 0 && b < P.B) {
            double t_n = 0.0, m_re = 0.0, m_im = 0.0, t_m2 = 0.0;
            for (int q = 0; q < NW; q++) {
                const double *r = red + (q * 64 + lane) * 4;
                const double nb = r[0];
                if (nb <= 0.0) continue;
                const double nn = t_n + nb, dr = r[1] - m_re, di = r[2] - m_im;
                t_m2 += r[3] + (dr * dr + di * di) * (t_n * nb / nn);
                m_re += dr * (nb / nn);
                m_im += di * (nb / nn);
                t_n = nn;
            }
            if (t_m2 < 0.0) t_m2 = 0.0;
            double *st = P.stats + ((size_t)b * P.nrec + chunk) * 4;
            st[0] = t_n; st[1] = m_re; st[2] = m_im; st[3] = t_m2;
        }
    }
};

// rows of a tile that another workgroup fills (segment-parallel mode, skip_rows): leave empty statistics
// records behind and clear the idle lanes of the last tile
template <typename OUT, int NT>
__device__ __forceinline__ void skipped_chunk(const FirParams &P, int tile, int chunk, int n0, int CH)
{
    const int sb = tile * 64 + (int)threadIdx.x;
    if (P.stats && threadIdx.x < 64 && sb < P.B) {
        double *st = P.stats + ((size_t)sb * P.nrec + chunk) * 4;
        st[0] = 0.0; st[1] = 0.0; st[2] = 0.0; st[3] = 0.0;
    }
    // the predecessors' workgroups store these rows (feed_lead); the idle lanes of the last tile
    // have no predecessor: clear them (the recurrence kernel loads whole rows)
    const int idle0 = P.B - tile * 64;
    if (P.feed_lead && P.dst && idle0 < 64) {
        OUT *d = (OUT *)P.dst + (size_t)tile * P.rows * 64;
        for (int i = threadIdx.x; i < CH * 64; i += NT) {
            const int r = n0 + (i >> 6), l = i & 63;
            if (l >= idle0 && r < P.rows) put(d + (size_t)r * 64 + l, 0.0, 0.0);
        }
    }
}

// NW waves per workgroup (4 or 8): each stages 64 / NW rows and computes CH / NW outputs per lane
template <typename IN, typename OUT, int OG, int KS, int NW>
__global__ __launch_bounds__(64 * NW) void fir_tile_kernel(FirParams P)
{
    typedef typename Stage<IN>::T ST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = P.M, CH = P.CH, W = P.W, stride = P.stride;
    const int m8 = (M + 7) & ~7;
    double *hrev = (double *)smem;                                  // [m8] reversed taps, zero padded
    ST *xs = (ST *)(smem + (size_t)m8 * 8);                         // [64][stride]
    double *red = (double *)xs;                                     // [4][64][4] reduction scratch, reuses xs

    const int tile = blockIdx.y, chunk = blockIdx.x;
    const int n0 = chunk * CH;
    if (tile > 0 && n0 + CH <= P.skip_rows) {           // rows another kernel fills (segment-parallel mode)
        skipped_chunk<OUT, 64 * NW>(P, tile, chunk, n0, CH);
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float isc = P.in_scale ? *P.in_scale : 1.0f;

    for (int i = threadIdx.x; i < m8; i += 64 * NW) hrev[i] = i < M ? P.h[M - 1 - i] : 0.0;
    const int x0 = n0 + P.start - (M - 1);
    stage_windows<IN, KS, NW>(P, xs, tile, x0, W, stride, isc, lane, w);
    __syncthreads();

    // lane = burst; this wave computes outputs [w*CH/NW, (w+1)*CH/NW) of the chunk, OG at a time
    LaneOut<OUT> lo;
    lo.init(P, tile, lane);
    const ST *xrow = xs + (size_t)lane * stride;
    const int per_wave = CH / NW;
#pragma unroll 1
    for (int g = 0; g * OG < per_wave; g++) {
        const int o0 = w * per_wave + g * OG;
        double are[OG], aim[OG];
#pragma unroll
        for (int o = 0; o < OG; o++) { are[o] = 0.0; aim[o] = 0.0; }
        for (int i0 = 0; i0 < M; i0 += 8) {
            ST xw[OG + 7];
#pragma unroll
            for (int j = 0; j < OG + 7; j++) xw[j] = xrow[o0 + i0 + j];
#pragma unroll
            for (int ii = 0; ii < 8; ii++) {
                const double hv = hrev[i0 + ii];
#pragma unroll
                for (int o = 0; o < OG; o++) {
                    are[o] = __builtin_fma(hv, (double)xw[o + ii].x, are[o]);
                    aim[o] = __builtin_fma(hv, (double)xw[o + ii].y, aim[o]);
                }
            }
        }
        lo.template block<OG>(n0 + o0, are, aim);
    }
    if (P.stats) lo.template reduce_stats<NW>(P, red, tile, chunk, lane, w);
}

// ---------------------------------------------------------------------------
// fir_stream_kernel: the matched filter for filters of up to FIR_MT taps on float32 staging (complex64
// or int16 input) -- the RRC of every reference call site (41 taps at sps 4, span 10).
// Same tile / chunk decomposition, same staging and the same outputs, bit for bit, as fir_tile_kernel;
// what differs is how a wave turns its window into outputs:
//   * the taps never touch LDS or VGPRs: they are wave-uniform, read with scalar loads (constant
//     address space: the buffer is written by fir_pad_taps_k before this launch and by nobody during
//     it) and used straight out of SGPRs as the multiplier of v_fmac_f64;
//   * a lane (= burst) streams its OG + FIR_MT - 1 window samples out of LDS ONCE, in order, widens each
//     to float64 once and adds it into every output it belongs to: OG * 82 FMAs for OG outputs with
//     2 * (OG + 40) conversions and (OG + 40) / 2 LDS reads (the tap-blocked loop of fir_tile_kernel
//     pays 26 conversions, 13 sample reads and 8 tap reads per 96 FMAs, and pads 41 taps to 48);
//   * the loop over the window is fully unrolled in groups of FIR_GS samples; the group's accumulators
//     pass through an empty asm so that the compiler keeps the groups in order (left alone it hoists
//     all loads and conversions to the top and needs 220 VGPRs).
// Each output's sum runs over the samples in ascending order, as in fir_tile_kernel.
constexpr int FIR_STREAM_OG = 14;   // outputs per lane and wave of the stream kernel (chunk = 8 waves x 14 = 112)
constexpr int FIR_MT = DSP_FIR_MT;     // taps the stream kernel is unrolled for (shorter filters are zero padded)
constexpr int FIR_GS = 4;       // window samples per scheduling group

typedef const __attribute__((address_space(4))) double *cdbl_ptr;     // constant address space: scalar loads

// OG outputs of one lane from its OG + FIR_MT - 1 window samples at xs[off ...] (see fir_stream_kernel)
template <typename ST, int OG>
__device__ __forceinline__ void stream_block(const ST *xs, int off, cdbl_ptr hp, double (&are)[OG], double (&aim)[OG])
{
#pragma unroll
    for (int o = 0; o < OG; o++) { are[o] = 0.0; aim[o] = 0.0; }
    constexpr int NS = OG + FIR_MT - 1;                 // window samples of a block of OG outputs
    constexpr int NG = (NS + FIR_GS - 1) / FIR_GS;
    ST cur[FIR_GS], nxt[FIR_GS];
#pragma unroll
    for (int q = 0; q < FIR_GS; q++) cur[q] = xs[off + q];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g + 1 < NG) {
#pragma unroll
            for (int q = 0; q < FIR_GS; q++)
                if ((g + 1) * FIR_GS + q < NS) nxt[q] = xs[off + (g + 1) * FIR_GS + q];
        }
#pragma unroll
        for (int q = 0; q < FIR_GS; q++) {
            const int s = g * FIR_GS + q;
            if (s < NS) {
                const double xr = (double)cur[q].x, xi = (double)cur[q].y;
#pragma unroll
                for (int o = 0; o < OG; o++) {
                    const int j = s - o;            // window sample s = x[n_o + start - (FIR_MT-1) + j]: tap FIR_MT-1-j
                    if (j >= 0 && j < FIR_MT) {
                        const double t = hp[FIR_MT - 1 - j];
                        are[o] = __builtin_fma(t, xr, are[o]);
                        aim[o] = __builtin_fma(t, xi, aim[o]);
                    }
                }
            }
        }
        // keep the groups in program order (see the header of fir_stream_kernel)
#pragma unroll
        for (int o = 0; o < OG; o++) asm volatile("" : "+v"(are[o]), "+v"(aim[o]));
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int q = 0; q < FIR_GS; q++) cur[q] = nxt[q];
    }
}

__global__ void fir_pad_taps_k(const double *__restrict__ h, int M, double *__restrict__ out)
{
    const int i = threadIdx.x;
    if (i < FIR_MT) out[i] = i < M ? h[i] : 0.0;
}

template <typename IN, typename OUT, int OG, int KS, int NW, int XF = 0>
__global__ __launch_bounds__(64 * NW) void fir_stream_kernel(FirParams P)
{
    typedef typename StageT<IN, XF>::T ST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CH = OG * NW, W = CH + FIR_MT - 1;
    const int stride = P.stride;
    ST *xs = (ST *)smem;                                            // [64][stride]
    double *red = (double *)xs;                                     // [NW][64][4] reduction scratch, reuses xs

    // edge mode (beside fir_strip_kernel): blockIdx.x walks the chunks the strips leave out
    const int tile = blockIdx.y + P.tile0;
    const int chunk = !P.edge ? (int)blockIdx.x
                              : ((int)blockIdx.x < P.c_int0 ? (int)blockIdx.x : P.c_int1 + ((int)blockIdx.x - P.c_int0));
    const int rec = P.edge ? P.rec0 + (int)blockIdx.x : chunk;     // statistics record of this workgroup
    const int n0 = chunk * CH;
    if (tile > 0 && n0 + CH <= P.skip_rows) {           // rows another workgroup fills (segment-parallel mode)
        skipped_chunk<OUT, 64 * NW>(P, tile, chunk, n0, CH);
        return;
    }
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float isc = P.in_scale ? *P.in_scale : 1.0f;
    // the filter as FIR_MT taps, zeros appended: y[n] = sum_{j < FIR_MT} h[j] x[n + start - j]
    const int x0 = n0 + P.start - (FIR_MT - 1);
    stage_windows<IN, KS, NW, XF>(P, xs, tile, x0, W, stride, isc, lane, w);
    __syncthreads();

    LaneOut<OUT> lo;
    lo.init(P, tile, lane);
    double are[OG], aim[OG];
    stream_block<ST, OG>(xs, lane * stride + w * OG, (cdbl_ptr)P.h_pad, are, aim);
    const int n_a = n0 + w * OG;
    // interior chunk of equal-length bursts, every output kept, counted and stored once: no predicates
    const bool plain = tile * 64 + 63 < P.B && !P.lens && !P.stat_lo && !P.feed_lead && n0 + CH <= P.len_all &&
                       n0 + CH <= P.rows;
    if (plain) {
        if (P.scale) lo.template block_plain<OG, true>(n_a, are, aim);
        else lo.template block_plain<OG, false>(n_a, are, aim);
    } else {
        lo.template block<OG>(n_a, are, aim);
    }
    if (P.stats) lo.template reduce_stats<NW>(P, red, tile, rec, lane, w);
}

// ---------------------------------------------------------------------------
// fir_strip_kernel: fir_stream_kernel made persistent along the burst.  One workgroup owns one FULL tile
// (64 bursts of equal length) x one strip of consecutive chunks and walks them in order:
//   * every input sample is read from HBM ONCE: the FIR_MT - 1 samples two neighbouring chunks share
//     stay in LDS (moved to the front of each row), only the chunk's CH new samples per burst are loaded
//     (the chunk-per-workgroup kernels read W / CH = 1.36 x the input);
//   * the loads of chunk c + 1 (two samples per lane, whole rows of 896 / 448 bytes) are issued before the
//     arithmetic of chunk c and consumed after it, inside the same loop iteration, so the compiler's
//     own vmcnt bookkeeping leaves the chunk's output stores in flight: a chunk's memory time hides
//     behind its arithmetic instead of adding to it (the compute core alone saturates the fp64 issue
//     rate with two waves per SIMD; two 512-thread workgroups per CU = four);
//   * the np.std statistics stay in the wave's registers for the whole strip: one (n, mean, M2) record
//     per (burst, strip, wave) at the end -- no per-chunk LDS reduction, two barriers per chunk;
//   * the bursts' ends cost a zero-select where a strip's first window is loaded and a second, masked
//     copy of the (short) epilogue and LDS write for the chunk that holds the end: all tests are
//     wave-uniform because the bursts are equally long.  A partly filled last tile is left to
//     fir_stream_kernel (launch_fir_stream_t runs it over just that tile).
// Outputs and per-output sums are the same bits as fir_stream_kernel / fir_tile_kernel.
// One load of a lane: two input samples (16 bytes of complex64, 8 bytes of int16 pairs).  Two, not four,
// for the int16 pairs: the conversions and LDS writes of a row's new samples are wave instructions
// whatever the number of lanes that take part, so 56 lanes x 2 samples cost half of 28 lanes x 4.
template <typename IN> struct Wide;
template <> struct Wide<f2v> { typedef float T __attribute__((ext_vector_type(4))); enum { SPL = 2 }; };
template <> struct Wide<s2v> { typedef short T __attribute__((ext_vector_type(4))); enum { SPL = 2 }; };
template <> struct Wide<d2v> { typedef double T __attribute__((ext_vector_type(2))); enum { SPL = 1 }; };   // one complex128
__device__ __forceinline__ f2v wide_get(const Wide<f2v>::T &v, int i) { return i ? f2v{v.z, v.w} : f2v{v.x, v.y}; }
__device__ __forceinline__ s2v wide_get(const Wide<s2v>::T &v, int i) { return s2v{v[2 * i], v[2 * i + 1]}; }
__device__ __forceinline__ d2v wide_get(const Wide<d2v>::T &v, int) { return d2v{v.x, v.y}; }

// STREAM: the "bursts" are the consecutive segments of ONE long stream (segment-parallel capture mode):
// segment g = stream samples [g * ld, (g + 1) * ld), len_all == ld, and lane l of tile t is segment
// l * ntiles + t (lane-major), so that in the output -- plain tiles of `rows` == ld rows, one after the
// other -- the rows before a tile are, lane by lane, the end of each lane's preceding segment and the rows
// after it the start of the following one: a window with warm-up and tail is a row range of that buffer
// and every filtered sample is stored exactly once.  A segment's window sees its true neighbours across
// the segment edges and zeros only beyond the stream's two ends.
// complex128 input (the drop-in path's sample type) stages float64 pairs: 150 KB of LDS, one workgroup per
// CU with up to 256 VGPRs, two loads per lane for a row's new samples -- that shape is HBM-bound (32 bytes
// per sample in and out), not fp64-bound.
template <typename IN, typename OUT, int OG, bool SCALED, bool STORE, bool STREAM = false, int XF = 0>
__global__ __launch_bounds__(512, sizeof(typename StageT<IN, XF>::T) == 16 ? 2 : 4) void fir_strip_kernel(FirParams P)
{
    static_assert(!(STREAM && XF), "the capture's segments are not derotated");
    typedef typename StageT<IN, XF>::T ST;
    typedef typename Wide<IN>::T WV;
    constexpr int NW = 8, RQ = 64 / NW, CH = OG * NW, HALO = FIR_MT - 1, W = CH + HALO;
    constexpr int SPL = Wide<IN>::SPL;                  // samples per load
    constexpr int LPR = CH / SPL;                       // lanes that load a row's new samples
    constexpr int WPR = W / SPL, NPL = (WPR + 63) / 64; // pieces of a whole window, loads per row for them
    constexpr int NLD = (LPR + 63) / 64;                // loads per lane for a row's new samples
    static_assert(CH % SPL == 0 && W % SPL == 0, "whole pieces");
    constexpr int NSH = (RQ * HALO + 63) / 64;          // halo samples a lane moves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int stride = P.stride;
    ST *xs = (ST *)smem;                                // [64][stride]

    const int tile = blockIdx.y, strip = blockIdx.x;
    const int c_lo = P.c_int0 + strip * P.cps, c_hi = c_lo + P.cps < P.c_int1 ? c_lo + P.cps : P.c_int1;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float isc = P.in_scale ? *P.in_scale : 1.0f;
    const cdbl_ptr hp = (cdbl_ptr)P.h_pad;
    // the RQ rows (bursts) this wave loads, as one wave-uniform base (XF: each row's own origin in the signal)
    const IN *rows0 = (const IN *)P.src + (int64_t)(tile * 64 + w * RQ) * P.ld + P.start - HALO;
    auto row_src = [&](int q) {                                               // sample 0 of row q
        if constexpr (XF != 0) return (const IN *)P.src + P.src_off[tile * 64 + w * RQ + q];
        else return rows0 + (int64_t)q * P.ld + HALO - P.start;
    };
    ST *myrows = xs + w * RQ * stride;
    RowX<IN, XF> rx[RQ];
#pragma unroll
    for (int q = 0; q < RQ; q++) rx[q].init(P, tile * 64 + w * RQ + q, isc);

    // this lane's burst: output row pointer, scale, statistics
    const int ntiles = gridDim.y;
    const int b = STREAM ? lane * ntiles + tile : tile * 64 + lane;
    auto row_burst = [&](int q) { return STREAM ? (w * RQ + q) * ntiles + tile : tile * 64 + w * RQ + q; };
    // STORE (P.dst != NULL) is a template parameter: with a run-time test around the stores the compiler
    // loses count of them and waits for them to complete before it lets the prefetched loads be used
    OUT *dst = (OUT *)P.dst + (size_t)tile * P.rows * 64 + lane;
    const double scale = SCALED ? P.scale[b] : 1.0;
    double s_re = 0.0, s_im = 0.0, s_sq = 0.0, c_re = 0.0, c_im = 0.0;

    const int len = P.len_all, rows = P.rows;
    int cnt = 0;                                        // outputs of this wave that count (wave-uniform)

    // ---- the first chunk's whole window (two-sample pieces; a piece lies wholly inside or outside the burst:
    //      launch_fir_tile checks that the burst's ends fall on piece boundaries) ----------------
    {
        WV v[RQ][NPL];
        const int g0 = c_lo * CH + P.start - HALO;      // burst sample of the window's first sample
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            const IN *src = row_src(q);                                        // the row's sample 0
            const int64_t row_at = (int64_t)row_burst(q) * P.ld;               // ... as a stream sample
#pragma unroll
            for (int k = 0; k < NPL; k++) {
                int g = g0 + (lane + 64 * k) * SPL;
                if (STREAM) {
                    int64_t ga = row_at + g;
                    ga = ga < 0 ? 0 : (ga > P.n_total - SPL ? P.n_total - SPL : ga);
                    v[q][k] = __builtin_nontemporal_load((const WV *)((const IN *)P.src + ga));
                } else {
                    g = g < 0 ? 0 : (g > len - SPL ? len - SPL : g);
                    v[q][k] = __builtin_nontemporal_load((const WV *)(src + g));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            const int64_t row_at = (int64_t)row_burst(q) * P.ld;
#pragma unroll
            for (int k = 0; k < NPL; k++) {
                const int i = lane + 64 * k, g = g0 + i * SPL;
                const bool in = STREAM ? (row_at + g >= 0 && row_at + g + SPL <= P.n_total) : (g >= 0 && g + SPL <= len);
                if (i < WPR) {
#pragma unroll
                    for (int u = 0; u < SPL; u++)
                        myrows[q * stride + i * SPL + u] = in ? rx[q](wide_get(v[q][k], u), g + u) : ST{0, 0};
                }
            }
        }
    }
    __syncthreads();

    for (int c = c_lo; c < c_hi; c++) {
        // chunk c + 1's new samples (the last chunk of the strip loads its own again and drops them:
        // an unconditional load keeps the loop free of joins the compiler would wait at); pieces past
        // the burst's end re-read its last one and are zeroed when they go to LDS
        WV pre[RQ][NLD];
        const int cn = c + 1 < c_hi ? c + 1 : c;
        int gn[NLD];                                   // first sample of this lane's j-th piece
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int piece = lane + 64 * j;
            gn[j] = cn * CH + P.start + (piece < LPR ? piece : LPR - 1) * SPL;
        }
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int g = gn[j] > len - SPL ? len - SPL : gn[j];
#pragma unroll
            for (int q = 0; q < RQ; q++) {
                if (STREAM) {       // the next segment's samples follow in memory; only the stream ends
                    // (wave-uniform row base + a 32-bit lane offset, clamped to the stream's last piece)
                    const int64_t row_at = (int64_t)row_burst(q) * P.ld;
                    const int64_t room = P.n_total - SPL - row_at;
                    const int lim = room > 0x7fffffff ? 0x7fffffff : (int)room;
                    pre[q][j] = __builtin_nontemporal_load((const WV *)((const IN *)P.src + row_at + (gn[j] > lim ? lim : gn[j])));
                } else {
                    pre[q][j] = __builtin_nontemporal_load((const WV *)(row_src(q) + g));
                }
            }
        }
        double are[OG], aim[OG];
        stream_block<ST, OG>(xs, lane * stride + w * OG, hp, are, aim);
        if (c == c_lo) { c_re = are[0]; c_im = aim[0]; }
        const int n_a = c * CH + w * OG;
        OUT *drow = dst + (size_t)n_a * 64;
        if (c * CH + CH <= len && c * CH + CH <= rows) {
            // every output is valid, counted and stored once
#pragma unroll
            for (int o = 0; o < OG; o++) {
                const double dr = are[o] - c_re, di = aim[o] - c_im;
                s_re += dr; s_im += di;
                s_sq = __builtin_fma(di, di, __builtin_fma(dr, dr, s_sq));        // statistics only: free to fuse
                if (STORE) put(drow + (size_t)o * 64, SCALED ? are[o] * scale : are[o], SCALED ? aim[o] * scale : aim[o]);
            }
            cnt += OG;
        } else {
            // the chunk that holds the burst's end: outputs past it are zeros and do not count (the
            // bursts are equally long, so every test is wave-uniform)
#pragma unroll
            for (int o = 0; o < OG; o++) {
                const bool valid = n_a + o < len;
                if (valid) {
                    const double dr = are[o] - c_re, di = aim[o] - c_im;
                    s_re += dr; s_im += di;
                    s_sq = __builtin_fma(di, di, __builtin_fma(dr, dr, s_sq));
                    cnt++;
                }
                if (STORE && n_a + o < rows)
                    put(drow + (size_t)o * 64, valid ? (SCALED ? are[o] * scale : are[o]) : 0.0,
                        valid ? (SCALED ? aim[o] * scale : aim[o]) : 0.0);
            }
        }
        if (c + 1 == c_hi) break;
        // the window of chunk c + 1 = the last HALO samples of this window + the CH prefetched ones.  Every
        // wave moves and fills the RQ rows it loaded; all waves must be done reading the window first.
        __syncthreads();
        {
            ST keep[NSH];
#pragma unroll
            for (int j = 0; j < NSH; j++) {
                const int idx = lane + 64 * j, q = idx / HALO, i = idx - q * HALO;
                if (idx < RQ * HALO) keep[j] = myrows[q * stride + CH + i];
            }
#pragma unroll
            for (int j = 0; j < NSH; j++) {
                const int idx = lane + 64 * j, q = idx / HALO, i = idx - q * HALO;
                if (idx < RQ * HALO) myrows[q * stride + i] = keep[j];
            }
            // nothing to mask: the new samples lie inside the burst (inside the stream: row 63 holds
            // the tile's last segment, 63 * ntiles + tile; if its samples are inside, every row's are)
            const bool whole = STREAM ? ((int64_t)(63 * ntiles + tile) * P.ld + cn * CH + P.start + CH <= P.n_total)
                                      : (cn * CH + P.start + CH <= len);
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                const int piece = lane + 64 * j;
                if (piece >= LPR) continue;
                if (whole) {
#pragma unroll
                    for (int q = 0; q < RQ; q++) {
#pragma unroll
                        for (int u = 0; u < SPL; u++)
                            myrows[q * stride + HALO + piece * SPL + u] = rx[q](wide_get(pre[q][j], u), gn[j] + u);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < RQ; q++) {
                        const bool in = STREAM ? ((int64_t)row_burst(q) * P.ld + gn[j] + SPL <= P.n_total)
                                               : (gn[j] + SPL <= len);
#pragma unroll
                        for (int u = 0; u < SPL; u++)
                            myrows[q * stride + HALO + piece * SPL + u] = in ? rx[q](wide_get(pre[q][j], u), gn[j] + u) : ST{0, 0};
                    }
                }
            }
        }
        __syncthreads();
    }
    if (P.stats) {
        // this lane's (n, mean, M2) over the strip as record strip * NW + w of its burst.  STREAM: a last
        // segment that the stream's end cuts short is left out here (its outputs past the end are the
        // filter's tail over zeros); the caller adds its statistics with a pass of its own
        const bool cut = STREAM && (int64_t)(b + 1) * P.ld > P.n_total;
        const double n = cut ? 0.0 : (double)cnt;
        const double mr = cnt ? s_re / n : 0.0, mi = cnt ? s_im / n : 0.0;    // mean of (y - c)
        double m2 = cnt ? s_sq - n * (mr * mr + mi * mi) : 0.0;
        if (m2 < 0.0) m2 = 0.0;
        double *st = P.stats + ((size_t)b * P.nrec + strip * NW + w) * 4;
        st[0] = n; st[1] = c_re + mr; st[2] = c_im + mi; st[3] = m2;
    }
}

// merge the chunk statistics of each burst (Chan et al.) -> std and 1/std
// (records [0, cnt_a) of the bursts below b_split, [0, cnt_b) of the others; `nrec` records apart)
__global__ void fir_stats_finish(const double *__restrict__ stats, int nrec, int cnt_a, int b_split, int cnt_b, int B,
                                 double *__restrict__ std_out, double *__restrict__ inv_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double n = 0.0, mr = 0.0, mi = 0.0, m2 = 0.0;
    const int cnt = b < b_split ? cnt_a : cnt_b;
    for (int c = 0; c < cnt; c++) {
        const double *s = stats + ((size_t)b * nrec + c) * 4;
        const double nb = s[0];
        if (nb <= 0.0) continue;
        const double nn = n + nb;
        const double dr = s[1] - mr, di = s[2] - mi;
        m2 += s[3] + (dr * dr + di * di) * (n * nb / nn);
        mr += dr * (nb / nn);
        mi += di * (nb / nn);
        n = nn;
    }
    const double sd = n > 0.0 ? sqrt(m2 / n) : __builtin_nan("");
    if (std_out) std_out[b] = sd;
    if (inv_out) inv_out[b] = 1.0 / sd;          // numpy: complex / real scalar = * (1/std)
}

// merge ALL partial statistics (every burst, every chunk) into one population std: the windows
// of a segmented capture are normalised by the std of the whole filtered stream (worker.py:177
// applied to one long signal).  One workgroup; inv_fill[0..B) all receive 1/std.
// With `partial` the workgroups each merge a slice and write one (n, mean, M2) record; the last
// step (partial == NULL, one workgroup) merges those records and produces the result.
__global__ __launch_bounds__(256) void fir_stats_global(const double *__restrict__ stats, int64_t count, int B,
                                                        double *__restrict__ std_out, double *__restrict__ inv_fill,
                                                        double *__restrict__ partial)
{
    __shared__ double red[256 * 4];
    double n = 0.0, mr = 0.0, mi = 0.0, m2 = 0.0;
    const int64_t per = (count + gridDim.x - 1) / gridDim.x;
    const int64_t c_lo = blockIdx.x * per, c_hi = c_lo + per < count ? c_lo + per : count;
    for (int64_t c = c_lo + threadIdx.x; c < c_hi; c += 256) {
        const double *s = stats + c * 4;
        const double nb = s[0];
        if (nb <= 0.0) continue;
        const double nn = n + nb;
        const double dr = s[1] - mr, di = s[2] - mi;
        m2 += s[3] + (dr * dr + di * di) * (n * nb / nn);
        mr += dr * (nb / nn);
        mi += di * (nb / nn);
        n = nn;
    }
    red[threadIdx.x * 4] = n; red[threadIdx.x * 4 + 1] = mr; red[threadIdx.x * 4 + 2] = mi; red[threadIdx.x * 4 + 3] = m2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            double *a = red + threadIdx.x * 4;
            const double *b = red + (threadIdx.x + st) * 4;
            if (b[0] > 0.0) {
                const double nn = a[0] + b[0];
                const double dr = b[1] - a[1], di = b[2] - a[2];
                a[3] += b[3] + (dr * dr + di * di) * (a[0] * b[0] / nn);
                a[1] += dr * (b[0] / nn);
                a[2] += di * (b[0] / nn);
                a[0] = nn;
            }
        }
        __syncthreads();
    }
    if (partial) {
        if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x];
        return;
    }
    const double sd = red[0] > 0.0 ? sqrt(red[3] / red[0]) : __builtin_nan("");
    if (threadIdx.x == 0 && std_out) *std_out = sd;
    if (inv_fill) for (int b = threadIdx.x; b < B; b += 256) inv_fill[b] = 1.0 / sd;
}

// generic fallback for long filters: one thread per output sample, burst-major in and out
template <typename IN, typename OUT>
__global__ void fir_direct_kernel(FirParams P, OUT *__restrict__ dst, int64_t dst_ld)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = P.lens ? P.lens[b] : P.len_all;
    if (n >= len) return;
    const IN *src = (const IN *)P.src + (int64_t)b * P.ld;
    const float isc = P.in_scale ? *P.in_scale : 1.0f;
    const int m = n + P.start;
    int j_hi = m < P.M - 1 ? m : P.M - 1;
    int j_lo = m - (len - 1) > 0 ? m - (len - 1) : 0;
    double sr = 0.0, si = 0.0;
    for (int j = j_hi; j >= j_lo; j--) {
        auto v = to_stage(src[m - j], isc);
        sr = __builtin_fma(P.h[j], (double)v.x, sr);
        si = __builtin_fma(P.h[j], (double)v.y, si);
    }
    const double scale = P.scale ? P.scale[b] : 1.0;
    put(dst + (int64_t)b * dst_ld + n, sr * scale, si * scale);
}

// exact two-pass np.std of burst-major complex rows (one workgroup per burst)
template <typename T>
__global__ __launch_bounds__(256) void burst_std_kernel(const T *__restrict__ y, int64_t ld, const int *lens,
                                                       int len_all, double *__restrict__ std_out,
                                                       double *__restrict__ inv_out)
{
    __shared__ double red[256 * 2];
    const int b = blockIdx.x;
    const int len = lens ? lens[b] : len_all;
    const T *row = y + (int64_t)b * ld;
    double sr = 0.0, si = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) { sr += (double)row[i].x; si += (double)row[i].y; }
    red[threadIdx.x * 2] = sr; red[threadIdx.x * 2 + 1] = si;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { red[threadIdx.x * 2] += red[(threadIdx.x + s) * 2]; red[threadIdx.x * 2 + 1] += red[(threadIdx.x + s) * 2 + 1]; }
        __syncthreads();
    }
    const double mr = red[0] / (double)len, mi = red[1] / (double)len;
    __syncthreads();
    double m2 = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) {
        const double dr = (double)row[i].x - mr, di = (double)row[i].y - mi;
        m2 += dr * dr + di * di;
    }
    red[threadIdx.x] = m2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sd = len > 0 ? sqrt(red[0] / (double)len) : __builtin_nan("");
        if (std_out) std_out[b] = sd;
        if (inv_out) inv_out[b] = 1.0 / sd;
    }
}

// y *= scale[b] on burst-major rows (complex128: y * (1/std) exactly as numpy divides)
template <typename T>
__global__ void burst_scale_kernel(T *__restrict__ y, int64_t ld, const int *lens, int len_all,
                                   const double *__restrict__ scale)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = lens ? lens[b] : len_all;
    if (n >= len) return;
    T v = y[(int64_t)b * ld + n];
    const double s = scale[b];
    put(&y[(int64_t)b * ld + n], (double)v.x * s, (double)v.y * s);
}

// complex128 -> complex64 rows (one rounding)
__global__ void burst_cast_kernel(const d2v *__restrict__ src, int64_t src_ld, f2v *__restrict__ dst, int64_t dst_ld,
                                  int n)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const d2v v = src[(int64_t)b * src_ld + i];
    dst[(int64_t)b * dst_ld + i] = f2v{(float)v.x, (float)v.y};
}

// ---- raw IQ loaders (hw1/gardner.py:77-91) -----------------------------------
// peak^2 = max(i^2 + q^2) over the capture, exact in integers
__device__ __forceinline__ unsigned iq_mag2(unsigned w)
{
    const int a = (short)(w & 0xffffu), c = (short)(w >> 16);
    return (unsigned)(a * a) + (unsigned)(c * c);
}
// max |i + jq|^2 over the capture: a read-only stream, 16-byte loads (4 samples), 4 of them in flight
// per lane; the few samples before the first 16-byte boundary and after the last whole word go one
// by one.  raw is 4-byte aligned (one IQ pair).
__global__ __launch_bounds__(256) void iq_i16_peak_kernel(const s2v *__restrict__ raw, int64_t n, unsigned *__restrict__ peak2)
{
    unsigned m = 0;
    const unsigned *w = (const unsigned *)raw;
    int64_t head = (int64_t)(((16 - ((uintptr_t)raw & 15)) & 15) >> 2);
    if (head > n) head = n;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const u4v *body = (const u4v *)(w + head);
    const int64_t nq = (n - head) >> 2, tail0 = head + 4 * nq;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    if (tid < head) m = iq_mag2(w[tid]);
    if (tail0 + tid < n) { const unsigned v = iq_mag2(w[tail0 + tid]); m = v > m ? v : m; }   // < 4 samples
    for (int64_t i = tid; i < nq; i += 4 * nthr) {
        u4v v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t j = i + u * nthr;
            v[u] = j < nq ? __builtin_nontemporal_load(body + j) : u4v{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned a0 = iq_mag2(v[u].x), a1 = iq_mag2(v[u].y), a2 = iq_mag2(v[u].z), a3 = iq_mag2(v[u].w);
            const unsigned b0 = a0 > a1 ? a0 : a1, b1 = a2 > a3 ? a2 : a3, c0 = b0 > b1 ? b0 : b1;
            m = c0 > m ? c0 : m;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_down(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(peak2, m);
}

// 1/peak in float32, as numpy's complex64 / float32 scalar computes it:
// peak = float32(hypot) (exact sqrt of the integer in double, rounded once), scl = 1.0f / peak
__global__ void iq_i16_scale_kernel(const unsigned *__restrict__ peak2, int normalize, float *__restrict__ iscale)
{
    float s = 1.0f;
    if (normalize && *peak2 > 0) {
        const float peak = (float)sqrt((double)*peak2);
        s = 1.0f / peak;
    }
    *iscale = s;
}

__global__ void iq_i16_convert_kernel(const s2v *__restrict__ raw, int64_t n, const float *__restrict__ iscale,
                                      f2v *__restrict__ out)
{
    const float s = *iscale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f2v{(float)raw[i].x * s, (float)raw[i].y * s};
}

template <typename IN, typename OUT, int XF = 0>
hipError_t launch_fir_stream_t(const FirParams &P, hipStream_t s)
{
    constexpr int OG = FIR_STREAM_OG, NW = 8, CH = OG * NW, W = CH + FIR_MT - 1, KS = (W + 63) / 64;
    if (P.CH != CH || P.W != W || !P.h_pad) return hipErrorInvalidValue;
    if (XF ? (!P.derot_f || !P.src_off || P.stream_mode || P.starts) : (P.derot_f != nullptr)) return hipErrorInvalidValue;
    const int ntiles = (P.B + 63) / 64;
    size_t lds = (size_t)64 * P.stride * sizeof(typename StageT<IN, XF>::T);       // 78 KB, 153 KB for complex128
    if (lds < (size_t)NW * 64 * 4 * 8) lds = (size_t)NW * 64 * 4 * 8;              // the statistics scratch
    hipLaunchKernelGGL(fir_pad_taps_k, dim3(1), dim3(64), 0, s, P.h, P.M, P.h_pad);
    void (*k)(FirParams) = fir_stream_kernel<IN, OUT, OG, KS, NW, XF>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    if (P.cps > 0) {
        // strips over the interior chunks of the full tiles ...
        void (*ks)(FirParams) = P.dst ? (P.scale ? fir_strip_kernel<IN, OUT, OG, true, true, false, XF>
                                                 : fir_strip_kernel<IN, OUT, OG, false, true, false, XF>)
                                      : fir_strip_kernel<IN, OUT, OG, false, false, false, XF>;
        if constexpr (XF == 0)
            if (P.stream_mode) ks = fir_strip_kernel<IN, OUT, OG, false, true, true>; // capture mode: unscaled, stored
        hipError_t e = hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        // 78 KB of LDS and <= 128 VGPRs: two workgroups per CU, so that one's load / barrier phases fall
        // into the other's arithmetic (2.20 ms against 2.31 with one workgroup per CU at 65536 x 8192)
        const size_t lds1 = lds;
        const int nstrips = (P.c_int1 - P.c_int0 + P.cps - 1) / P.cps;
        hipLaunchKernelGGL(ks, dim3(nstrips, P.strip_tiles), dim3(512), lds1, s, P);
        // ... the chunks at the ends of their bursts ...
        const int n_edge = P.c_int0 + (P.nchunk - P.c_int1);
        if (n_edge > 0) {
            FirParams E = P;
            E.edge = 1; E.rec0 = nstrips * 8; E.B = P.strip_tiles * 64;
            hipLaunchKernelGGL(k, dim3(n_edge, P.strip_tiles), dim3(64 * NW), lds, s, E);
        }
        // ... and a partly filled last tile, chunk by chunk
        if (ntiles > P.strip_tiles) {
            FirParams L = P;
            L.tile0 = P.strip_tiles;
            hipLaunchKernelGGL(k, dim3(P.nchunk, ntiles - P.strip_tiles), dim3(64 * NW), lds, s, L);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k, dim3(P.nchunk, ntiles), dim3(64 * NW), lds, s, P);
    return hipGetLastError();
}

template <typename IN, typename OUT, int OG, int OG8>
hipError_t launch_fir_tile_t(const FirParams &P, hipStream_t s)
{
    const int nchunk = P.nchunk, ntiles = (P.B + 63) / 64;
    size_t xs_bytes = (size_t)64 * P.stride * sizeof(typename Stage<IN>::T);
    if (xs_bytes < (size_t)8 * 64 * 4 * 8) xs_bytes = (size_t)8 * 64 * 4 * 8;      // the statistics scratch: [NW][64][4] doubles
    size_t lds = (size_t)((P.M + 7) & ~7) * 8 + xs_bytes;
    if (P.W > 192) return hipErrorInvalidValue;
    // 8 waves per workgroup (two workgroups per CU = 4 waves per SIMD: the SIMD issues an fp64 instruction
    // every 2.7 ticks instead of 3.5 at 2 waves; 4.18 -> 3.66 ms per pass) when each wave's share of the
    // chunk is a whole number of OG8-output blocks; 4 waves with OG-output blocks otherwise
    const bool w8 = OG8 > 0 && P.waves != 4 && P.CH >= 8 * OG8 && (P.CH / 8) % OG8 == 0;
    void (*k)(FirParams);
    if (w8) k = P.W > 128 ? fir_tile_kernel<IN, OUT, (OG8 > 0 ? OG8 : OG), 3, 8> : fir_tile_kernel<IN, OUT, (OG8 > 0 ? OG8 : OG), 2, 8>;
    else k = P.W > 128 ? fir_tile_kernel<IN, OUT, OG, 3, 4> : fir_tile_kernel<IN, OUT, OG, 2, 4>;
    if (P.W > 128) {
        // more than 64 KB of dynamic LDS has to be asked for (per device: cheap, so every launch)
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(nchunk, ntiles), dim3(w8 ? 512 : 256), lds, s, P);
    return hipGetLastError();
}

}  // namespace

// choose the chunk geometry; returns false if the filter is too long for the LDS-tiled kernel
bool fir_plan(int in_kind, int M, int n_out, FirParams &P, bool allow_stream, bool c128_stream)
{
    P.stream = 0;
    if ((in_kind != DSP_IN_C128 || c128_stream) && M <= FIR_MT && allow_stream) {
        // the stream kernel (fir_stream_kernel): 8 waves x FIR_STREAM_OG outputs per chunk
        P.stream = 1;
        P.CH = FIR_STREAM_OG * 8; P.W = P.CH + FIR_MT - 1; P.stride = P.W | 1;
        P.nchunk = (n_out + P.CH - 1) / P.CH;
        return true;
    }
    const int elem = in_kind == DSP_IN_C128 ? 16 : 8;
    const int og = in_kind == DSP_IN_C128 ? 4 : 8;
    if (M > 4096) return false;
    const int m8 = (M + 7) & ~7;
    const size_t budget = 80 * 1024 - (size_t)m8 * 8;   // two workgroups per CU (160 KB of LDS)
    int best = 0;
    // CH / 4 outputs per wave, a whole number of og-output register blocks
    for (int CH = 128; CH >= 4 * og; CH -= 4 * og) {
        int W = CH + m8 - 1;               // highest staged index read: CH + ceil8(M) - 2
        int stride = W | 1;
        if (W <= 192 && (size_t)64 * stride * elem <= budget) { best = CH; break; }
    }
    if (!best) return false;
    P.CH = best; P.W = best + m8 - 1; P.stride = P.W | 1;
    P.nchunk = (n_out + best - 1) / best;
    return true;
}

// Strips of the persistent kernel over the interior chunks [c_int0, c_int1) of the full tiles: as many as
// spread the workgroups over the CUs evenly, each at least 8 chunks long.  Sets cps (0 = not eligible:
// chunk-per-workgroup kernels only), the interior range, strip_tiles and the statistics record layout.
static void fir_plan_strips(FirParams &P, int in_kind)
{
    P.cps = 0; P.edge = 0; P.tile0 = 0; P.rec0 = 0; P.strip_tiles = 0;
    P.nrec = P.nchunk; P.rec_full = P.nchunk;
    if (!P.stream || !P.strip_ok || P.starts || P.lens || P.skip_rows || P.feed_lead || P.stat_lo || P.stat_hi) return;
    const int spl = in_kind == DSP_IN_C128 ? 1 : 2;                    // samples per load (Wide<IN>)
    const int esz = in_kind == DSP_IN_I16 ? 4 : in_kind == DSP_IN_C128 ? 16 : 8;
    const int piece = spl * esz;                                       // bytes per load: 8 or 16
    if ((P.start - (FIR_MT - 1)) % spl) return;
    // bursts cut at arbitrary samples (src_off) start on a sample, not on a piece: global loads need only
    // the element's own alignment, and the pieces are counted from each burst's first sample
    if (!P.src_off && (P.ld % spl || ((uintptr_t)P.src % piece) || (P.ld * esz) % piece)) return;
    if (P.len_all % spl || P.rows < P.len_all || P.start < 0) return;   // the bursts' ends fall between two loads' pieces
    // (the stream may end anywhere inside the B segments, on a 16-byte piece: a cut segment and empty ones
    //  behind it are zeros to the filter and left out of its statistics)
    if (P.stream_mode && (P.B % 64 || P.ld != P.len_all || P.n_total > (int64_t)P.B * P.ld || P.n_total <= 0 ||
                          P.n_total % spl || !P.dst || P.scale ||
                          P.rows != P.len_all)) { P.stream_mode = 0; return; }
    const int c0 = 0, c1 = P.nchunk;                                     // the strips cover every chunk
    const int nint = c1 - c0, full = P.B / 64;
    if (nint < 8 || full < 1) return;
    const int n_cu = P.n_cu > 0 ? P.n_cu : 256;
    const int n_edge = c0 + (P.nchunk - c1);
    int best = 0;
    double best_cost = 0.0;
    for (int ns = 1; ns <= 16; ns++) {
        const int cps = (nint + ns - 1) / ns;
        if (cps < 8) break;
        const int real = (nint + cps - 1) / cps;
        if (real * 8 + n_edge > P.nchunk + 8) break;                   // the workspace holds nchunk + 8 records per burst
        const long wgs = (long)full * real;
        // rounds of workgroups over the CUs x strip length (+ ~2 chunks of fixed cost per strip)
        const double cost = (double)((wgs + n_cu - 1) / n_cu) * (cps + 2);
        if (!best || cost < best_cost * 0.97) { best = cps; best_cost = cost; }
    }
    if (!best) return;
    P.cps = best; P.c_int0 = c0; P.c_int1 = c1; P.strip_tiles = full;
    P.rec_full = 8 * ((nint + best - 1) / best) + n_edge;
    P.nrec = P.rec_full > P.nchunk ? P.rec_full : P.nchunk;
    if (P.stream_mode) P.nrec = P.rec_full;         // every record is written: the stream's std merges them all
}

bool fir_stream_mode_plan(FirParams &P, int in_kind)
{
    if (!P.stream_mode) return false;
    fir_plan_strips(P, in_kind);
    if (!P.cps) P.stream_mode = 0;
    return P.stream_mode != 0;
}

// complex128 bursts: the persistent strip kernel (float64 staging, one workgroup per CU) where its plan
// applies -- full tiles of equally long, 16-byte aligned bursts -- else P keeps fir_tile_kernel's plan.
// P: every input field filled, planned by fir_plan(..., allow_stream = false).
bool fir_try_c128_strips(FirParams &P)
{
    if (P.M > FIR_MT || P.B < 64) return false;
    FirParams T = P;
    if (!fir_plan(DSP_IN_C128, P.M, P.nchunk * P.CH, T, true, true) || !T.stream) return false;
    T.nchunk = (P.len_all + T.CH - 1) / T.CH;
    T.strip_ok = 1;
    fir_plan_strips(T, DSP_IN_C128);
    if (T.cps <= 0) return false;
    P = T;
    return true;
}

// cut + derotation + filter in one pass (FirParams::derot_f): the stream / strip kernels over float64
// staging, output complex128 tiles
bool fir_plan_derot(FirParams &P, int in_kind)
{
    if (in_kind != DSP_IN_C64 && in_kind != DSP_IN_C128) return false;
    if (!P.derot_f || !P.src_off || P.M > FIR_MT || P.lens || P.starts || P.stream_mode || P.in_scale) return false;
    FirParams T = P;
    if (!fir_plan(DSP_IN_C128, P.M, P.len_all, T, true, true) || !T.stream) return false;
    P.stream = 1; P.CH = T.CH; P.W = T.W; P.stride = T.stride; P.nchunk = T.nchunk;
    // The statistics are merged strip by strip or chunk by chunk, and np.std's last bit follows the grouping:
    // the fused pass must group as the two-pass form does (complex128 bursts of their own, one sample per
    // load).  Without strips the two-pass form runs fir_tile_kernel's chunks (fewer than 64 bursts, fewer than
    // 8 chunks); complex64 signals load two samples at a time and lose the strips when the packet length or
    // the filter's start is odd -- those shapes stay with the two passes.
    FirParams A = P, C = P;
    fir_plan_strips(A, in_kind);
    C.src = nullptr; C.src_off = nullptr; C.derot_f = nullptr; C.ld = P.len_all;
    fir_plan_strips(C, DSP_IN_C128);
    return C.cps > 0 && A.cps == C.cps && A.strip_tiles == C.strip_tiles;
}

hipError_t launch_fir_derot(FirParams &P, int in_kind, hipStream_t s)
{
    if (!P.stream || !P.derot_f || !P.src_off) return hipErrorInvalidValue;
    fir_plan_strips(P, in_kind);               // samples per load of the INPUT type; float64 staging either way
    if (in_kind == DSP_IN_C64) return launch_fir_stream_t<f2v, d2v, 1>(P, s);
    if (in_kind == DSP_IN_C128) return launch_fir_stream_t<d2v, d2v, 1>(P, s);
    return hipErrorInvalidValue;
}

hipError_t launch_fir_tile(FirParams &P, int in_kind, hipStream_t s)
{
    if (P.derot_f || P.src_off) return hipErrorInvalidValue;    // launch_fir_derot's
    const int want_stream_mode = P.stream_mode;
    fir_plan_strips(P, in_kind);
    if (want_stream_mode && (!P.cps || !P.stream_mode)) return hipErrorInvalidValue;   // see fir_stream_mode_plan
    if (P.stream) {
        // the start of the zero-padded filter moves with the padding: none here (zeros are appended)
        if (in_kind == DSP_IN_I16) return launch_fir_stream_t<s2v, f2v>(P, s);
        if (in_kind == DSP_IN_C64) return launch_fir_stream_t<f2v, f2v>(P, s);
        if (in_kind == DSP_IN_C128) return launch_fir_stream_t<d2v, d2v>(P, s);
        return hipErrorInvalidValue;
    }
    switch (in_kind) {
        case DSP_IN_I16: return launch_fir_tile_t<s2v, f2v, 8, 6>(P, s);
        case DSP_IN_C64: return launch_fir_tile_t<f2v, f2v, 8, 6>(P, s);
        case DSP_IN_C128: return launch_fir_tile_t<d2v, d2v, 4, 0>(P, s);
    }
    return hipErrorInvalidValue;
}

// ... one wave per burst: few bursts with many records each (a single long burst: 1313 records merged by one
// thread took 0.31 ms of a 9 ms call).  Lanes merge every 64th record, then pairs of lanes merge.
__device__ __forceinline__ void chan_merge(double &n, double &mr, double &mi, double &m2, double nb, double br, double bi,
                                           double b2)
{
    if (nb <= 0.0) return;
    const double nn = n + nb;
    const double dr = br - mr, di = bi - mi;
    m2 += b2 + (dr * dr + di * di) * (n * nb / nn);
    mr += dr * (nb / nn);
    mi += di * (nb / nn);
    n = nn;
}
__global__ __launch_bounds__(64) void fir_stats_finish_wave(const double *__restrict__ stats, int nrec, int cnt_a, int b_split,
                                                            int cnt_b, int B, double *__restrict__ std_out,
                                                            double *__restrict__ inv_out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    double n = 0.0, mr = 0.0, mi = 0.0, m2 = 0.0;
    const int cnt = b < b_split ? cnt_a : cnt_b;
    for (int c = lane; c < cnt; c += 64) {
        const double *r = stats + ((size_t)b * nrec + c) * 4;
        chan_merge(n, mr, mi, m2, r[0], r[1], r[2], r[3]);
    }
    for (int d = 1; d < 64; d <<= 1) {
        const double on = __shfl_xor(n, d), or_ = __shfl_xor(mr, d), oi = __shfl_xor(mi, d), o2 = __shfl_xor(m2, d);
        // both lanes of a pair merge (lower, upper) in that order, so they hold the same bits afterwards
        if (lane & d) {
            double tn = on, tr = or_, ti = oi, t2 = o2;
            chan_merge(tn, tr, ti, t2, n, mr, mi, m2);
            n = tn; mr = tr; mi = ti; m2 = t2;
        } else {
            chan_merge(n, mr, mi, m2, on, or_, oi, o2);
        }
    }
    if (lane == 0) {
        const double sd = n > 0.0 ? sqrt(m2 / n) : __builtin_nan("");
        if (std_out) std_out[b] = sd;
        if (inv_out) inv_out[b] = 1.0 / sd;
    }
}

hipError_t launch_fir_stats_finish(const FirParams &P, double *std_out, double *inv_out, hipStream_t s)
{
    // after launch_fir_tile: the bursts of the tiles the strips covered hold rec_full records, the others nchunk
    const int split = P.cps > 0 ? P.strip_tiles * 64 : 0;
    const int most = P.rec_full > P.nchunk ? P.rec_full : P.nchunk;
    if (P.B <= 2048 && most >= 128)
        hipLaunchKernelGGL(fir_stats_finish_wave, dim3(P.B), dim3(64), 0, s, P.stats, P.nrec, P.rec_full, split, P.nchunk,
                           P.B, std_out, inv_out);
    else
        hipLaunchKernelGGL(fir_stats_finish, dim3((P.B + 255) / 256), dim3(256), 0, s, P.stats, P.nrec, P.rec_full, split,
                           P.nchunk, P.B, std_out, inv_out);
    return hipGetLastError();
}

hipError_t launch_fir_stats_global(const double *stats, int64_t count, int B, double *std_out, double *inv_fill,
                                   double *partial, int n_partial, hipStream_t s)
{
    if (partial && n_partial > 1 && count > 4096) {
        hipLaunchKernelGGL(fir_stats_global, dim3(n_partial), dim3(256), 0, s, stats, count, B, nullptr, nullptr, partial);
        hipLaunchKernelGGL(fir_stats_global, dim3(1), dim3(256), 0, s, partial, (int64_t)n_partial, B, std_out, inv_fill,
                           nullptr);
    } else {
        hipLaunchKernelGGL(fir_stats_global, dim3(1), dim3(256), 0, s, stats, count, B, std_out, inv_fill, nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_fir_direct(const FirParams &P, int in_kind, void *dst, int64_t dst_ld, int n_out, hipStream_t s)
{
    dim3 grid((n_out + 255) / 256, P.B), block(256);
    switch (in_kind) {
        // the direct form always writes complex128 (statistics and scaling stay in float64)
        case DSP_IN_I16: hipLaunchKernelGGL((fir_direct_kernel<s2v, d2v>), grid, block, 0, s, P, (d2v *)dst, dst_ld); break;
        case DSP_IN_C64: hipLaunchKernelGGL((fir_direct_kernel<f2v, d2v>), grid, block, 0, s, P, (d2v *)dst, dst_ld); break;
        case DSP_IN_C128: hipLaunchKernelGGL((fir_direct_kernel<d2v, d2v>), grid, block, 0, s, P, (d2v *)dst, dst_ld); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_burst_std(const void *y, int c128, int64_t ld, const int *lens, int len_all, int B,
                            double *std_out, double *inv_out, hipStream_t s)
{
    if (c128) hipLaunchKernelGGL(burst_std_kernel<d2v>, dim3(B), dim3(256), 0, s, (const d2v *)y, ld, lens, len_all, std_out, inv_out);
    else hipLaunchKernelGGL(burst_std_kernel<f2v>, dim3(B), dim3(256), 0, s, (const f2v *)y, ld, lens, len_all, std_out, inv_out);
    return hipGetLastError();
}

hipError_t launch_burst_scale(void *y, int c128, int64_t ld, const int *lens, int len_all, int B, int n_max,
                              const double *scale, hipStream_t s)
{
    dim3 grid((n_max + 255) / 256, B), block(256);
    if (c128) hipLaunchKernelGGL(burst_scale_kernel<d2v>, grid, block, 0, s, (d2v *)y, ld, lens, len_all, scale);
    else hipLaunchKernelGGL(burst_scale_kernel<f2v>, grid, block, 0, s, (f2v *)y, ld, lens, len_all, scale);
    return hipGetLastError();
}

hipError_t launch_iq_i16(const void *raw, int64_t n, int normalize, unsigned *peak2, float *iscale, void *out,
                         hipStream_t s)
{
    hipError_t e = hipMemsetAsync(peak2, 0, 4, s);
    if (e != hipSuccess) return e;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    int pblocks = (int)((n / 4 + 1023) / 1024);          // 4 x 16 bytes per lane and pass
    if (pblocks > 8192) pblocks = 8192;
    if (pblocks < 1) pblocks = 1;
    if (normalize) hipLaunchKernelGGL(iq_i16_peak_kernel, dim3(pblocks), dim3(256), 0, s, (const s2v *)raw, n, peak2);
    hipLaunchKernelGGL(iq_i16_scale_kernel, dim3(1), dim3(1), 0, s, peak2, normalize, iscale);
    if (out) hipLaunchKernelGGL(iq_i16_convert_kernel, dim3(blocks), dim3(256), 0, s, (const s2v *)raw, n, iscale, (f2v *)out);
    return hipGetLastError();
}

hipError_t launch_burst_cast(const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int B, int n, hipStream_t s)
{
    if (B <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(burst_cast_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, (const d2v *)src, src_ld,
                       (f2v *)dst, dst_ld, n);
    return hipGetLastError();
}

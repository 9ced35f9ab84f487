// segment_kernels.hip -- segment-parallel mode for ONE long capture (north-star (c)).
//
// The sequential loops of gardner2.py / fll_func.py cannot be parallelised inside one stream,
// but they forget their start: after a warm-up the timing loop sits at the same mu wherever
// it was started.  The capture is therefore cut into S segments of seg_len samples; segment s
// becomes one lane ("burst") of the recurrence kernel that starts `lead` samples BEFORE its
// seam (warm-up overlap) and runs `tail` samples past the next seam.  These kernels do the
// bookkeeping around that launch:
//
//   seg_windows  window start / length of every segment
//   seg_scan     per window: the recurrence kernel wrote the sample position k + mu of every
//                produced symbol; the last symbol in front of the window's closing seam ends
//                the window's share (binary search, positions are increasing)
//   seg_range    per window: its share starts half a symbol after the predecessor's last
//                symbol, so shares tile the stream with no gap and no duplicate; the QPSK
//                quadrant ambiguity of a freshly locked decision-directed loop is resolved by a
//                vote over the symbols both windows decoded (the warm-up overlap)
//   seg_offsets  exclusive scan of the share sizes, cumulative rotation
//   seg_gather   rotated decisions (and mu / frequency) of all shares, concatenated
//
// There is no reference counterpart; the oracle restates the same
// definition on the CPU (oracle/oracle.py sync_capture_segmented).
#include "dsp_internal.h"
#include "sync_lane.h"

#pragma clang fp contract(off)

namespace {

// quadrant index (Re<0) + 2 (Im<0) after a turn by r * 90 degrees
__device__ __forceinline__ unsigned rot_quadrant(unsigned q, int r)
{
    const unsigned turn[4] = {1u, 3u, 0u, 2u};       // +90 degrees: (re, im) -> (-im, re)
    for (int i = 0; i < (r & 3); i++) q = turn[q & 3u];
    return q & 3u;
}

__global__ void seg_windows_k(int S, int64_t n_total, int64_t seg_len, int64_t lead, int64_t tail,
                              int64_t *__restrict__ starts, int *__restrict__ lens,
                              int *__restrict__ own_lo, int *__restrict__ own_hi, int map_tiles)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (map_tiles > 0 ? 64 * map_tiles : S)) return;
    const int at = seg_slot(s, map_tiles);
    if (s >= S) {                                   // an empty window behind the stream's end
        starts[at] = 0; lens[at] = 0;
        if (own_lo) own_lo[at] = 0;
        if (own_hi) own_hi[at] = 0;
        return;
    }
    int64_t st = (int64_t)s * seg_len - lead;
    if (st < 0) st = 0;
    int64_t en = (int64_t)(s + 1) * seg_len + tail;
    if (en > n_total || s == S - 1) en = n_total;
    starts[at] = st;
    lens[at] = (int)(en - st);
    // the segment's own samples inside its window (the partition the stream statistics run over)
    int64_t o_hi = (int64_t)(s + 1) * seg_len;
    if (o_hi > n_total) o_hi = n_total;
    if (own_lo) own_lo[at] = (int)((int64_t)s * seg_len - st);
    if (own_hi) own_hi[at] = (int)(o_hi - st);
}

// rows of a tiled float array of the window in slot s
__device__ __forceinline__ float tile_f32(const void *base, int out_rows, int s, int row)
{
    return ((const float *)base)[((size_t)(s >> 6) * out_rows + row) * 64 + (s & 63)];
}
__device__ __forceinline__ unsigned tile_u8(const uint8_t *base, int out_rows, int s, int row)
{
    return base[((size_t)(s >> 6) * out_rows + row) * 64 + (s & 63)];
}

// Stream position of symbol row r of the window in slot s: row 0 is the window's first sample
// (gardner2.py:59), row r >= 1 is k + mu of the iteration that produced it (gardner2.py:95), which
// the recurrence kernel wrote to row r - 1 of `pos`.  Positions grow by sps + (mu change) per
// row, so they are strictly increasing and a binary search finds the seams.
__device__ __forceinline__ double row_position(const SegParams &P, int s, int r)
{
    const double st = (double)P.starts[s];
    return r == 0 ? st : st + (double)tile_f32(P.pos, P.out_rows, s, r - 1);
}
// number of rows r in [0, n) with position < x (strict = true) or <= x
__device__ __forceinline__ int rows_below(const SegParams &P, int s, int n, double x, bool strict)
{
    int lo = 0, hi = n;                      // invariant: rows < lo are below, rows >= hi are not
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double p = row_position(P, s, mid);
        if (strict ? p < x : p <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one lane per window: the last symbol in front of the window's closing seam ends its share
__global__ __launch_bounds__(64) void seg_scan_k(SegParams P)
{
    // thread = slot, so that a wave's windows share a tile and its loads of a row coalesce
    const int at = blockIdx.x * 64 + threadIdx.x;
    if (at >= P.n_slots) return;
    const int s = seg_of_slot(at, P.map_tiles);
    if (s >= P.S) return;
    const int n = P.n_out[at];
    int j1 = n;
    if (s != P.S - 1) j1 = rows_below(P, at, n, (double)(s + 1) * (double)P.seg_len, true);
    P.j1[at] = j1;
    P.p_last[at] = j1 > 0 ? row_position(P, at, j1 - 1) : -1.0e300;
}

__global__ __launch_bounds__(64) void seg_range_k(SegParams P, int vote_len)
{
    const int at = blockIdx.x * 64 + threadIdx.x;              // thread = slot (see seg_scan_k)
    if (at >= P.n_slots) return;
    const int s = seg_of_slot(at, P.map_tiles);
    if (s >= P.S) return;
    const int before = seg_slot(s > 0 ? s - 1 : 0, P.map_tiles);
    const int j1 = P.j1[at];
    int j0 = 0, rot = 0;
    if (s > 0) {
        // the share starts with the first symbol more than half a symbol after the predecessor's last
        j0 = rows_below(P, at, j1, P.p_last[before] + 0.5 * (double)P.sps, false);
        // rows j0-1-m of this window and j1'-1-m of the predecessor are the same symbols
        const int jp = P.j1[before];
        int V = vote_len;
        if (V > j0) V = j0;
        if (V > jp) V = jp;
        // every pair of decisions agrees under exactly one of the four turns: one pass, four counters
        int agree[4] = {0, 0, 0, 0};
        for (int m0 = 0; m0 < V; m0 += 32) {                // 32 pairs of loads in flight (8: 98 us at 32768 windows)
            unsigned a[32], b[32];
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const int m = m0 + i < V ? m0 + i : V - 1;
                a[i] = tile_u8(P.dec, P.out_rows, at, j0 - 1 - m);
                b[i] = tile_u8(P.dec, P.out_rows, before, jp - 1 - m);
            }
#pragma unroll
            for (int i = 0; i < 32; i++) {
                if (m0 + i >= V) continue;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) agree[r4] += rot_quadrant(a[i], r4) == b[i];
            }
        }
        int best = -1;
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++)
            if (agree[r4] > best) { best = agree[r4]; rot = r4; }
    }
    if (j0 > j1) j0 = j1;
    P.j0[at] = j0;
    P.rot[at] = rot;
}

// exclusive scan of the share sizes and running rotation (one workgroup; S <= a few 10^5)
__global__ __launch_bounds__(1024) void seg_offsets_k(SegParams P)
{
    __shared__ long long cnt[1024];
    __shared__ int rsum[1024];
    const int t = threadIdx.x;
    const int per = (P.S + 1023) / 1024;
    const int lo = t * per, hi = (lo + per < P.S) ? lo + per : P.S;
    long long c = 0; int r = 0;
    for (int s0 = lo; s0 < hi; s0 += 8) {                   // 24 loads in flight: the slots of a slice lie far apart
        int a1[8], a0[8], ar[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int at = seg_slot(s0 + i < hi ? s0 + i : hi - 1, P.map_tiles);
            a1[i] = P.j1[at]; a0[i] = P.j0[at]; ar[i] = P.rot[at];
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (s0 + i < hi) { c += a1[i] - a0[i]; r += ar[i]; }
    }
    cnt[t] = c; rsum[t] = r;
    __syncthreads();
    // exclusive scan of the 1024 partial sums (Hillis-Steele, inclusive, then shifted by one)
    for (int d = 1; d < 1024; d <<= 1) {
        const long long c1 = t >= d ? cnt[t - d] : 0;
        const int r1 = t >= d ? rsum[t - d] : 0;
        __syncthreads();
        cnt[t] += c1; rsum[t] += r1;
        __syncthreads();
    }
    if (t == 1023) P.off[P.n_slots] = cnt[1023];
    long long a = cnt[t] - c; int b = rsum[t] - r;         // exclusive: the sums in front of this thread's slice
    for (int s0 = lo; s0 < hi; s0 += 8) {
        int a1[8], a0[8], ar[8], at[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            at[i] = seg_slot(s0 + i < hi ? s0 + i : hi - 1, P.map_tiles);
            a1[i] = P.j1[at[i]]; a0[i] = P.j0[at[i]]; ar[i] = P.rot[at[i]];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (s0 + i >= hi) break;
            b += ar[i];                      // window s is turned by the sum of the relative turns up to s
            P.off[at[i]] = a;
            a += a1[i] - a0[i];
            P.rot[at[i]] = b & 3;
        }
    }
}

// Warm-up rows of the windows from their predecessors (segment-parallel mode with the matched filter).
// Row r < lead of window w holds stream sample starts[w] + r, which is row starts[w] + r - starts[w-1]
// of window w - 1 -- one of its own (already filtered) rows when lead <= seg_len.  The filter therefore
// skips those rows in every tile but the first and this copy fills them: the same values bit for bit
// (a sample's filter sum does not depend on the window it is computed in), at a copy's cost instead of
// the filter's.  Idle lanes of the last tile are cleared.  Grid: (lead / 64 rounded up, tiles - 1).
__global__ __launch_bounds__(256) void seg_lead_copy_k(float2 *__restrict__ tiles, int rows,
                                                       const int64_t *__restrict__ starts, int S, int lead)
{
    const int lane = threadIdx.x & 63, tile = blockIdx.y + 1;
    const int w = tile * 64 + lane;
    const int64_t shift = w < S ? starts[w] - starts[w - 1] : 0;
    const float2 *src = tiles + (size_t)((w - 1) >> 6) * rows * 64 + ((w - 1) & 63);
    float2 *dst = tiles + (size_t)tile * rows * 64 + lane;
    const int r0 = blockIdx.x * 64 + (threadIdx.x >> 6);
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {                      // 64 rows per workgroup, 16 loads in flight per lane
        const int r = r0 + 4 * i;
        v[i] = (w < S && r < lead) ? src[(size_t)(shift + r) * 64] : float2{0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = r0 + 4 * i;
        if (r < lead) dst[(size_t)r * 64] = v[i];
    }
}

// Initial timing phase of every window (segment-parallel mode).  A window's timing loop starts cold;
// started half a symbol away from the true timing it sits on the unstable zero of the Gardner
// S-curve and may hang there for thousands of symbols (observed: one window in 16 384 still unlocked
// after a 4096-symbol warm-up).  So each window starts at the sampling phase p in [0, sps) whose
// strobes x[p + k sps], k < n_sym, carry the most energy -- the spectral line of |x|^2 at the symbol
// rate (Oerder-Meyr) read on the sample grid, at most half a sample from the lock point.  Lane =
// window over the tiled filter output; sums in float64, in k order (the oracle adds in the same order).
// frac: the line's phase instead of its largest sample -- the energies E_p are samples of A + B cos(2 pi
// (t - tau) / sps), so tau = atan2(sum E_p sin(2 pi p / sps), sum E_p cos(2 pi p / sps)) * sps / (2 pi) in
// [0, sps), rounded to 1/1024 sample (which makes it the oracle's number whatever the last bits of atan2 and
// of the table are).  Measured against the sequential loop's mu at that point: 0.013 sample rms over 512
// symbols (0.29 for the sample grid), so the windows' loops have 1/20 of the distance to converge over.
__device__ __forceinline__ double seg_phase_tau(double cs, double sn, int sps)
{
    double tau = atan2(sn, cs) * (double)sps / 6.283185307179586;
    if (tau < 0.0) tau += (double)sps;
    const double q = floor(tau * 1024.0 + 0.5) / 1024.0;
    return q >= (double)sps ? q - (double)sps : q;
}
__device__ __forceinline__ double seg_phase_pick(const double *e, int stride, int sps, int frac)
{
    if (!frac) {
        double best = -1.0;
        int arg = 0;
        for (int p = 0; p < sps; p++)
            if (e[p * stride] > best) { best = e[p * stride]; arg = p; }
        return (double)arg;
    }
    double cs = 0.0, sn = 0.0;
    for (int p = 0; p < sps; p++) {
        const double t = (double)p / (double)sps;
        cs += e[p * stride] * cospi(2.0 * t);               // exact at the quarter turns
        sn += e[p * stride] * sinpi(2.0 * t);
    }
    return seg_phase_tau(cs, sn, sps);
}

// Waves per sampling phase: the fractional estimate reads twice the rows, so its strobes are summed in
// `parts` contiguous parts of ceil(ns / parts) each (k order inside a part, the parts added first to last) by
// as many waves -- part of its definition, the oracle adds the same way.  The sample grid's sum is one part.
__host__ __device__ inline int seg_phase_parts(int sps, int frac) { return !frac ? 1 : (sps <= 4 ? 4 : (sps <= 8 ? 2 : 1)); }

__global__ __launch_bounds__(1024) void seg_phase_k(const float2 *__restrict__ tiles, int rows,
                                                    const int *__restrict__ lens, int S, int sps, int n_sym,
                                                    double *__restrict__ mu0, const float2 *__restrict__ tile0, int frac)
{
    // one wave per (sampling phase, part) (sps <= 8), each adding its strobes in k order, 16 loads in flight;
    // wave 0 then adds the parts and takes the first maximum over the phases, or the line's phase
    __shared__ double e[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tile = blockIdx.x, w = tile * 64 + lane;
    const int len = w < S ? lens[w] : 0;
    const int ns = n_sym < len / sps ? n_sym : len / sps;
    // (rows = rows between consecutive tiles' row 0; tile 0 may live elsewhere: SyncParams x_tile_rows / x_tile0)
    const float2 *base = (tile == 0 && tile0 ? tile0 : tiles + (ptrdiff_t)tile * rows * 64) + lane;
    if (sps <= 8) {
        const int parts = seg_phase_parts(sps, frac), ph = wv / parts, part = wv - ph * parts;
        const int plen = (ns + parts - 1) / parts;
        const int k_lo = part * plen, k_hi = k_lo + plen < ns ? k_lo + plen : ns;
        double acc = 0.0;
        for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
            float2 v[16];
#pragma unroll
            for (int kk = 0; kk < 16; kk++)
                v[kk] = k0 + kk < k_hi ? base[(size_t)((k0 + kk) * sps + ph) * 64] : float2{0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 16; kk++)
                if (k0 + kk < k_hi) acc += (double)v[kk].x * (double)v[kk].x + (double)v[kk].y * (double)v[kk].y;
        }
        e[wv][lane] = acc;
        __syncthreads();
        if (wv == 0 && w < S) {
            double ep[8];
            for (int p = 0; p < sps; p++) {
                double t = e[p * parts][lane];
                for (int q = 1; q < parts; q++) t += e[p * parts + q][lane];
                ep[p] = t;
            }
            mu0[w] = seg_phase_pick(ep, 1, sps, frac);
        }
    } else if (wv == 0 && w < S) {
        // one lane per window, phase after phase (the sample grid: first maximum)
        double best = -1.0, cs = 0.0, sn = 0.0;
        int arg = 0;
        for (int p = 0; p < sps; p++) {
            double acc = 0.0;
            for (int k = 0; k < ns; k++) {
                const float2 v = base[(size_t)(k * sps + p) * 64];
                acc += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
            }
            if (acc > best) { best = acc; arg = p; }
            const double t = (double)p / (double)sps;
            cs += acc * cospi(2.0 * t);
            sn += acc * sinpi(2.0 * t);
        }
        mu0[w] = frac ? seg_phase_tau(cs, sn, sps) : (double)arg;
    }
}

// The kept rows [j0, j1) of every window, rotated and concatenated.  A workgroup takes 64 rows of one
// tile -- one contiguous piece of each tiled array -- reads it with 16-byte loads (the next array's loads
// are in flight while the current one goes out), turns it through LDS ([lane][row], one array after the
// other in the same 17 KB) and writes, window by window, 64 consecutive outputs per wave (measured at
// 2^30 samples: 64 rows 0.80 ms, 128 rows 0.97, 256 rows 1.35; round 1s 64 x 64 blocks with 4-byte loads 1.25).
constexpr int GATHER_ROWS = 64;
constexpr int GATHER_F4 = GATHER_ROWS * 16 / 256;            // 16-byte pieces of a 4-byte array per thread

// rows [r_first, r_first + GATHER_ROWS) of a tile's 4-byte array; rows outside [0, n_rows) read as 0
__device__ __forceinline__ void gather_load_f32(const float *__restrict__ tile_base, int r_first, int n_rows,
                                                float4 (&v)[GATHER_F4])
{
#pragma unroll
    for (int k = 0; k < GATHER_F4; k++) {
        const int c = threadIdx.x + 256 * k;                 // piece c: row c / 16, lanes 4 * (c % 16) ..
        const int row = r_first + (c >> 4);
        v[k] = (row >= 0 && row < n_rows) ? *(const float4 *)(tile_base + (ptrdiff_t)row * 64 + (c & 15) * 4)
                                          : float4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void gather_turn_f32(const float4 (&v)[GATHER_F4], float *lds)    // lds[lane][GATHER_ROWS + 1]
{
#pragma unroll
    for (int k = 0; k < GATHER_F4; k++) {
        const int c = threadIdx.x + 256 * k, i = c >> 4, l = (c & 15) * 4;
        lds[(l + 0) * (GATHER_ROWS + 1) + i] = v[k].x; lds[(l + 1) * (GATHER_ROWS + 1) + i] = v[k].y;
        lds[(l + 2) * (GATHER_ROWS + 1) + i] = v[k].z; lds[(l + 3) * (GATHER_ROWS + 1) + i] = v[k].w;
    }
}

__global__ __launch_bounds__(256) void seg_gather_k(SegParams P, uint8_t *__restrict__ dec_out,
                                                    float *__restrict__ mu_out, float *__restrict__ freq_out,
                                                    int64_t capacity)
{
    __shared__ float buf[64 * (GATHER_ROWS + 1)];
    __shared__ int s_j0[64], s_j1[64], s_rot[64];
    __shared__ long long s_off[64];
    const int tile = blockIdx.y, r0 = blockIdx.x * GATHER_ROWS;
    const int lane = threadIdx.x & 63;
    {
        // rows no window of this tile keeps (the warm-up rows, mostly): nothing to read or write
        const int sw = tile * 64 + lane;
        const bool real = sw < P.n_slots && seg_of_slot(sw, P.map_tiles) < P.S;
        int lo = real ? P.j0[sw] : 0x7fffffff, hi = real ? P.j1[sw] : 0;
        if (threadIdx.x < 64) {
            s_j0[lane] = real ? lo : 0; s_j1[lane] = real ? hi : 0;      // an empty range for the idle slots
            s_rot[lane] = real ? P.rot[sw] : 0; s_off[lane] = real ? P.off[sw] : 0;
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const int ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
            lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
        }
        if (r0 + GATHER_ROWS <= lo || r0 >= hi) return;      // uniform over the workgroup
    }
    const int rr = threadIdx.x % GATHER_ROWS, row = r0 + rr, bb0 = threadIdx.x / GATHER_ROWS;
    constexpr int BBS = 256 / GATHER_ROWS;
    const size_t tile_at = (size_t)tile * P.out_rows * 64;
    // ---- all loads of the decisions and of mu first ----
    constexpr int ND = GATHER_ROWS * 4 / 256;                // 16-byte pieces of the decisions per thread
    uint4 vd[ND];
    float4 vf[GATHER_F4];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        const int c = threadIdx.x + 256 * k, r = r0 + (c >> 2);
        vd[k] = r < P.out_rows ? *(const uint4 *)(P.dec + tile_at + (size_t)r * 64 + (c & 3) * 16) : uint4{0, 0, 0, 0};
    }
    // mu history row r - 1 belongs to symbol row r
    if (mu_out) gather_load_f32((const float *)P.mu + tile_at, r0 - 1, P.out_rows, vf);
    else if (freq_out) gather_load_f32((const float *)P.freq + tile_at, r0, P.out_rows, vf);
    {
        uint8_t *lds8 = (uint8_t *)buf;                       // [lane][GATHER_ROWS + 4]
#pragma unroll
        for (int k = 0; k < ND; k++) {
            const int c = threadIdx.x + 256 * k, i = c >> 2, l = (c & 3) * 16;
            const unsigned w[4] = {vd[k].x, vd[k].y, vd[k].z, vd[k].w};
#pragma unroll
            for (int j = 0; j < 16; j++) lds8[(l + j) * (GATHER_ROWS + 4) + i] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
        }
        __syncthreads();
        for (int bb = bb0; bb < 64; bb += BBS) {
            const int j0 = s_j0[bb], j1 = s_j1[bb];
            if (row < j0 || row >= j1) continue;
            const int64_t at = s_off[bb] + (row - j0);
            if (at < capacity) dec_out[at] = (uint8_t)rot_quadrant(lds8[bb * (GATHER_ROWS + 4) + rr], s_rot[bb]);
        }
        __syncthreads();
    }
    if (mu_out) {
        gather_turn_f32(vf, buf);
        if (freq_out) gather_load_f32((const float *)P.freq + tile_at, r0, P.out_rows, vf);
        __syncthreads();
        for (int bb = bb0; bb < 64; bb += BBS) {
            const int j0 = s_j0[bb], j1 = s_j1[bb];
            if (row < j0 || row >= j1) continue;
            const int64_t at = s_off[bb] + (row - j0);
            if (at < capacity) mu_out[at] = buf[bb * (GATHER_ROWS + 1) + rr];
        }
        __syncthreads();
    }
    if (freq_out) {
        gather_turn_f32(vf, buf);
        __syncthreads();
        for (int bb = bb0; bb < 64; bb += BBS) {
            const int j0 = s_j0[bb], j1 = s_j1[bb];
            if (row < j0 || row >= j1) continue;
            const int64_t at = s_off[bb] + (row - j0);
            if (at < capacity) freq_out[at] = buf[bb * (GATHER_ROWS + 1) + rr];
        }
    }
}

// lagrange_resample(signal, positions, half_len) -- hw1/test_lagrange_resampling.py:34-72:
// one thread per position, the (2h+1)-point barycentric form with the reference's index
// clipping and exact-node shortcut (sync_lane.h lagrange_at, the interpolator the timing loop
// uses when interp_half_len > 0).
template <typename T>
struct VecFetch {
    const T *x;
    __device__ __forceinline__ dspm::cd operator()(int i) const
    {
        dspm::cd z; z.re = (double)x[i].x; z.im = (double)x[i].y; return z;
    }
};
template <typename T>
__global__ void lagrange_resample_k(const T *__restrict__ x, int len, const double *__restrict__ pos,
                                    int64_t P, int half_len, double2 *__restrict__ out)
{
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= P) return;
    dspm::LagrangeW L;
    dspm::lagrange_weights(half_len, L);
    dspm::cd z = dspm::lagrange_at(pos[p], len, L, VecFetch<T>{x});
    out[p] = make_double2(z.re, z.im);
}

}  // namespace

hipError_t launch_lagrange_resample(const void *x, int c128, int len, const double *pos, int64_t P, int half_len,
                                    void *out, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    const unsigned nb = (unsigned)((P + 255) / 256);
    if (c128) hipLaunchKernelGGL(lagrange_resample_k<double2>, dim3(nb), dim3(256), 0, s, (const double2 *)x, len, pos, P,
                                 half_len, (double2 *)out);
    else hipLaunchKernelGGL(lagrange_resample_k<float2>, dim3(nb), dim3(256), 0, s, (const float2 *)x, len, pos, P,
                            half_len, (double2 *)out);
    return hipGetLastError();
}

hipError_t launch_seg_windows(int S, int64_t n_total, int64_t seg_len, int64_t lead, int64_t tail,
                              int64_t *starts, int *lens, int *own_lo, int *own_hi, hipStream_t s, int map_tiles)
{
    const int n = map_tiles > 0 ? 64 * map_tiles : S;
    hipLaunchKernelGGL(seg_windows_k, dim3((n + 255) / 256), dim3(256), 0, s, S, n_total, seg_len, lead, tail,
                       starts, lens, own_lo, own_hi, map_tiles);
    return hipGetLastError();
}

// Lane-major segment tiles: stream sample i is row i % seg_len of lane (i / seg_len) / nt of tile
// (i / seg_len) % nt.  Row r < win_rows of lane l of `tile0` = window sample r of slot l (zero past the
// window's end); row r < pad_rows behind the last tile = the sample that follows lane l's last segment.
static __global__ __launch_bounds__(256) void seg_edge_tiles_k(float2 *tiles, int nt, int64_t seg_len, int64_t n_total,
                                                        const int64_t *__restrict__ starts, const int *__restrict__ lens,
                                                        float2 *__restrict__ tile0, int win_rows, int pad_rows)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= win_rows + pad_rows) return;
    int64_t i;
    bool in;
    if (r < win_rows) { i = starts[lane] + r; in = r < lens[lane]; }
    else { i = ((int64_t)lane * nt + nt) * seg_len + (r - win_rows); in = true; }
    float2 v = float2{0.f, 0.f};
    if (in && i < n_total) {
        const int64_t g = i / seg_len;
        v = tiles[((size_t)(g % nt) * seg_len + (size_t)(i - g * seg_len)) * 64 + (size_t)(g / nt)];
    }
    if (r < win_rows) tile0[(size_t)r * 64 + lane] = v;
    else tiles[((size_t)nt * seg_len + (size_t)(r - win_rows)) * 64 + lane] = v;
}

hipError_t launch_seg_edge_tiles(void *tiles, int map_tiles, int64_t seg_len, int64_t n_total, const int64_t *starts,
                                 const int *lens, void *tile0, int win_rows, int pad_rows, hipStream_t s)
{
    hipLaunchKernelGGL(seg_edge_tiles_k, dim3((win_rows + pad_rows + 3) / 4), dim3(256), 0, s, (float2 *)tiles, map_tiles,
                       seg_len, n_total, starts, lens, (float2 *)tile0, win_rows, pad_rows);
    return hipGetLastError();
}

hipError_t launch_seg_lead_copy(void *tiles, int rows, const int64_t *starts, int S, int lead, hipStream_t s)
{
    const int ntiles = (S + 63) / 64;
    if (ntiles < 2 || lead <= 0) return hipSuccess;
    hipLaunchKernelGGL(seg_lead_copy_k, dim3((lead + 63) / 64, ntiles - 1), dim3(256), 0, s, (float2 *)tiles, rows, starts, S,
                       lead);
    return hipGetLastError();
}

hipError_t launch_seg_phase(const void *tiles, int rows, const int *lens, int S, int sps, int n_sym, double *mu0,
                            hipStream_t s, int x_tile_rows, const void *x_tile0, int frac)
{
    hipLaunchKernelGGL(seg_phase_k, dim3((S + 63) / 64), dim3(sps <= 8 ? 64 * sps * seg_phase_parts(sps, frac) : 64), 0, s,
                       (const float2 *)tiles,
                       x_tile_rows ? x_tile_rows : rows, lens, S, sps, n_sym, mu0, (const float2 *)x_tile0, frac);
    return hipGetLastError();
}

hipError_t launch_seg_stitch(const SegParams &P, int overlap_vote, hipStream_t s)
{
    const int nb = (P.n_slots + 63) / 64;
    hipLaunchKernelGGL(seg_scan_k, dim3(nb), dim3(64), 0, s, P);
    hipLaunchKernelGGL(seg_range_k, dim3(nb), dim3(64), 0, s, P, overlap_vote);
    hipLaunchKernelGGL(seg_offsets_k, dim3(1), dim3(1024), 0, s, P);
    return hipGetLastError();
}

hipError_t launch_seg_gather(const SegParams &P, uint8_t *dec_out, float *mu_out, float *freq_out,
                             int64_t capacity, hipStream_t s)
{
    dim3 grid((P.out_rows + GATHER_ROWS - 1) / GATHER_ROWS, (P.n_slots + 63) / 64);
    hipLaunchKernelGGL(seg_gather_k, grid, dim3(256), 0, s, P, dec_out, mu_out, freq_out, capacity);
    return hipGetLastError();
}

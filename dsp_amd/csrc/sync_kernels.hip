// sync_kernels.hip -- the receiver synchronisation recurrences on gfx950.
//
// One wavefront (one 64-thread workgroup) owns one tile of 64 bursts, one
// burst per lane, and walks the Gardner timing loop (gardner2.py:63-98 or
// hw1/gardner.py:45-72) and the decision-directed FLL (fll_func.py:14-26)
// symbol by symbol with all loop state in VGPRs (float64).
//
// Memory system design (CDNA4):
//   * the tile's input is ONE contiguous stream of 64-lane rows; the wave
//     streams it into a private LDS ring with LDS-DMA (global_load_lds_dwordx4,
//     1 KiB per instruction, no VGPR staging) several symbols ahead of use;
//   * each lane then picks the rows its own fractional delay needs out of the
//     ring with ds_read (bank-conflict free: lane l always reads column l);
//   * outputs leave as full-row stores (64 lanes x 4/8/16 B contiguous).
//   The DMA and the stores are issued through inline asm so that their count
//   in the in-order vmcnt queue is exact: once per block of T symbols the wave
//   waits with s_waitcnt vmcnt(N), N = the stores issued since the previous
//   block's DMA, which retires that DMA while leaving the stores in flight.
//   Lanes whose rows fall outside the ring window (symbol slips, tails) read
//   global memory directly; that path is rare and keeps every input correct.
#include "dsp_internal.h"
#include "sync_lane.h"

#pragma clang fp contract(off)

using dspm::cd;

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

// ---- asm VMEM helpers (exactly one vector-memory instruction each) ---------
__device__ __forceinline__ void lds_dma_16B(const void *g, unsigned lds_off)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void st_b8(uint8_t *p, unsigned v)
{
    asm volatile("global_store_byte %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_val(float *p, double v)
{
    float f = (float)v;
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(f) : "memory");
}
__device__ __forceinline__ void st_val(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_cplx(f2v *p, double re, double im)
{
    f2v v = {(float)re, (float)im};
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_cplx(d2v *p, double re, double im)
{
    d2v v = {re, im};
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

#define WCASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (n) {
        WCASE(1) WCASE(2) WCASE(3) WCASE(4) WCASE(5) WCASE(6) WCASE(7) WCASE(8)
        WCASE(9) WCASE(10) WCASE(11) WCASE(12) WCASE(13) WCASE(14) WCASE(15) WCASE(16)
        WCASE(17) WCASE(18) WCASE(19) WCASE(20) WCASE(21) WCASE(22) WCASE(23) WCASE(24)
        WCASE(25) WCASE(26) WCASE(27) WCASE(28) WCASE(29) WCASE(30) WCASE(31) WCASE(32)
        WCASE(33) WCASE(34) WCASE(35) WCASE(36) WCASE(37) WCASE(38) WCASE(39) WCASE(40)
        WCASE(41) WCASE(42) WCASE(43) WCASE(44) WCASE(45) WCASE(46) WCASE(47) WCASE(48)
        WCASE(49) WCASE(50) WCASE(51) WCASE(52) WCASE(53) WCASE(54) WCASE(55) WCASE(56)
        WCASE(57) WCASE(58) WCASE(59) WCASE(60) WCASE(61) WCASE(62) WCASE(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// transcendental constants: loaded once per lane through an opaque index so that they
// stay in VGPRs (see dsp_math.h)
__device__ const double k_math_table[dspm::K_COUNT] = DSP_MATH_TABLE_INIT;

__device__ __forceinline__ dspm::MathK load_math_consts()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    dspm::MathK k;
#pragma unroll
    for (int i = 0; i < dspm::K_COUNT; i++) k.v[i] = k_math_table[z + i];
    return k;
}

template <typename IN> struct InT;
template <> struct InT<f2v> { typedef float S; typedef float OS; typedef f2v OC; enum { RPP = 2 }; };
template <> struct InT<d2v> { typedef double S; typedef double OS; typedef d2v OC; enum { RPP = 1 }; };

// ---------------------------------------------------------------------------
template <typename IN, int VARIANT, bool FLL, bool LAGR>
__global__ __launch_bounds__(64) void sync_kernel(SyncParams P)
{
    typedef typename InT<IN>::OS OS;     // output scalar (float for c64 in, double for c128 in)
    typedef typename InT<IN>::OC OC;     // output complex
    constexpr int RPP = InT<IN>::RPP;    // rows per 1 KiB DMA piece

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int b = tile * 64 + lane;
    const bool lane_ok = b < P.B;
    const int len = lane_ok ? (P.lens ? P.lens[b] : P.len_all) : 0;
    const double scale = (lane_ok && P.scale) ? P.scale[b] : 1.0;
    const int in_rows = P.in_rows, out_rows = P.out_rows, sps = P.sps;
    const IN *xt = (const IN *)P.x + (size_t)tile * in_rows * 64;
    const size_t obase = (size_t)tile * out_rows * 64 + lane;

    const int R = P.ring_rows, RM = R - 1, T = P.T;
    const bool ring = P.use_ring != 0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    int win_lo = 0, win_hi = 0;          // rows [win_lo, win_hi) are valid in the ring
    int F = 0;                           // DMA front: rows [0, F) have been issued

    auto fetch = [&](int row) -> cd {
        IN v;
        if ((unsigned)(row - win_lo) < (unsigned)(win_hi - win_lo)) {
            v = ((const IN *)smem)[(row & RM) * 64 + lane];
        } else {
            int r = row < 0 ? 0 : (row >= in_rows ? in_rows - 1 : row);
            v = xt[(size_t)r * 64 + lane];
            asm volatile("" : "+v"(v));          // keep the load's wait inside this branch
        }
        cd z;
        z.re = (double)v.x * scale;
        z.im = (double)v.y * scale;
        return z;
    };
    auto issue = [&](int F_new) {
        F_new = F_new < in_rows ? F_new : in_rows;
        for (int r = F; r < F_new; r += RPP) {
            const unsigned char *g = (const unsigned char *)(xt + (size_t)r * 64) + lane * 16;
            lds_dma_16B(g, lds_base + (unsigned)((r & RM) * 64 * (int)sizeof(IN)));
        }
        F = F_new > F ? F_new : F;
    };

    if (ring) {                          // prologue: rows for block 0
        issue(sps * T + P.C);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        win_hi = F;
    }

    dspm::LagrangeW LW;
    const dspm::LagrangeW *Lp = nullptr;
    if (LAGR) { dspm::lagrange_weights(P.half_len, LW); Lp = &LW; }

    // ---- per-lane loop state ------------------------------------------------
    dspm::GState gs;
    gs.mu = (VARIANT == DSP_VARIANT_GARDNER2) ? P.mu0 : 0.0;
    gs.k = (VARIANT == DSP_VARIANT_GARDNER2) ? sps : 0;
    gs.active = lane_ok && len > 0;
    int n_symbols = 0;                   // ted: (len - 2*sps)//sps (floor)
    if (VARIANT == DSP_VARIANT_TED) {
        int ns = len - 2 * sps;
        n_symbols = ns >= 0 ? ns / sps : -((-ns + sps - 1) / sps);
        gs.active = lane_ok;
    }
    double freq = 0.0, integ = 0.0, i0 = 0.0;
    if (FLL && lane_ok) {
        if (P.fll_in) { freq = P.fll_in[2 * b]; integ = P.fll_in[2 * b + 1]; }
        if (P.i0) i0 = P.i0[b];
    }
    const double kp = P.kp, ki = P.ki, alpha = P.alpha;
    dspm::MathK MK;
    if (FLL) MK = load_math_consts();

    OC *o_rec = (OC *)P.rec; OS *o_terr = (OS *)P.terr; OS *o_mu = (OS *)P.mu;
    OC *o_sig = (OC *)P.sig; OS *o_freq = (OS *)P.freq; OS *o_ef = (OS *)P.ef;
    uint8_t *o_dec = P.dec;

    // first symbol: recovered[0] = signal[0] (gardner2.py:59) / x[0] for the FLL-only form
    constexpr int ROFF = (VARIANT == DSP_VARIANT_TED) ? 1 : 0;   // FLL lags the TED by one row
    bool pending = false;
    cd rec_prev; rec_prev.re = 0.0; rec_prev.im = 0.0;
    int n_rec = 0;
    if (VARIANT != DSP_VARIANT_TED) {
        if (gs.active) { rec_prev = fetch(0); pending = true; n_rec = 1; }
        if (VARIANT == DSP_VARIANT_GARDNER2 && o_rec) st_cplx(o_rec + obase, rec_prev.re, rec_prev.im);
    }
    if (VARIANT == DSP_VARIANT_NONE) gs.active = false;
    bool overflow = false;

    // straight-line LDS access for the fast path (row known to be inside the ring window)
    auto lds_at = [&](int row) -> cd {
        IN v = ((const IN *)smem)[(row & RM) * 64 + lane];
        cd z;
        z.re = (double)v.x * scale;
        z.im = (double)v.y * scale;
        return z;
    };
    const double spsd = (double)sps, lend = (double)len;
    const double halfd = (VARIANT == DSP_VARIANT_TED) ? spsd / 2.0 : (double)(sps / 2);

    int it = 0;
    bool running = true;
    for (int blk = 0; running; blk++) {
        if (ring) {
            if (blk > 0) wait_vmcnt(P.nwait);   // the previous block's DMA has landed
            win_hi = F;
            issue(sps * (blk + 2) * T + P.C);
            win_lo = F - R > 0 ? F - R : 0;
        }
        for (int t = 0; t < T; t++, it++) {
            bool more = (VARIANT == DSP_VARIANT_NONE) ? pending : (gs.active || (FLL && pending));
            if (!__any(more)) { running = false; break; }
            const int rec_row = (VARIANT == DSP_VARIANT_TED) ? it : it + 1;
            const int fll_row = it - ROFF;
            if (fll_row >= out_rows) { running = false; break; }      // capacity (uniform)

            // ---- FLL on the symbol appended by the previous iteration: computed by every
            //      lane, committed by select (no divergent control flow) ----------------
            double f_sr = 0.0, f_si = 0.0, f_ef = 0.0, f_freq = 0.0;
            unsigned f_quad = 0;
            if (FLL) {
                double nf = freq, ni = integ;
                dspm::FllOut fo = dspm::fll_step(MK, rec_prev.re, rec_prev.im, (double)fll_row + i0,
                                                 nf, ni, kp, ki);
                freq = pending ? nf : freq;
                integ = pending ? ni : integ;
                f_sr = pending ? fo.sr : 0.0; f_si = pending ? fo.si : 0.0;
                f_ef = pending ? fo.ef : 0.0; f_freq = pending ? fo.freq : 0.0;
                f_quad = pending ? (unsigned)fo.quad : 0u;
            }

            // ---- timing loop: one pass of the reference's loop body --------
            dspm::GOut go; go.produced = false; go.rec.re = 0.0; go.rec.im = 0.0; go.err = 0.0; go.mu = 0.0;
            if (VARIANT == DSP_VARIANT_NONE) {
                const bool have = pending && (it + 1 < len);
                if (it + 1 >= win_lo && it + 1 < win_hi) {            // uniform
                    cd v = lds_at(it + 1);
                    go.rec.re = have ? v.re : 0.0; go.rec.im = have ? v.im : 0.0;
                } else if (have) {
                    go.rec = fetch(it + 1);
                }
                go.produced = have;
            } else {
                if (gs.active && rec_row >= out_rows) { gs.active = false; overflow = true; }
                bool fast = !LAGR;
                bool ok = false;
                double mu = gs.mu, i_a = 0.0, i_b = 0.0, i_c = 0.0;
                int r_a = 0, r_b = 0, r_c = 0;
                if (!LAGR) {
                    if (VARIANT == DSP_VARIANT_GARDNER2) {
                        // gardner2.py:63-80 without the wrap loops (a lane that needs them, or
                        // rows outside the ring, sends the wave down the generic path)
                        const bool cont = gs.active && (gs.k <= len - sps);
                        const double kd = (double)gs.k;
                        const bool odd = cont && !(mu < spsd && mu >= -spsd);   // wrap or NaN
                        i_a = (kd - spsd) + mu;           // idx_early
                        i_b = (kd - halfd) + mu;          // idx_mid
                        i_c = kd + mu;                    // idx_late
                        ok = cont && !(i_a < 0.0) && !(i_c + 1.0 >= lend);
                        r_a = __double2int_rz(i_a); r_b = __double2int_rz(i_b); r_c = __double2int_rz(i_c);
                        const bool inwin = (r_a >= win_lo) && (r_c + 1 < win_hi);
                        fast = !__any(odd || (ok && !inwin));
                    } else {
                        // hw1/gardner.py:45-53
                        const bool cont = gs.active && (gs.k < n_symbols);
                        const double base = (double)(gs.k * sps);
                        i_a = base + mu;                                   // idx_k
                        i_c = (double)((gs.k + 1) * sps) + mu;             // idx_next
                        i_b = base + mu + halfd;                           // idx_mid
                        ok = cont && !(i_c + 1.0 >= lend || i_b + 1.0 >= lend);
                        const double fa = dspm::d_floor(i_a), fb = dspm::d_floor(i_b), fc = dspm::d_floor(i_c);
                        r_a = __double2int_rz(fa); r_b = __double2int_rz(fb); r_c = __double2int_rz(fc);
                        const bool inwin = (r_a >= win_lo) && (r_c + 1 < win_hi) && (r_a >= 0);
                        fast = !__any(ok && !inwin);
                    }
                }
                if (fast) {
                    cd a0 = lds_at(r_a), a1 = lds_at(r_a + 1);
                    cd b0 = lds_at(r_b), b1 = lds_at(r_b + 1);
                    cd c0 = lds_at(r_c), c1 = lds_at(r_c + 1);
                    cd xa = dspm::lerp(a0, a1, i_a - (double)r_a);
                    cd xb = dspm::lerp(b0, b1, i_b - (double)r_b);
                    cd xc = dspm::lerp(c0, c1, i_c - (double)r_c);
                    const double e = dspm::ted_error(xa, xb, xc);
                    cd rec; rec.re = 0.0; rec.im = 0.0;
                    double mu_n, mu_hist;
                    if (VARIANT == DSP_VARIANT_GARDNER2) {
                        mu_n = mu - alpha * e;                              // :91
                        mu_hist = mu_n;
                        const double i_o = (double)gs.k + mu_n;             // :95
                        const int r_o = __double2int_rz(i_o);
                        const int hi_lim = win_hi < len ? win_hi : len;
                        const bool ofast = (i_o > -1.0e9 && i_o < 1.0e9) && r_o >= win_lo && r_o + 1 < hi_lim;
                        if (!__any(ok && !ofast)) {
                            rec = dspm::lerp(lds_at(r_o), lds_at(r_o + 1), i_o - (double)r_o);
                        } else if (ok) {
                            rec = dspm::g2_interp(i_o, len, (const dspm::LagrangeW *)nullptr, fetch);
                        }
                        if (ok) gs.k += sps;                                // :98
                    } else {
                        mu_hist = mu;                                       // :63
                        rec = xa;                                           // :72
                        mu_n = mu + alpha * e;                              // :66
                        mu_n = mu_n < -spsd ? -spsd : mu_n;                 // :69
                        mu_n = mu_n > spsd ? spsd : mu_n;
                        if (ok) gs.k += 1;
                    }
                    gs.mu = ok ? mu_n : mu;
                    gs.active = ok;
                    go.produced = ok;
                    go.err = ok ? e : 0.0;
                    go.mu = ok ? mu_hist : 0.0;
                    go.rec.re = ok ? rec.re : 0.0;
                    go.rec.im = ok ? rec.im : 0.0;
                } else {
                    if (VARIANT == DSP_VARIANT_GARDNER2)
                        go = dspm::gardner2_iter(gs, len, sps, alpha, Lp, fetch);
                    else
                        go = dspm::ted_iter(gs, len, sps, n_symbols, alpha, Lp, fetch);
                }
            }
            // ---- stores: unconditional, one instruction per enabled output -
            const int frow = fll_row < 0 ? 0 : fll_row;
            const size_t fo_off = obase + (size_t)frow * 64;
            if (FLL) {
                if (o_sig) st_cplx(o_sig + fo_off, f_sr, f_si);
                if (o_freq) st_val(o_freq + fo_off, f_freq);
                if (o_ef) st_val(o_ef + fo_off, f_ef);
                if (o_dec) st_b8(o_dec + fo_off, f_quad);
            }
            if (VARIANT != DSP_VARIANT_NONE) {
                const int rrow = rec_row < out_rows ? rec_row : out_rows - 1;
                const size_t g_off = obase + (size_t)it * 64;
                if (o_rec) st_cplx(o_rec + obase + (size_t)rrow * 64, go.rec.re, go.rec.im);
                if (o_terr) st_val(o_terr + g_off, go.err);
                if (o_mu) st_val(o_mu + g_off, go.mu);
            }
            pending = go.produced;
            rec_prev = go.rec;
            n_rec += go.produced ? 1 : 0;
        }
    }
    if (lane_ok) {
        if (VARIANT != DSP_VARIANT_NONE && gs.active) overflow = true;   // stopped by capacity
        if (FLL && pending) overflow = true;
        if (P.n_out) P.n_out[b] = n_rec;
        if (P.status) P.status[b] = overflow ? 1 : 0;
        if (FLL && P.fll_out) { P.fll_out[2 * b] = freq; P.fll_out[2 * b + 1] = integ; }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Ring schedule.  Iteration `it` of the timing loop touches rows
// [sps*(it-1) - h, sps*(it+2) + 2 + h); block b (T iterations) starts with rows
// < sps*(b+1)*T + C landed and issues rows < sps*(b+2)*T + C, which needs
// ring_rows >= sps*(2T+1) + C + h (+ slack for lanes that slipped a symbol).
SyncPlan sync_plan(int elem_bytes, int sps, int half_len, int variant, int n_store, int ring_rows_req)
{
    SyncPlan pl;
    int R = ring_rows_req > 0 ? ring_rows_req : 64;
    int rpp = 1024 / (64 * elem_bytes);
    int step = (variant == DSP_VARIANT_NONE) ? 1 : sps;
    int h = half_len;
    int C = step + 2 + h + 2;
    C += C & 1;
    int slack = (variant == DSP_VARIANT_NONE) ? 0 : step;
    int T = ((R - C - h - slack) / step - 1) / 2;
    if (T > 8) T = 8;
    if (n_store > 0) while (T > 1 && T * n_store > 63) T--;
    if ((step * T) % rpp) T -= 1;                 // whole DMA pieces per block
    pl.ring_rows = R;
    pl.C = C;
    pl.T = T;
    pl.use_ring = (T >= 1 && T * n_store <= 63 && step * (2 * T + 1) + C + h <= R) ? 1 : 0;
    if (!pl.use_ring) pl.T = 8;
    pl.nwait = pl.T * n_store;
    pl.lds_bytes = pl.use_ring ? R * 64 * elem_bytes : 0;
    return pl;
}

template <typename IN, int VARIANT, bool FLL>
static hipError_t launch_v(const SyncParams &P, size_t lds, hipStream_t s)
{
    int ntiles = (P.B + 63) / 64;
    if (P.half_len > 0)
        hipLaunchKernelGGL((sync_kernel<IN, VARIANT, FLL, true>), dim3(ntiles), dim3(64), lds, s, P);
    else
        hipLaunchKernelGGL((sync_kernel<IN, VARIANT, FLL, false>), dim3(ntiles), dim3(64), lds, s, P);
    return hipGetLastError();
}

template <typename IN>
static hipError_t launch_in(const SyncParams &P, int variant, bool fll, size_t lds, hipStream_t s)
{
    switch (variant) {
        case DSP_VARIANT_GARDNER2:
            return fll ? launch_v<IN, DSP_VARIANT_GARDNER2, true>(P, lds, s)
                       : launch_v<IN, DSP_VARIANT_GARDNER2, false>(P, lds, s);
        case DSP_VARIANT_TED:
            return fll ? launch_v<IN, DSP_VARIANT_TED, true>(P, lds, s)
                       : launch_v<IN, DSP_VARIANT_TED, false>(P, lds, s);
        case DSP_VARIANT_NONE:
            return launch_v<IN, DSP_VARIANT_NONE, true>(P, lds, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_sync(const SyncParams &P, bool c128, int variant, bool fll, size_t lds_bytes,
                       hipStream_t stream)
{
    if (P.B <= 0) return hipSuccess;
    return c128 ? launch_in<d2v>(P, variant, fll, lds_bytes, stream)
                : launch_in<f2v>(P, variant, fll, lds_bytes, stream);
}

// dsp_math.h -- float64 building blocks of the Gardner / FLL recurrences.
//
// Shared by the HIP kernels (device) and by the host unit tests that check the
// same source against the oracle without a GPU (tests/hostmath).  All
// arithmetic that mirrors the reference is written so that every product and
// sum is rounded separately (the reference is CPython / numpy scalar code with
// no FMA): contraction is switched OFF for the whole translation unit and
// fma() appears only inside the polynomial kernels of sincos / atan, where the
// reference calls libm.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#if defined(__HIPCC__)
#define DSP_HD __host__ __device__ __forceinline__
#else
#define DSP_HD static inline
#endif

namespace dspm {

struct cd { double re, im; };

DSP_HD double d_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
DSP_HD double d_abs(double a) { return __builtin_fabs(a); }
DSP_HD double d_rint(double a) { return __builtin_rint(a); }
DSP_HD double d_floor(double a) { return __builtin_floor(a); }
DSP_HD double d_copysign(double a, double b) { return __builtin_copysign(a, b); }

// ---------------------------------------------------------------------------
// Constants of the transcendental kernels, kept in one table so that the HIP
// kernels can hold them in VGPRs for the whole loop (64-bit literals would
// otherwise be rebuilt with two scalar moves per use, every iteration).
enum {
    K_TWO_OVER_PI, K_P1, K_P2, K_P3,
    K_S1, K_S2, K_S3, K_S4, K_S5, K_S6,
    K_C1, K_C2, K_C3, K_C4, K_C5, K_C6,
    K_AT0, K_AT1, K_AT2, K_AT3, K_AT4, K_AT5, K_AT6, K_AT7, K_AT8, K_AT9, K_AT10,
    K_ATAN_HALF_HI, K_ATAN_HALF_LO, K_PIO4_HI, K_PIO4_LO, K_PIO2_HI, K_PIO2_LO, K_PI_HI, K_PI_LO,
    K_TWO_PI, K_SCL, K_COUNT
};

#define DSP_MATH_TABLE_INIT {                                                              \
    6.36619772367581382433e-01,  /* 2/pi                                  */               \
    1.57079632679489655800e+00,  /* P1 = fl(pi/2)                         */               \
    6.12323399573676603587e-17,  /* P2 = fl(pi/2 - P1)                    */               \
    -1.49738490485916983294e-33, /* P3 = fl(pi/2 - P1 - P2)               */               \
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,  \
    2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,   \
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,   \
    -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,  \
    3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,   \
    -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,  \
    6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,   \
    -3.65315727442169155270e-02, 1.62858201153657823623e-02,                               \
    4.63647609000806093515e-01, 2.26987774529616870924e-17,  /* atan(1/2) hi, lo */        \
    7.85398163397448278999e-01, 3.06161699786838301793e-17,  /* pi/4 hi, lo      */        \
    1.57079632679489655800e+00, 6.12323399573676603587e-17,  /* pi/2 hi, lo      */        \
    3.14159265358979311600e+00, 1.22464679914735320717e-16,  /* pi hi, lo        */        \
    6.283185307179586,           /* 2 * np.pi                             */               \
    0.7071067811865475           /* 1.0 / np.sqrt(2)                      */               \
}

struct MathK { double v[K_COUNT]; };

DSP_HD MathK math_consts_literal()
{
    const double t[K_COUNT] = DSP_MATH_TABLE_INIT;
    MathK k;
    for (int i = 0; i < K_COUNT; i++) k.v[i] = t[i];
    return k;
}

// In-kernel stamps of the diagnostic build (-DDSP_TRACE, never the product): DSP_TR(tr, i)
// adds the s_memtime ticks since the previous mark to segment i.  Empty otherwise.
#if defined(DSP_TRACE)
struct TrAcc { unsigned long long acc[16]; unsigned long long last; };
DSP_HD void tr_mark(TrAcc *tr, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (!tr) return;
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    __builtin_amdgcn_sched_barrier(0);
    if (i >= 0) tr->acc[i] += t - tr->last;
    tr->last = t;
#endif
}
#define DSP_TR(tr, i) dspm::tr_mark(tr, i)
#else
struct TrAcc { };
#define DSP_TR(tr, i) ((void)0)
#endif

// fused multiply-add, one rounding.  (A hand-placed v_fma_f64 through inline asm was
// measured 21% SLOWER than letting hipcc pick v_fmac_f64 + moves: the compiler interleaves
// the independent sin / cos / atan chains, which matters more than the instruction count.)
DSP_HD double fma3(double a, double b, double c) { return __builtin_fma(a, b, c); }

// wave-uniform vote (device) / plain predicate (host build of the same source)
#if defined(__HIP_DEVICE_COMPILE__)
#define DSP_WAVE_ALL(p) (__all((int)(p)) != 0)
#else
#define DSP_WAVE_ALL(p) (p)
#endif

// x - floor(x) for x >= 0, exact
DSP_HD double fract_pos(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fract(x);
#else
    return x - __builtin_floor(x);
#endif
}

// num / den to <= 1 ulp without the scaling / fix-up steps of the IEEE sequence (operands
// here are sample magnitudes, far from the exponent limits): rcp + 2 Newton steps + one
// residual correction.
DSP_HD double div_fast(double num, double den)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = num * r;
    double rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r, q);
#else
    return num / den;
#endif
}

// ---------------------------------------------------------------------------
// sin and cos of a float64 angle in radians, any magnitude up to ~1e15.
// Argument reduction by pi/2 in three FMA steps (pi/2 split into three
// doubles, ~160 bits), then the fdlibm minimax kernels on [-pi/4, pi/4].
// Absolute error <= ~2e-16; the reference's libm is < 1 ulp, so derotated
// samples agree to ~1e-16 relative.
DSP_HD void sincos_f64(const MathK &K, double x, double &s, double &c)
{
    double k = d_rint(x * K.v[K_TWO_OVER_PI]);
    double nk = -k;
    double r = fma3(nk, K.v[K_P1], x);
    r = fma3(nk, K.v[K_P2], r);
    r = fma3(nk, K.v[K_P3], r);
    double z = r * r;
    double ps = fma3(z, K.v[K_S6], K.v[K_S5]);
    ps = fma3(z, ps, K.v[K_S4]);
    ps = fma3(z, ps, K.v[K_S3]);
    ps = fma3(z, ps, K.v[K_S2]);
    ps = fma3(z, ps, K.v[K_S1]);
    double sn = fma3(r * z, ps, r);
    double pc = fma3(z, K.v[K_C6], K.v[K_C5]);
    pc = fma3(z, pc, K.v[K_C4]);
    pc = fma3(z, pc, K.v[K_C3]);
    pc = fma3(z, pc, K.v[K_C2]);
    pc = fma3(z, pc, K.v[K_C1]);
    double cs = fma3(z * z, pc, fma3(-0.5, z, 1.0));
    // quadrant = k mod 4: adding 1.5 * 2^52 leaves k (two's complement) in the low mantissa
    // bits, valid for |k| < 2^51
    double kk = k + 6755399441055744.0;
    unsigned long long kb;
    __builtin_memcpy(&kb, &kk, 8);
    unsigned q = (unsigned)kb;
    double s0 = (q & 1) ? cs : sn;
    double c0 = (q & 1) ? sn : cs;
    // sign flips as bit operations on the high word
    unsigned long long sb, cb;
    __builtin_memcpy(&sb, &s0, 8);
    __builtin_memcpy(&cb, &c0, 8);
    sb ^= (unsigned long long)((q << 30) & 0x80000000u) << 32;
    cb ^= (unsigned long long)(((q + 1) << 30) & 0x80000000u) << 32;
    __builtin_memcpy(&s, &sb, 8);
    __builtin_memcpy(&c, &cb, 8);
}

// ---------------------------------------------------------------------------
// atan2(y, x), all quadrants, fdlibm atan kernel with one division.  For t = min/max in
// [0,1]:  t >= 7/16 -> atan(1) + atan((t-1)/(t+1)) (|arg| <= 0.392, inside the kernel's
// |x| < 7/16 range), else atan(t); the interval choice is folded into the quotient by selects.
// `easy`: the caller knows x > 0 and |y| <= x for every lane of the wave that matters
// (the FLL's phase error lies in [-pi/4, pi/4]); the octant logic is then skipped.  Both
// paths compute identical values for such inputs.
DSP_HD double atan2_f64(const MathK &K, double y, double x, bool easy)
{
    double ax = d_abs(x), ay = d_abs(y);
    bool swap = false;
    double mx = ax, mn = ay;
    if (!easy) {
        swap = ay > ax;
        mx = swap ? ay : ax;
        mn = swap ? ax : ay;
    }
    bool zero = mx == 0.0;                       // atan2(+-0, +-0): quotient 0, signs below
    bool c2 = (16.0 * mn >= 7.0 * mx) && !zero;
    double num = c2 ? mn - mx : mn;
    double den = c2 ? mn + mx : mx;
    double hi = c2 ? K.v[K_PIO4_HI] : 0.0;
    double lo = c2 ? K.v[K_PIO4_LO] : 0.0;
    den = zero ? 1.0 : den;
    double t = div_fast(num, den);
    double z = t * t;
    double w = z * z;
    double s1 = fma3(w, K.v[K_AT10], K.v[K_AT8]);
    s1 = fma3(w, s1, K.v[K_AT6]);
    s1 = fma3(w, s1, K.v[K_AT4]);
    s1 = fma3(w, s1, K.v[K_AT2]);
    s1 = fma3(w, s1, K.v[K_AT0]);
    s1 = z * s1;
    double s2 = fma3(w, K.v[K_AT9], K.v[K_AT7]);
    s2 = fma3(w, s2, K.v[K_AT5]);
    s2 = fma3(w, s2, K.v[K_AT3]);
    s2 = fma3(w, s2, K.v[K_AT1]);
    s2 = w * s2;
    // atan(t) = hi - ((t*(s1+s2) - lo) - t)
    double a = hi - ((t * (s1 + s2) - lo) - t);
    if (!easy) {
        a = swap ? K.v[K_PIO2_HI] - (a - K.v[K_PIO2_LO]) : a;
        a = __builtin_signbit(x) ? K.v[K_PI_HI] - (a - K.v[K_PI_LO]) : a;
    }
    return d_copysign(a, y);
}

// ---------------------------------------------------------------------------
// One iteration of fll_func's loop body (fll_func.py:15-26) for sample x at
// absolute index i.  state: curr_freq, integrator.  kp, ki: fll_func.py:5-6.
// `live`: this lane's result will be used (dead lanes do not veto the wave's fast path).
struct FllOut { double sr, si, ef, freq; int quad; };

DSP_HD FllOut fll_step(const MathK &K, double xr, double xi, double idx, double &curr_freq,
                       double &integrator, double kp, double ki, bool live = true, TrAcc *tr = nullptr)
{
    double phase = K.v[K_TWO_PI] * curr_freq * idx;     // :16
    double s, c;
    sincos_f64(K, phase, s, c);                         // exp(-1j*phase) = (c, -s)
    DSP_TR(tr, 2);
    double sr = xr * c + xi * s;                        // xr*wr - xi*wi, wi = -s
    double si = xi * c - xr * s;                        // xr*wi + xi*wr
    // :19 decided = (sign(re) + 1j*sign(im)) / sqrt(2); np.sign(0) = 0, NaN stays NaN through
    // the products below
    double cr = d_copysign(K.v[K_SCL], sr), ci = d_copysign(K.v[K_SCL], si);
    cr = (sr == 0.0) ? 0.0 : cr;
    ci = (si == 0.0) ? 0.0 : ci;
    double zr = sr * cr + si * ci;                      // sample * conj(decided)
    double zi = si * cr - sr * ci;
    bool easy = DSP_WAVE_ALL(!live || (zr > 0.0 && d_abs(zi) <= zr));
    DSP_TR(tr, 3);
    double error = atan2_f64(K, zi, zr, easy);          // :21
    DSP_TR(tr, 4);
    integrator += error * ki;                           // :22
    double ef_n = error * kp + integrator;              // :23
    curr_freq += ef_n;                                  // :25
    FllOut o;
    o.sr = sr; o.si = si; o.ef = ef_n; o.freq = curr_freq;
    o.quad = (int)(sr < 0.0) + 2 * (int)(si < 0.0);
    return o;
}

// ---------------------------------------------------------------------------
// The same iteration restated for instruction-level parallelism (round 3): same values to ~1e-16,
// a much shorter dependency chain and no wave vote, for the blocks in which every lane of the wave
// is live.  A lone wave issues a DEPENDENT instruction every ~8 cycles and independent ones every
// ~5 (tools/ubench/issue_latency.hip), and fll_step above compiles to one serial chain.
//   * the quadrant index k comes from 4*f*i (= phase * 2/pi up to rounding), formed beside the
//     phase product instead of after it; pi/2 is taken off in two FMA steps (the third term, k*1.5e-33,
//     is below 1e-17 for every k that fits the int32 quadrant conversion);
//   * the quarter turns are applied to the SAMPLE (swap / negate, off the chain) instead of to
//     (sin, cos): same products, same sums, same bits as rotating by the adjusted pair;
//   * sin and cos polynomials in Estrin form, interleaved;
//   * decided = (sign(re) + 1j*sign(im))/sqrt(2) makes sample*conj(decided) = (P + Q) + 1j*sigma*(Q - P)
//     with P = |re|/sqrt(2), Q = |im|/sqrt(2), sigma = sign(re)*sign(im) -- bit for bit the reference's
//     four products and two sums (sign changes are exact) as long as neither component is exactly 0
//     (np.sign(0) = 0 breaks the identity: `zmin` reports min(P, Q) so that the caller can send such
//     a block to fll_step); the error is then sign * atan(|Q - P| / (P + Q)), first octant by construction;
//   * second interval of the atan (t >= 7/16): (t - 1)/(t + 1) = -min(P, Q)/max(P, Q), no cancellation;
//   * reciprocal refined as r0*(1 + e)*(1 + e^2) (error e^4 like two Newton steps, one level less).
// kmax reports max |4*f*i|: beyond 2^30 the int32 quadrant is wrong (caller: fll_step).
DSP_HD void sincos_quarter(const MathK &K, double r, double &sn, double &cs)
{
    const double z = r * r;
    const double z2 = z * z;
    const double rz = r * z;
    const double hz = fma3(-0.5, z, 1.0);
    const double s12 = fma3(z, K.v[K_S2], K.v[K_S1]), c12 = fma3(z, K.v[K_C2], K.v[K_C1]);
    const double s34 = fma3(z, K.v[K_S4], K.v[K_S3]), c34 = fma3(z, K.v[K_C4], K.v[K_C3]);
    const double s56 = fma3(z, K.v[K_S6], K.v[K_S5]), c56 = fma3(z, K.v[K_C6], K.v[K_C5]);
    const double sq = fma3(z2, s56, s34), cq = fma3(z2, c56, c34);
    const double ps = fma3(z2, sq, s12), pc = fma3(z2, cq, c12);
    sn = fma3(rz, ps, r);
    cs = fma3(z2, pc, hz);
}

DSP_HD double div_short(double num, double den)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r0, 1.0);
    const double e2 = e * e;
    const double r1 = __builtin_fma(r0, e, r0);
    const double r2 = __builtin_fma(r1, e2, r1);
    const double q = num * r2;
    const double rem = __builtin_fma(-den, q, num);
    return __builtin_fma(rem, r2, q);
#else
    return num / den;
#endif
}

// bit-level helpers of the select-free formulation below.  (Consecutive v_cndmask_b32 in their short
// encoding -- the compiler's choice for every ?: on a double -- issue at 19 cycles each on gfx950,
// tools/ubench/issue_latency.hip k_cmp_4cnd_vcc; v_bfi_b32 / v_and_b32 with a computed mask at 4-6.)
DSP_HD unsigned hi_word(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return (unsigned)(u >> 32); }
DSP_HD unsigned lo_word(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return (unsigned)u; }
DSP_HD double from_words(unsigned hi, unsigned lo)
{
    unsigned long long u = ((unsigned long long)hi << 32) | lo;
    double x; __builtin_memcpy(&x, &u, 8);
    return x;
}
// mask ? a : b per bit (mask all ones or all zeros)
// v_bfi_b32 D = (S0 & S1) | (~S0 & S2): ONE instruction; left to itself hipcc emits v_and_b32 + v_and_or_b32 (with
// ~mask kept in a register) or v_cndmask for this expression -- two issue slots per select on a wave that fills its
// SIMD's VALU by itself (119-132 instructions per symbol).  Whether it pays is measured per kernel family
// (profiles/r03_bfi_ab.txt) and chosen by the caller: SEL = 1 the asm form, 0 the plain expression.
#if defined(DSP_NO_BFI)
#define DSP_BFI_DEFAULT 0
#else
#define DSP_BFI_DEFAULT 1
#endif
#if defined(DSP_NO_BFI_T)
#define DSP_BFI_T_DEFAULT 0
#else
#define DSP_BFI_T_DEFAULT 1
#endif
template <int SEL>
DSP_HD unsigned bsel_s(unsigned mask, unsigned a, unsigned b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (SEL) {
        unsigned d;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
        return d;
    }
#endif
    return (a & mask) | (b & ~mask);
}
template <int SEL>
DSP_HD double bsel64_s(unsigned mask, double a, double b)
{
    return from_words(bsel_s<SEL>(mask, hi_word(a), hi_word(b)), bsel_s<SEL>(mask, lo_word(a), lo_word(b)));
}
DSP_HD unsigned bsel(unsigned mask, unsigned a, unsigned b) { return bsel_s<DSP_BFI_DEFAULT>(mask, a, b); }
DSP_HD double bsel64(unsigned mask, double a, double b) { return bsel64_s<DSP_BFI_DEFAULT>(mask, a, b); }
// the timing loop's pick3 (sync_kernels.hip), switched on its own
DSP_HD unsigned bsel_t(unsigned mask, unsigned a, unsigned b) { return bsel_s<DSP_BFI_T_DEFAULT>(mask, a, b); }
DSP_HD double bsel64_t(unsigned mask, double a, double b) { return bsel64_s<DSP_BFI_T_DEFAULT>(mask, a, b); }

// evidence of a block of fll_step_fast calls, kept as integers (high words): no compare, no select
struct FllEvidence {
    unsigned zbad;      // max over the block of (hi(P) - 1) and (hi(Q) - 1) as unsigned: >= 0x7fefffff when a
                        // component was zero (or denormal), infinite or NaN
    unsigned kmax;      // max over the block of hi(|4*f*i|): >= hi(2^30) when the int32 quadrant is unsafe
};
DSP_HD void fll_evidence_init(FllEvidence &e) { e.zbad = 0u; e.kmax = 0u; }
DSP_HD bool fll_evidence_bad(const FllEvidence &e) { return e.zbad >= 0x7fefffffu || e.kmax >= 0x41d00000u; }

template <int SEL = DSP_BFI_DEFAULT>
DSP_HD FllOut fll_step_fast(const MathK &K, double xr, double xi, double idx, double idx4, double &curr_freq,
                            double &integrator, double kp, double ki, FllEvidence &ev)
{
    const double t1 = K.v[K_TWO_PI] * curr_freq;
    const double kq = curr_freq * idx4;
    const double phase = t1 * idx;                      // :16, the reference's two roundings
    const double k = d_rint(kq);
    double r = fma3(-k, K.v[K_P1], phase);
    r = fma3(-k, K.v[K_P2], r);
    const unsigned kh = hi_word(kq) & 0x7fffffffu;
    ev.kmax = kh > ev.kmax ? kh : ev.kmax;
    // x * (-1j)^k, off the chain
#if defined(__HIP_DEVICE_COMPILE__)
    const int q = __double2int_rz(k);
#else
    const int q = (int)(long long)k;
#endif
    const unsigned odd = 0u - ((unsigned)q & 1u);       // all ones when k is odd
    const unsigned neg_r = ((unsigned)q << 30) & 0x80000000u;
    const unsigned neg_i = (((unsigned)q + 1u) << 30) & 0x80000000u;
    const double ar = from_words(bsel_s<SEL>(odd, hi_word(xi), hi_word(xr)) ^ neg_r, bsel_s<SEL>(odd, lo_word(xi), lo_word(xr)));
    const double ai = from_words(bsel_s<SEL>(odd, hi_word(xr), hi_word(xi)) ^ neg_i, bsel_s<SEL>(odd, lo_word(xr), lo_word(xi)));
    double sn, cs;
    sincos_quarter(K, r, sn, cs);
    const double sr = ar * cs + ai * sn;                // :17
    const double si = ai * cs - ar * sn;
    const double P = d_abs(sr) * K.v[K_SCL], Q = d_abs(si) * K.v[K_SCL];
    const double S = P + Q, D = Q - P;                  // :21: re and sigma * im of sample * conj(decided)
    const double mn = __builtin_fmin(P, Q), mx = __builtin_fmax(P, Q);
    const unsigned zp = hi_word(P) - 1u, zq = hi_word(Q) - 1u;
    const unsigned zpq = zp > zq ? zp : zq;
    ev.zbad = zpq > ev.zbad ? zpq : ev.zbad;
    const double aD = d_abs(D);
    // second interval (t >= 7/16) <=> 7*S - 16*|D| < 0; at equality either interval serves
    const double c2d = fma3(-16.0, aD, 7.0 * S);
    const unsigned c2 = (unsigned)((int)hi_word(c2d) >> 31);
    const double num = bsel64_s<SEL>(c2, -mn, aD);
    const double den = bsel64_s<SEL>(c2, mx, S);
    const double hi = from_words(hi_word(K.v[K_PIO4_HI]) & c2, lo_word(K.v[K_PIO4_HI]) & c2);
    const double lo = from_words(hi_word(K.v[K_PIO4_LO]) & c2, lo_word(K.v[K_PIO4_LO]) & c2);
#ifdef DSP_EXP_FLL_CHEAP          // experiment: which wave is the bottleneck (results are wrong)
    const double t = num * den;
#else
    const double t = div_short(num, den);
#endif
    const double z = t * t;
    const double w = z * z;
    const double w2 = w * w;
    const double e0 = fma3(w, K.v[K_AT2], K.v[K_AT0]), f0 = fma3(w, K.v[K_AT3], K.v[K_AT1]);
    const double e1 = fma3(w, K.v[K_AT6], K.v[K_AT4]), f1 = fma3(w, K.v[K_AT7], K.v[K_AT5]);
    const double e2 = fma3(w, K.v[K_AT10], K.v[K_AT8]);
    const double s1 = z * fma3(w2, fma3(w2, e2, e1), e0);
    const double s2 = w * fma3(w2, fma3(w2, K.v[K_AT9], f1), f0);
    const double a = hi - ((t * (s1 + s2) - lo) - t);
    // sign of sigma * (Q - P)
    const double error = from_words(hi_word(a) ^ ((hi_word(sr) ^ hi_word(si) ^ hi_word(D)) & 0x80000000u), lo_word(a));
    integrator += error * ki;                           // :22
    const double ef_n = error * kp + integrator;        // :23
    curr_freq += ef_n;                                  // :25
    FllOut o;
    o.sr = sr; o.si = si; o.ef = ef_n; o.freq = curr_freq;
    // (Re < 0) + 2 (Im < 0) from the sign bits: exact for every sample the evidence lets through
    // (-0 and NaN, where the sign bit and the comparison disagree, send the block to fll_step)
    o.quad = (int)((hi_word(sr) >> 31) | ((hi_word(si) >> 30) & 2u));
    return o;
}

// ---------------------------------------------------------------------------
// Linear interpolation sig[i]*(1-frac) + sig[i+1]*frac with the reference's
// rounding (gardner2.py:47-48, hw1/gardner.py:19).
DSP_HD cd lerp(cd a, cd b, double frac)
{
    double w0 = 1.0 - frac;
    cd z;
    z.re = a.re * w0 + b.re * frac;
    z.im = a.im * w0 + b.im * frac;
    return z;
}

// Gardner TED: real(x_mid * conj(x_late - x_early)) (gardner2.py:87).
DSP_HD double ted_error(cd xe, cd xm, cd xl)
{
    double dr = xl.re - xe.re, di = xl.im - xe.im;
    return xm.re * dr + xm.im * di;
}

}  // namespace dspm

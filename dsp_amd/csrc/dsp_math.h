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
// sin and cos of a float64 angle in radians, any magnitude up to ~1e15.
// Argument reduction by pi/2 in three FMA steps (pi/2 split into three
// doubles, ~160 bits), then the fdlibm minimax kernels on [-pi/4, pi/4].
// Absolute error <= ~2e-16; the reference's libm is < 1 ulp, so derotated
// samples agree to ~1e-16 relative.
DSP_HD void sincos_f64(double x, double &s, double &c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632679489655800e+00;   // fl(pi/2)
    const double P2 = 6.12323399573676603587e-17;   // fl(pi/2 - P1)
    const double P3 = -1.49738490485916983294e-33;  // fl(pi/2 - P1 - P2)
    double k = d_rint(x * TWO_OVER_PI);
    double r = d_fma(-k, P1, x);
    r = d_fma(-k, P2, r);
    r = d_fma(-k, P3, r);
    double z = r * r;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = d_fma(z, S6, S5);
    ps = d_fma(z, ps, S4);
    ps = d_fma(z, ps, S3);
    ps = d_fma(z, ps, S2);
    ps = d_fma(z, ps, S1);
    double sn = d_fma(r * z, ps, r);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = d_fma(z, C6, C5);
    pc = d_fma(z, pc, C4);
    pc = d_fma(z, pc, C3);
    pc = d_fma(z, pc, C2);
    pc = d_fma(z, pc, C1);
    double cs = d_fma(z * z, pc, d_fma(-0.5, z, 1.0));
    // quadrant = k mod 4, valid for any |k| < 2^52
    double kq = k * 0.25;
    int q = (int)((kq - d_floor(kq)) * 4.0);
    double s0 = (q & 1) ? cs : sn;
    double c0 = (q & 1) ? sn : cs;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// ---------------------------------------------------------------------------
// atan2(y, x), all quadrants, fdlibm atan kernel with one division.
DSP_HD double atan2_f64(double y, double x)
{
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    const double ATAN_HALF_HI = 4.63647609000806093515e-01, ATAN_HALF_LO = 2.26987774529616870924e-17;
    const double PIO4_HI = 7.85398163397448278999e-01, PIO4_LO = 3.06161699786838301793e-17;
    const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    const double PI_HI = 3.14159265358979311600e+00, PI_LO = 1.22464679914735320717e-16;
    double ax = d_abs(x), ay = d_abs(y);
    double mx = ax > ay ? ax : ay;
    double mn = ax > ay ? ay : ax;
    // t = mn/mx in [0,1]; fold fdlibm's interval reduction into the quotient
    double num = mn, den = mx, hi = 0.0, lo = 0.0;
    double m16 = 16.0 * mn;
    if (mx == 0.0) {                   // atan2(+-0, +-0): quotient 0, signs below
        den = 1.0;
    } else if (m16 >= 11.0 * mx) {     // t >= 11/16: atan(1) + atan((t-1)/(t+1))
        num = mn - mx; den = mn + mx; hi = PIO4_HI; lo = PIO4_LO;
    } else if (m16 >= 7.0 * mx) {      // 7/16 <= t < 11/16: atan(1/2) + atan((2t-1)/(2+t))
        num = 2.0 * mn - mx; den = 2.0 * mx + mn; hi = ATAN_HALF_HI; lo = ATAN_HALF_LO;
    }
    double t = num / den;
    double z = t * t;
    double w = z * z;
    double s1 = d_fma(w, aT10, aT8);
    s1 = d_fma(w, s1, aT6);
    s1 = d_fma(w, s1, aT4);
    s1 = d_fma(w, s1, aT2);
    s1 = d_fma(w, s1, aT0);
    s1 = z * s1;
    double s2 = d_fma(w, aT9, aT7);
    s2 = d_fma(w, s2, aT5);
    s2 = d_fma(w, s2, aT3);
    s2 = d_fma(w, s2, aT1);
    s2 = w * s2;
    // atan(t) = hi - ((t*(s1+s2) - lo) - t)
    double a = hi - ((t * (s1 + s2) - lo) - t);
    if (ay > ax) a = PIO2_HI - (a - PIO2_LO);
    if (__builtin_signbit(x)) a = PI_HI - (a - PI_LO);
    return d_copysign(a, y);
}

// ---------------------------------------------------------------------------
// Loop constants of fll_func (fll_func.py:5-6), computed on the host.
struct FllConst { double kp, ki; };

// One iteration of fll_func's loop body (fll_func.py:15-26) for sample x at
// absolute index i.  state: curr_freq, integrator.
struct FllOut { double sr, si, ef, freq; int quad; };

DSP_HD FllOut fll_step(double xr, double xi, double idx, double &curr_freq, double &integrator,
                       double kp, double ki)
{
    const double TWO_PI = 6.283185307179586;            // 2 * np.pi
    const double SCL = 0.7071067811865475;              // 1.0 / np.sqrt(2)
    double phase = TWO_PI * curr_freq * idx;            // :16
    double s, c;
    sincos_f64(phase, s, c);                            // exp(-1j*phase) = (c, -s)
    double sr = xr * c + xi * s;                        // xr*wr - xi*wi, wi = -s
    double si = xi * c - xr * s;                        // xr*wi + xi*wr
    double gr = (double)((sr > 0.0) - (sr < 0.0));      // np.sign (sign(0) = 0)
    double gi = (double)((si > 0.0) - (si < 0.0));
    if (sr != sr) gr = sr;
    if (si != si) gi = si;
    double cr = gr * SCL, ci = gi * SCL;
    double zr = sr * cr + si * ci;                      // sample * conj(decided)
    double zi = si * cr - sr * ci;
    double error = atan2_f64(zi, zr);                   // :21
    integrator += error * ki;                           // :22
    double ef_n = error * kp + integrator;              // :23
    curr_freq += ef_n;                                  // :25
    FllOut o;
    o.sr = sr; o.si = si; o.ef = ef_n; o.freq = curr_freq;
    o.quad = (int)(sr < 0.0) + 2 * (int)(si < 0.0);
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

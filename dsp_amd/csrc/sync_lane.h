// sync_lane.h -- one lane's view of the sequential recurrences.
//
// A "lane" owns one burst (or one segment of a long capture).  The functions
// here advance that lane by ONE loop iteration of the reference's loops; the
// sample source is abstracted as fetch(row) -> complex double, so the same
// source runs inside the HIP kernel (LDS ring / global memory) and in the host
// unit tests (plain arrays).
//
//   gardner2_iter   gardner2.py:63-98   (gardner_timing_recovery)
//   ted_iter        hw1/gardner.py:45-72, read_file.py:75-102 (gardner_ted)
//   lagrange_at     hw1/test_lagrange_resampling.py:54-70 (one position)
#pragma once
#include "dsp_math.h"

namespace dspm {

// Interpolator selection: half_len == 0 -> the reference's 2-point linear
// interpolation; half_len = h > 0 -> (2h+1)-point barycentric Lagrange
// (optional mode; the reference's Gardner loops only ever use linear).
#define DSP_MAX_HALF_LEN 8

struct LagrangeW { double w[2 * DSP_MAX_HALF_LEN + 1]; int half_len; };

// weights w_i = prod_{j != i} 1/(o_i - o_j)   (test_lagrange_resampling.py:45-50)
DSP_HD void lagrange_weights(int half_len, LagrangeW &L)
{
    L.half_len = half_len;
    int T = 2 * half_len + 1;
    for (int i = 0; i < T; i++) {
        double w = 1.0;
        for (int j = 0; j < T; j++) {
            if (i == j) continue;
            w *= 1.0 / ((double)(i - half_len) - (double)(j - half_len));
        }
        L.w[i] = w;
    }
}

// numpy's pairwise sum for 8 <= n <= 128 doubles / n < 8 (see oracle).  T <= 17.
DSP_HD double np_sum_small(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return 0.0 + r;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return 0.0 + res;
}

// complex variant: a holds n doubles (re, im interleaved), n = 2*T
DSP_HD void np_sum_small_c(const double *a, int n, double &rr, double &ri)
{
    if (n < 8) {
        rr = -0.0; ri = -0.0;
        for (int i = 0; i < n; i += 2) { rr += a[i]; ri += a[i + 1]; }
    } else {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        rr = ((r[0] + r[2]) + (r[4] + r[6]));
        ri = ((r[1] + r[3]) + (r[5] + r[7]));
        for (; i < n; i += 2) { rr += a[i]; ri += a[i + 1]; }
    }
    rr = 0.0 + rr; ri = 0.0 + ri;
}

// One output of lagrange_resample at position pos (indices clipped to the
// signal, exact-node shortcut, numer/denom by numpy's summation order).
template <class Fetch>
DSP_HD cd lagrange_at(double pos, int len, const LagrangeW &L, Fetch &&fetch)
{
    const int h = L.half_len, T = 2 * h + 1;
    double fl = d_floor(pos);
    // keep the integer conversion defined for wild positions
    if (!(fl > -1.0e9)) fl = -1.0e9;
    if (!(fl < 1.0e9)) fl = 1.0e9;
    int base = (int)fl;
    double mu = pos - (double)base;
    double tmp[2 * (2 * DSP_MAX_HALF_LEN + 1)];
    double den[2 * DSP_MAX_HALF_LEN + 1];
    cd exact; bool has_exact = false;
    for (int i = 0; i < T; i++) {
        int ix = base + (i - h);
        ix = ix < 0 ? 0 : ix;
        ix = ix > len - 1 ? len - 1 : ix;
        cd s = fetch(ix);
        double d = mu - (double)(i - h);
        if (!has_exact && d_abs(d) < 1e-12) { has_exact = true; exact = s; }
        double scl = 1.0 / d;
        tmp[2 * i] = (L.w[i] * s.re) * scl;       // (w*s) / d  as complex / real
        tmp[2 * i + 1] = (L.w[i] * s.im) * scl;
        den[i] = L.w[i] / d;
    }
    if (has_exact) return exact;
    double nr, ni;
    np_sum_small_c(tmp, 2 * T, nr, ni);
    double dsum = np_sum_small(den, T);
    double scl = 1.0 / dsum;
    cd z; z.re = nr * scl; z.im = ni * scl;
    return z;
}

// ---------------------------------------------------------------------------
struct GState {
    double mu;      // fractional delay, samples
    int k;          // gardner2: current symbol index in samples; ted: symbol counter
    bool active;    // loop still running for this lane
};

struct GOut {
    bool produced;  // this iteration appended to the outputs
    cd rec;         // recovered sample
    double err;     // timing error
    double mu;      // mu_history entry
};

// interpolate() of gardner2.py:38-50 (int() truncation, bounds -> 0.0 / last).
template <class Fetch>
DSP_HD cd g2_interp(double idx, int len, const LagrangeW *L, Fetch &&fetch)
{
    cd z; z.re = 0.0; z.im = 0.0;
    if (!(idx > -1.0e9 && idx < 1.0e9)) return z;      // also catches NaN
    if (L) {
        int ii = (int)idx;
        if (ii < 0 || ii >= len) return z;
        return lagrange_at(idx, len, *L, fetch);
    }
    int ii = (int)idx;                                 // toward zero, like int()
    double frac = idx - (double)ii;
    if (ii < 0 || ii >= len) return z;
    if (ii + 1 < len) return lerp(fetch(ii), fetch(ii + 1), frac);
    return fetch(ii);
}

// One pass of the while-loop body of gardner_timing_recovery (gardner2.py:63-98).
template <class Fetch>
DSP_HD GOut gardner2_iter(GState &st, int len, int sps, double alpha, const LagrangeW *L,
                          Fetch &&fetch)
{
    GOut o; o.produced = false; o.rec.re = 0.0; o.rec.im = 0.0; o.err = 0.0; o.mu = 0.0;
    if (!st.active) return o;
    if (!(st.k <= len - sps)) { st.active = false; return o; }          // :63
    const double spsd = (double)sps;
    double mu = st.mu;
    int k = st.k;
    // :65-70.  A non-finite mu can never satisfy the reference's loop exit; stop the lane.
    if (!(d_abs(mu) < 1.0e9)) { st.active = false; return o; }
    while (mu >= spsd) { mu -= spsd; k += sps; }
    while (mu < -spsd) { mu += spsd; k -= sps; }
    double idx_early = (double)(k - sps) + mu;                           // :73
    double idx_mid = (double)(k - sps / 2) + mu;                         // :74
    double idx_late = (double)k + mu;                                    // :75
    if (idx_early < 0.0 || idx_late + 1.0 >= (double)len) {              // :78-80
        st.active = false; st.mu = mu; st.k = k; return o;
    }
    cd xe = g2_interp(idx_early, len, L, fetch);
    cd xm = g2_interp(idx_mid, len, L, fetch);
    cd xl = g2_interp(idx_late, len, L, fetch);
    double e = ted_error(xe, xm, xl);                                    // :87
    mu = mu - alpha * e;                                                 // :91
    o.produced = true; o.err = e; o.mu = mu;
    o.rec = g2_interp((double)k + mu, len, L, fetch);                    // :95
    k += sps;                                                            // :98
    st.mu = mu; st.k = k;
    return o;
}

// interpolate_sample of hw1/gardner.py:8-19 (np.floor; last sample unlerped).
template <class Fetch>
DSP_HD cd ted_interp(double idx, int len, const LagrangeW *L, Fetch &&fetch)
{
    if (L) return lagrange_at(idx, len, *L, fetch);
    double fl = d_floor(idx);
    int lo = (int)fl;
    int hi = lo + 1;
    if (lo < 0) lo += len;                 // python's negative index (unreachable: mu >= -sps)
    if (hi >= len) return fetch(lo);
    double frac = idx - fl;
    return lerp(fetch(lo), fetch(hi), frac);
}

// One pass of the for-loop body of gardner_ted (hw1/gardner.py:45-72).
// st.k counts symbols; n_symbols = (len - 2*sps)//sps is checked by the caller's
// bound passed as n_symbols.
template <class Fetch>
DSP_HD GOut ted_iter(GState &st, int len, int sps, int n_symbols, double delta_mu,
                     const LagrangeW *L, Fetch &&fetch)
{
    GOut o; o.produced = false; o.rec.re = 0.0; o.rec.im = 0.0; o.err = 0.0; o.mu = 0.0;
    if (!st.active) return o;
    if (!(st.k < n_symbols)) { st.active = false; return o; }            // :45
    const double spsd = (double)sps;
    double mu = st.mu;
    int k = st.k;
    double idx_k = (double)(k * sps) + mu;                               // :47
    double idx_next = (double)((k + 1) * sps) + mu;                      // :48
    double idx_mid = (double)(k * sps) + mu + spsd / 2.0;                // :49
    if (idx_next + 1.0 >= (double)len || idx_mid + 1.0 >= (double)len) { // :52
        st.active = false; return o;
    }
    cd xk = ted_interp(idx_k, len, L, fetch);
    cd xk1 = ted_interp(idx_next, len, L, fetch);
    cd xm = ted_interp(idx_mid, len, L, fetch);
    double e = ted_error(xk, xm, xk1);                                   // :61
    o.produced = true; o.err = e; o.mu = mu; o.rec = xk;                 // :62-63,:72
    mu += delta_mu * e;                                                  // :66
    mu = mu < -spsd ? -spsd : mu;                                        // :69
    mu = mu > spsd ? spsd : mu;
    st.mu = mu; st.k = k + 1;
    return o;
}

}  // namespace dspm

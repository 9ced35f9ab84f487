/*
 * dsp_oracle.c -- CPU restatement (float64, same operation order) of the
 * receiver-synchronisation hot path of boryaska/dsp.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The
 * product (dsp_amd/) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked in
 * tests/test_oracle_golden.py against golden vectors produced by importing the
 * reference itself in the build container (tests/golden/make_golden.py); the
 * sequential loops (gardner_timing_recovery, gardner_ted, fll_func) reproduce
 * the reference bit for bit on those vectors, the FIR to ~1e-15 relative (the
 * reference's np.convolve sums through BLAS in an unspecified order).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile).
 * -ffp-contract=off is load-bearing: the reference rounds every product and
 * sum separately (CPython / numpy scalar arithmetic has no FMA).
 *
 * Each function cites the reference file:line it follows.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double re, im; } c128;
typedef struct { float re, im; } c64;

/* ------------------------------------------------------------------------- */
/* numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src), the
 * summation np.sum / np.std use; reproduced so that np.std and
 * rrc_filter's normalisation match bit for bit.  A 1-D contiguous reduction is
 * 0.0 + pairwise(all n) (checked empirically against numpy 2.2.6).           */
static double np_pairwise(const double *a, int64_t n, int64_t stride)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i * stride];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise(a, n2, stride) + np_pairwise(a + n2 * stride, n - n2, stride);
    }
}

/* complex pairwise sum: n counts DOUBLES (2 per element), as numpy's
 * pairwise_sum_CDOUBLE does.                                                 */
static void np_pairwise_c(double *rr, double *ri, const double *a, int64_t n)
{
    if (n < 8) {
        *rr = -0.0; *ri = -0.0;   /* numpy >= 1.25 starts these at -0.0 */
        for (int64_t i = 0; i < n; i += 2) { *rr += a[i]; *ri += a[i + 1]; }
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        *rr = ((r[0] + r[2]) + (r[4] + r[6]));
        *ri = ((r[1] + r[3]) + (r[5] + r[7]));
        for (; i < n; i += 2) { *rr += a[i]; *ri += a[i + 1]; }
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        double rr1, ri1, rr2, ri2;
        np_pairwise_c(&rr1, &ri1, a, n2);
        np_pairwise_c(&rr2, &ri2, a + n2, n - n2);
        *rr = rr1 + rr2; *ri = ri1 + ri2;
    }
}

double orc_sum(const double *a, int64_t n) { return 0.0 + np_pairwise(a, n, 1); }

/* python floor division for ints */
static int64_t floordiv(int64_t a, int64_t b)
{
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}

/* ------------------------------------------------------------------------- */
/* a1: rrc_filter(sps, span, alpha) -- read_file.py:5-36 (copy diff_method.py:8-39).
 * n = arange(-span*sps//2, span*sps//2+1)  :17   (unary minus binds first)
 * returns the tap count, or -1 if cap is too small.                          */
int64_t orc_rrc_taps(int sps, int span, double alpha, double *h, int64_t cap)
{
    int64_t lo = floordiv(-(int64_t)span * sps, 2);
    int64_t hi = floordiv((int64_t)span * sps, 2) + 1;
    int64_t n = hi - lo;
    if (n <= 0) return 0;
    if (n > cap) return -1;
    for (int64_t i = 0; i < n; i++) {
        int64_t ti = lo + i;
        double t = (double)ti;
        if (ti == 0) {
            h[i] = (1 + alpha * (4 / M_PI - 1));                      /* :22 */
        } else if (fabs(t) == sps / (4 * alpha)) {                    /* :23 */
            h[i] = (alpha / sqrt(2.0)) * ((1 + 2 / M_PI) * sin(M_PI / (4 * alpha)) +
                                          (1 - 2 / M_PI) * cos(M_PI / (4 * alpha)));
        } else {
            double num = sin(M_PI * t / sps * (1 - alpha)) +
                         4 * alpha * t / sps * cos(M_PI * t / sps * (1 + alpha)); /* :29-30 */
            double den = M_PI * t / sps * (1 - pow(4 * alpha * t / sps, 2.0)); /* :31 */
            h[i] = num / den;
        }
    }
    /* :35  h / sqrt(sum(h**2))  (array power 2 is x*x; sum is pairwise) */
    double *sq = (double *)malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; i++) sq[i] = h[i] * h[i];
    double nrm = sqrt(orc_sum(sq, n));
    free(sq);
    for (int64_t i = 0; i < n; i++) h[i] = h[i] / nrm;
    return n;
}

/* ------------------------------------------------------------------------- */
/* a2: np.convolve(x, h, mode='same') with real taps -- worker.py:176,
 * read_file.py:122; == full convolution cropped at (M-1)//2, which for odd M
 * is apply_filter's crop by len(h)//2 (hw1/test_rrc_cascade.py:103-108).
 * Requires N >= M (numpy swaps the operands otherwise; callers never do).     */
int orc_conv_same(const c128 *x, int64_t N, const double *h, int64_t M, c128 *y)
{
    if (M > N || M <= 0) return -1;
    int64_t start = (M - 1) / 2;
    for (int64_t n = 0; n < N; n++) {
        int64_t m = n + start;              /* index into the full convolution */
        double sr = 0.0, si = 0.0;
        /* full[m] = sum_j h[j] x[m-j]; walk x ascending like correlate does */
        int64_t j_hi = m < M - 1 ? m : M - 1;
        int64_t j_lo = m - (N - 1) > 0 ? m - (N - 1) : 0;
        for (int64_t j = j_hi; j >= j_lo; j--) {
            sr += x[m - j].re * h[j];
            si += x[m - j].im * h[j];
        }
        y[n].re = sr; y[n].im = si;
    }
    return 0;
}

/* np.std(y) of a complex vector (ddof=0) -- worker.py:177.  numpy _var():
 * mean = sum(y) * (1/N) (complex / real goes through Smith's algorithm, i.e.
 * multiplication by the reciprocal), d = y - mean, |d|^2 = re*re - im*(-im),
 * var = sum(|d|^2) / N, std = sqrt(var).                                      */
double orc_std_complex(const c128 *y, int64_t N)
{
    if (N <= 0) return NAN;
    double sr, si;
    np_pairwise_c(&sr, &si, (const double *)y, 2 * N);
    sr = 0.0 + sr; si = 0.0 + si;
    double scl = 1.0 / ((double)N + 0.0 * 0.0);
    double mr = (sr + si * 0.0) * scl, mi = (si - sr * 0.0) * scl;
    double *a = (double *)malloc(sizeof(double) * N);
    for (int64_t i = 0; i < N; i++) {
        double dr = y[i].re - mr, di = y[i].im - mi;
        a[i] = dr * dr - di * (-di);
    }
    double var = orc_sum(a, N) / (double)N;
    free(a);
    return sqrt(var);
}

/* y / std for complex128 y and a real scalar: numpy promotes the scalar to
 * complex and divides by Smith's algorithm -> multiplication by 1.0/std.      */
void orc_scale_complex(c128 *y, int64_t N, double std)
{
    double scl = 1.0 / (std + 0.0 * 0.0);
    for (int64_t i = 0; i < N; i++) {
        double r = y[i].re, im = y[i].im;
        y[i].re = (r + im * 0.0) * scl;
        y[i].im = (im - r * 0.0) * scl;
    }
}

/* ------------------------------------------------------------------------- */
/* a4: nested interpolate(sig, idx) -- gardner2.py:38-50 (int() truncation).   */
static c128 g2_interp(const c128 *sig, int64_t len, double idx)
{
    c128 z = {0.0, 0.0};
    int64_t ii = (int64_t)idx;            /* int(): toward zero */
    double frac = idx - (double)ii;
    if (ii < 0 || ii >= len) return z;    /* :44-45 */
    if (ii + 1 < len) {                   /* :47-48 */
        double w0 = 1 - frac;
        z.re = sig[ii].re * w0 + sig[ii + 1].re * frac;
        z.im = sig[ii].im * w0 + sig[ii + 1].im * frac;
        return z;
    }
    return sig[ii];                       /* :49-50 */
}

/* a3: gardner_timing_recovery(signal, sps, alpha, mu_initial) -- gardner2.py:6-100.
 * Outputs: rec[n_rec] (n_rec = n+1), err[n], mu[n]; returns n_rec, -1 for an
 * empty input (the reference raises IndexError at :59), -2 if cap is too small.*/
int64_t orc_gardner2(const c128 *x, int64_t len, int sps, double alpha, double mu0,
                     c128 *rec, double *err, double *muh, int64_t cap)
{
    if (len <= 0) return -1;
    if (cap < 1) return -2;
    double mu = mu0;
    int64_t n = 0;
    rec[0] = x[0];                                    /* :59 */
    int64_t k = sps;                                  /* :61 */
    while (k <= len - sps) {                          /* :63 */
        while (mu >= sps) { mu -= sps; k += sps; }    /* :65-67 */
        while (mu < -sps) { mu += sps; k -= sps; }    /* :68-70 */
        double idx_early = (double)(k - sps) + mu;    /* :73 */
        double idx_mid = (double)(k - sps / 2) + mu;  /* :74  sps//2 */
        double idx_late = (double)k + mu;             /* :75 */
        if (idx_early < 0 || idx_late + 1 >= (double)len ||
            (double)k + mu < 0 || (double)k + mu + 1 >= (double)len) break; /* :78-80 */
        c128 xe = g2_interp(x, len, idx_early);
        c128 xm = g2_interp(x, len, idx_mid);
        c128 xl = g2_interp(x, len, idx_late);
        double dr = xl.re - xe.re, di = xl.im - xe.im;
        /* :87  real(x_mid * conj(d)) = mr*dr - mi*(-di) */
        double e = xm.re * dr - xm.im * (-di);
        mu = mu - alpha * e;                          /* :91 */
        if (n + 1 >= cap) return -2;
        err[n] = e; muh[n] = mu;
        rec[n + 1] = g2_interp(x, len, (double)k + mu); /* :95 */
        n++;
        k += sps;                                     /* :98 */
    }
    return n + 1;
}

/* a5: interpolate_sample -- hw1/gardner.py:8-19 (np.floor).                   */
static c128 ted_interp(const c128 *x, int64_t len, double idx)
{
    int64_t lo = (int64_t)floor(idx);
    int64_t hi = lo + 1;
    if (lo < 0) lo += len;                 /* python negative index (never hit) */
    if (hi >= len) return x[lo];
    double frac = idx - (double)(int64_t)floor(idx);
    double w0 = 1 - frac;
    c128 z;
    z.re = x[lo].re * w0 + x[hi].re * frac;
    z.im = x[lo].im * w0 + x[hi].im * frac;
    return z;
}

/* a5: gardner_ted(x, sps[, delta_mu]) -- hw1/gardner.py:22-74 (delta_mu=0.002
 * hard-coded :43), read_file.py:53-104 (parameter).  Returns n.               */
int64_t orc_gardner_ted(const c128 *x, int64_t len, int sps, double delta_mu,
                        c128 *y, double *err, double *muh, int64_t cap)
{
    int64_t n_symbols = floordiv(len - 2 * (int64_t)sps, sps);   /* :37 */
    double mu = 0.0;
    int64_t n = 0;
    for (int64_t k = 0; k < n_symbols; k++) {
        double idx_k = (double)(k * sps) + mu;                   /* :47 */
        double idx_next = (double)((k + 1) * sps) + mu;          /* :48 */
        double idx_mid = (double)(k * sps) + mu + sps / 2.0;     /* :49 */
        if (idx_next + 1 >= (double)len || idx_mid + 1 >= (double)len) break; /* :52 */
        c128 xk = ted_interp(x, len, idx_k);
        c128 xk1 = ted_interp(x, len, idx_next);
        c128 xm = ted_interp(x, len, idx_mid);
        double dr = xk1.re - xk.re, di = xk1.im - xk.im;
        double e = xm.re * dr - xm.im * (-di);                   /* :61 */
        if (n >= cap) return -2;
        err[n] = e; muh[n] = mu;                                 /* :62-63 */
        mu += delta_mu * e;                                      /* :66 */
        if (mu < -sps) mu = -sps;                                /* :69 clip */
        if (mu > sps) mu = sps;
        y[n] = xk;                                               /* :72 */
        n++;
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* a6: fll_func(signal, Bn, y) -- fll_func.py:3-29 (QPSK branch of
 * worker.py:18-78 is identical).  state = {curr_freq, integrator}; i0 is the
 * absolute index of x[0] (0 for the reference; non-zero only for the
 * segment-parallel mode, which has no reference counterpart).
 * Returns the final curr_freq.                                               */
double orc_fll_state(const c128 *x, int64_t M, double Bn, double y, int64_t i0,
                     const double *state_in, double *state_out,
                     c128 *out, double *freq, double *ef)
{
    double curr_freq = state_in ? state_in[0] : 0.0;
    double integrator = state_in ? state_in[1] : 0.0;
    double kp = 4 * Bn / 1000 / (y + 1 / (4 * y));                 /* :5 */
    double ki = 4 * pow(Bn / 1000 / (y + 1 / (4 * y)), 2.0);       /* :6 */
    const double two_pi = 2 * M_PI;
    const double scl = 1.0 / (sqrt(2.0) + 0.0 * 0.0);  /* complex / np.sqrt(2) */
    for (int64_t i = 0; i < M; i++) {
        double phase = two_pi * curr_freq * (double)(i + i0);      /* :16 */
        /* :17  np.exp(-1j*phase): cexp(0 - i phase) = (cos, -sin) via libm */
        double complex w = cexp(CMPLX(0.0, -phase));
        double wr = creal(w), wi = cimag(w);
        double sr = x[i].re * wr - x[i].im * wi;
        double si = x[i].re * wi + x[i].im * wr;
        if (out) { out[i].re = sr; out[i].im = si; }               /* :18 */
        /* :19 decided = (sign(re) + 1j*sign(im)) / sqrt(2) */
        double gr = (sr > 0) - (sr < 0), gi = (si > 0) - (si < 0);
        if (isnan(sr)) gr = NAN;
        if (isnan(si)) gi = NAN;
        double cr = (gr + gi * 0.0) * scl, ci = (gi - gr * 0.0) * scl;
        /* :21 angle(sample * conj(decided)); conj -> (cr, -ci) */
        double zr = sr * cr - si * (-ci);
        double zi = sr * (-ci) + si * cr;
        double error = atan2(zi, zr);
        integrator += error * ki;                                  /* :22 */
        double ef_n = error * kp + integrator;                     /* :23 */
        if (ef) ef[i] = ef_n;
        curr_freq += ef_n;                                         /* :25 */
        if (freq) freq[i] = curr_freq;
    }
    if (state_out) { state_out[0] = curr_freq; state_out[1] = integrator; }
    return curr_freq;
}

double orc_fll(const c128 *x, int64_t M, double Bn, double y,
               c128 *out, double *freq, double *ef)
{
    return orc_fll_state(x, M, Bn, y, 0, NULL, NULL, out, freq, ef);
}

/* a7: the bit-exact parity object: quadrant index (Re<0) + 2*(Im<0) of
 * signal_list[i] (fll_func.py:19 slicer; worker.py:87-88).                    */
void orc_quadrants(const c128 *z, int64_t M, uint8_t *q)
{
    for (int64_t i = 0; i < M; i++)
        q[i] = (uint8_t)((z[i].re < 0) + 2 * (z[i].im < 0));
}

/* ------------------------------------------------------------------------- */
/* a9: lagrange_resample(signal, positions, half_len) --
 * hw1/test_lagrange_resampling.py:34-72.                                      */
int orc_lagrange_resample(const c128 *sig, int64_t len, const double *pos, int64_t P,
                          int half_len, c128 *out)
{
    int T = 2 * half_len + 1;
    if (T > 64 || len <= 0) return -1;
    double w[64], off[64];
    for (int i = 0; i < T; i++) off[i] = (double)(i - half_len);
    for (int i = 0; i < T; i++) {                    /* :45-50 */
        w[i] = 1.0;
        for (int j = 0; j < T; j++) {
            if (i == j) continue;
            w[i] *= 1.0 / (off[i] - off[j]);
        }
    }
    for (int64_t p = 0; p < P; p++) {
        int64_t base = (int64_t)floor(pos[p]);       /* :55 */
        double mu = pos[p] - (double)base;           /* :56 */
        c128 s[64];
        double diffs[64];
        int exact = -1;
        for (int i = 0; i < T; i++) {
            int64_t ix = base + (i - half_len);
            if (ix < 0) ix = 0;                      /* :59 clip */
            if (ix > len - 1) ix = len - 1;
            s[i] = sig[ix];
            diffs[i] = mu - off[i];                  /* :62 */
            if (exact < 0 && fabs(diffs[i]) < 1e-12) exact = i;   /* :63 */
        }
        if (exact >= 0) { out[p] = s[exact]; continue; }          /* :64-66 */
        /* :68 numer = sum(weights * samples / diffs): (w*s) then * (1/d) */
        double tmp[128], den[64];
        for (int i = 0; i < T; i++) {
            double pr = w[i] * s[i].re - 0.0 * s[i].im;
            double pi = w[i] * s[i].im + 0.0 * s[i].re;
            double scl = 1.0 / (diffs[i] + 0.0 * 0.0);
            tmp[2 * i] = (pr + pi * 0.0) * scl;
            tmp[2 * i + 1] = (pi - pr * 0.0) * scl;
            den[i] = w[i] / diffs[i];                /* :69 */
        }
        double nr, ni;
        np_pairwise_c(&nr, &ni, tmp, 2 * T);
        nr = 0.0 + nr; ni = 0.0 + ni;
        double d = orc_sum(den, T);
        double scl = 1.0 / (d + 0.0 * 0.0);          /* :70 complex / real */
        out[p].re = (nr + ni * 0.0) * scl;
        out[p].im = (ni - nr * 0.0) * scl;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a8: read_iq_int16(path, normalize=True) -- hw1/gardner.py:77-91, on an
 * in-memory word buffer.  float32 arithmetic throughout, as numpy does for
 * complex64: peak = max(hypotf(i,q)); iq * (1.0f/peak) (Smith division).
 * Returns the number of complex samples, -1 if n_words < 2 (ValueError :80).  */
int64_t orc_read_iq_int16(const int16_t *raw, int64_t n_words, int normalize, c64 *out)
{
    if (n_words < 2) return -1;
    int64_t n = n_words / 2;                         /* :82-83 drop odd word */
    float peak = 0.0f;
    for (int64_t i = 0; i < n; i++) {
        out[i].re = (float)raw[2 * i];
        out[i].im = (float)raw[2 * i + 1];
        float a = hypotf(out[i].re, out[i].im);
        if (a > peak) peak = a;
    }
    if (normalize && peak > 0) {                     /* :87-90 */
        float scl = 1.0f / (peak + 0.0f * 0.0f);
        for (int64_t i = 0; i < n; i++) {
            float r = out[i].re, im = out[i].im;
            out[i].re = (r + im * 0.0f) * scl;
            out[i].im = (im - r * 0.0f) * scl;
        }
    }
    return n;
}

/* float32 loader idiom: s[::2] + 1j*s[1::2] -> complex64, no scaling
 * (read_file.py:108-109, gardner_real_file_2.py:9-10, diff_method.py:165-166). */
int64_t orc_read_iq_f32(const float *raw, int64_t n_words, c64 *out)
{
    int64_t n = n_words / 2;
    for (int64_t i = 0; i < n; i++) { out[i].re = raw[2 * i]; out[i].im = raw[2 * i + 1]; }
    return n;
}

/* ------------------------------------------------------------------------- */
/* Composite used by the parity tests and by bench.py's cpu_baseline leg: the
 * receiver chain of worker.py:299-304 on ONE burst of complex64 samples that
 * are ALREADY matched-filtered and normalised (the unit the north-star metric
 * counts): upcast -> gardner_timing_recovery(alpha, mu0) -> fll_func(Bn, y)
 * -> quadrants.  scale multiplies every sample first (1.0 = none; the device
 * path folds 1/std in the same way).  Any output pointer may be NULL.
 * Returns n_rec (= number of FLL symbols), <0 on error.                       */
int64_t orc_sync_chain_c64(const c64 *x, int64_t N, double scale, int sps, double alpha,
                           double mu0, double Bn, double yy,
                           c128 *rec, double *err, double *muh,
                           c128 *sig, double *freq, double *ef, uint8_t *dec)
{
    if (N <= 0) return -1;
    int64_t cap = N / (sps > 0 ? sps : 1) + 8;
    c128 *xx = (c128 *)malloc(sizeof(c128) * N);
    c128 *r = rec ? rec : (c128 *)malloc(sizeof(c128) * cap);
    double *e = err ? err : (double *)malloc(sizeof(double) * cap);
    double *m = muh ? muh : (double *)malloc(sizeof(double) * cap);
    for (int64_t i = 0; i < N; i++) {
        xx[i].re = (double)x[i].re * scale;
        xx[i].im = (double)x[i].im * scale;
    }
    int64_t n_rec = orc_gardner2(xx, N, sps, alpha, mu0, r, e, m, cap);
    if (n_rec > 0) {
        c128 *s = sig ? sig : (c128 *)malloc(sizeof(c128) * n_rec);
        orc_fll(r, n_rec, Bn, yy, s, freq, ef);
        if (dec) orc_quadrants(s, n_rec, dec);
        if (!sig) free(s);
    }
    free(xx);
    if (!rec) free(r);
    if (!err) free(e);
    if (!muh) free(m);
    return n_rec;
}

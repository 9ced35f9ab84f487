// hostmath.cpp -- TEST HARNESS (not product): compiles the device math headers
// (dsp_amd/csrc/dsp_math.h, sync_lane.h) for the host CPU so that the exact
// source the HIP kernels run per lane can be checked against the oracle on a
// machine without a GPU.  Loaded only by tests/test_hostmath.py.
#include "../../dsp_amd/csrc/sync_lane.h"

using namespace dspm;

static const MathK KK = math_consts_literal();

extern "C" {

void hm_sincos(const double *x, int64_t n, double *s, double *c)
{
    for (int64_t i = 0; i < n; i++) sincos_f64(KK, x[i], s[i], c[i]);
}

void hm_atan2(const double *y, const double *x, int64_t n, double *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = atan2_f64(KK, y[i], x[i], false);
}

// fll_func over a complex128 array with the device step function
double hm_fll(const double *x, int64_t M, double kp, double ki, int64_t i0, double *state,
              double *out, double *freq, double *ef, uint8_t *quad)
{
    double f = state ? state[0] : 0.0, integ = state ? state[1] : 0.0;
    for (int64_t i = 0; i < M; i++) {
        FllOut o = fll_step(KK, x[2 * i], x[2 * i + 1], (double)(i + i0), f, integ, kp, ki);
        out[2 * i] = o.sr; out[2 * i + 1] = o.si; freq[i] = o.freq; ef[i] = o.ef;
        quad[i] = (uint8_t)o.quad;
    }
    if (state) { state[0] = f; state[1] = integ; }
    return f;
}

// the same loop through the restated step (fll_step_fast) in blocks of `blk` symbols with the kernel's
// rule: a block in which a component of a derotated sample was exactly zero (or the quadrant index
// left the int32 range) is run again from its start state through fll_step.  Returns the number of
// blocks that were run again.
int64_t hm_fll_fast(const double *x, int64_t M, double kp, double ki, int64_t i0, int64_t blk, double *out,
                    double *freq, double *ef, uint8_t *quad, double *final_freq)
{
    double f = 0.0, integ = 0.0;
    int64_t redone = 0;
    for (int64_t b0 = 0; b0 < M; b0 += blk) {
        const int64_t n = M - b0 < blk ? M - b0 : blk;
        const double f0 = f, g0 = integ;
        FllEvidence ev;
        fll_evidence_init(ev);
        for (int64_t i = b0; i < b0 + n; i++) {
            const double idx = (double)(i + i0);
            FllOut o = fll_step_fast(KK, x[2 * i], x[2 * i + 1], idx, 4.0 * idx, f, integ, kp, ki, ev);
            out[2 * i] = o.sr; out[2 * i + 1] = o.si; freq[i] = o.freq; ef[i] = o.ef; quad[i] = (uint8_t)o.quad;
        }
        if (fll_evidence_bad(ev)) {
            f = f0; integ = g0; redone++;
            for (int64_t i = b0; i < b0 + n; i++) {
                FllOut o = fll_step(KK, x[2 * i], x[2 * i + 1], (double)(i + i0), f, integ, kp, ki);
                out[2 * i] = o.sr; out[2 * i + 1] = o.si; freq[i] = o.freq; ef[i] = o.ef; quad[i] = (uint8_t)o.quad;
            }
        }
    }
    *final_freq = f;
    return redone;
}

struct ArrFetch {
    const double *x; double scale;
    cd operator()(int row) const { cd v; v.re = x[2 * row] * scale; v.im = x[2 * row + 1] * scale; return v; }
};

// gardner_timing_recovery with the device per-lane iteration
int64_t hm_gardner2(const double *x, int64_t len, int sps, double alpha, double mu0, int half_len,
                    double scale, double *rec, double *err, double *mu, int64_t cap)
{
    if (len <= 0) return -1;
    ArrFetch f{x, scale};
    LagrangeW L; const LagrangeW *Lp = nullptr;
    if (half_len > 0) { lagrange_weights(half_len, L); Lp = &L; }
    GState st; st.mu = mu0; st.k = sps; st.active = true;
    cd r0 = f(0);
    rec[0] = r0.re; rec[1] = r0.im;
    int64_t n = 0;
    while (st.active && n + 1 < cap) {
        GOut o = gardner2_iter(st, (int)len, sps, alpha, Lp, f);
        if (!o.produced) break;
        err[n] = o.err; mu[n] = o.mu;
        rec[2 * (n + 1)] = o.rec.re; rec[2 * (n + 1) + 1] = o.rec.im;
        n++;
    }
    return n + 1;
}

int64_t hm_gardner_ted(const double *x, int64_t len, int sps, double delta_mu, int half_len,
                       double *y, double *err, double *mu, int64_t cap)
{
    ArrFetch f{x, 1.0};
    LagrangeW L; const LagrangeW *Lp = nullptr;
    if (half_len > 0) { lagrange_weights(half_len, L); Lp = &L; }
    int64_t ns = len - 2 * (int64_t)sps;
    int64_t n_symbols = ns >= 0 ? ns / sps : -((-ns + sps - 1) / sps);
    GState st; st.mu = 0.0; st.k = 0; st.active = true;
    int64_t n = 0;
    while (st.active && n < cap) {
        GOut o = ted_iter(st, (int)len, sps, (int)n_symbols, delta_mu, Lp, f);
        if (!o.produced) break;
        err[n] = o.err; mu[n] = o.mu; y[2 * n] = o.rec.re; y[2 * n + 1] = o.rec.im;
        n++;
    }
    return n;
}

void hm_lagrange(const double *x, int64_t len, const double *pos, int64_t P, int half_len, double *out)
{
    ArrFetch f{x, 1.0};
    LagrangeW L; lagrange_weights(half_len, L);
    for (int64_t p = 0; p < P; p++) {
        cd z = lagrange_at(pos[p], (int)len, L, f);
        out[2 * p] = z.re; out[2 * p + 1] = z.im;
    }
}

}  // extern "C"

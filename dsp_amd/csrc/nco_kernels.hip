// nco_kernels.hip -- accumulating-phase FLL, the facade for the reference's class
// FrequencyLockedLoop (used by fll_real_signal.py:66-74,209-216 and f_rel.py:123-132).
//
// PARITY UNPINNED: the class' source file fll.py is absent from the reference (only a Python
// 3.13 .pyc remains), so there is nothing to restate.  What is known are the constructor
// arguments (detector_method in {cross_product, atan2, decision_directed}, Kp, Ki, freq_limit),
// the call  corrected, freq_estimate = fll.process_signal(x)  and the history keys errors /
// freq_estimates / phases.  The loop below is OUR definition of such a loop (documented in
// DESIGN.md); it is checked against a CPU restatement of the same definition only.
//
//   phase_0 = 0, freq_0 = 0, integ_0 = 0, y_{-1} = 0
//   y_n   = x_n * exp(-j * phase_n)
//   e_n   = cross_product:      Im(y_n * conj(y_{n-1})) / (|y_n| |y_{n-1}| + 1e-12)
//           atan2:              angle(y_n * conj(y_{n-1}))
//           decision_directed:  angle(y_n * conj(d_n)),  d_n = sign(Re y_n) + j sign(Im y_n)
//   integ = clip(integ + Ki * e_n, -freq_limit, freq_limit)           (anti-windup)
//   freq  = clip(Kp * e_n + integ, -freq_limit, freq_limit)           (cycles / sample)
//   phase_{n+1} = wrap(phase_n + 2 pi freq)   into (-pi, pi]
//   history: errors[n] = e_n, freq_estimates[n] = freq, phases[n] = phase_{n+1}
//   freq_estimate returned = mean of freq over the last quarter of the samples
// One lane per burst over the tiled layout, float64 state in VGPRs.
#include "dsp_internal.h"
#include "dsp_math.h"

#pragma clang fp contract(off)

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ const double k_nco_table[dspm::K_COUNT] = DSP_MATH_TABLE_INIT;

template <typename IN, typename OS, typename OC>
__global__ __launch_bounds__(64) void fll_class_kernel(const IN *__restrict__ x, const int *__restrict__ lens,
                                                       int len_all, int B, int rows, int detector, double Kp,
                                                       double Ki, double flim, OC *__restrict__ out,
                                                       OS *__restrict__ errs, OS *__restrict__ freqs,
                                                       OS *__restrict__ phases, double *__restrict__ freq_final)
{
    const int lane = threadIdx.x, tile = blockIdx.x, b = tile * 64 + lane;
    const int len = b < B ? (lens ? lens[b] : len_all) : 0;
    int max_len = len;
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(max_len, d); max_len = o > max_len ? o : max_len; }
    dspm::MathK K;
    for (int i = 0; i < dspm::K_COUNT; i++) K.v[i] = k_nco_table[i];
    const size_t base = (size_t)tile * rows * 64 + lane;
    const double two_pi = K.v[dspm::K_TWO_PI], pi = K.v[dspm::K_PI_HI];
    double phase = 0.0, integ = 0.0, pr = 0.0, pi_prev = 0.0, facc = 0.0;
    const int tail0 = len - (len + 3) / 4;           // last quarter
    for (int n = 0; n < max_len; n++) {
        const IN v = x[base + (size_t)n * 64];
        const bool live = n < len;
        const double xr = (double)v.x, xi = (double)v.y;
        double s, c;
        dspm::sincos_f64(K, phase, s, c);
        const double yr = xr * c + xi * s, yi = xi * c - xr * s;
        double e;
        if (detector == 2) {
            const double dr = yr > 0.0 ? 1.0 : (yr < 0.0 ? -1.0 : 0.0), di = yi > 0.0 ? 1.0 : (yi < 0.0 ? -1.0 : 0.0);
            e = dspm::atan2_f64(K, yi * dr - yr * di, yr * dr + yi * di, false);
        } else {
            const double zr = yr * pr + yi * pi_prev, zi = yi * pr - yr * pi_prev;
            if (detector == 1) {
                e = dspm::atan2_f64(K, zi, zr, false);
            } else {
                const double m = sqrt(yr * yr + yi * yi) * sqrt(pr * pr + pi_prev * pi_prev) + 1e-12;
                e = zi / m;
            }
        }
        double ni = integ + Ki * e;
        ni = ni > flim ? flim : (ni < -flim ? -flim : ni);
        double nf = Kp * e + ni;
        nf = nf > flim ? flim : (nf < -flim ? -flim : nf);
        double np = phase + two_pi * nf;
        np = np > pi ? np - two_pi : (np <= -pi ? np + two_pi : np);
        if (live) {
            integ = ni; phase = np; pr = yr; pi_prev = yi;
            if (n >= tail0) facc += nf;
        }
        const size_t at = base + (size_t)n * 64;
        if (out) out[at] = live ? OC{(OS)yr, (OS)yi} : OC{0, 0};
        if (errs) errs[at] = live ? (OS)e : (OS)0;
        if (freqs) freqs[at] = live ? (OS)nf : (OS)0;
        if (phases) phases[at] = live ? (OS)np : (OS)0;
    }
    if (b < B && freq_final) freq_final[b] = len > 0 ? facc / (double)(len - tail0) : 0.0;
}

// ---------------------------------------------------------------------------------------------
// fll_func with the constellation switch of worker.Processing.fll_func (worker.py:18-78; SURVEY.md
// 8f rank 4).  The loop is fll_func.py's (absolute-index phase 2 pi f i, PI filter kp / ki); only
// the slicer changes:
//   kind 0  bpsk    decided = sign(Re) + 0j                                        :36
//   kind 1  qpsk    decided = (sign(Re) + j sign(Im)) / sqrt(2)                    :38
//   kind 2  table   the point of `points` nearest to the sample, np.abs distance, the first of
//                   equally near points wins (strict <)                            :40-66
// error = angle(sample * conj(decided)) for all of them (:70).  dec receives the decision index:
// table position; qpsk quadrant (Re<0) + 2 (Im<0); bpsk 0 / 1 / 2 for Re > 0 / < 0 / == 0.
// One lane per burst over the tiled layout; the table (<= 64 points) sits in LDS.
// np.abs of a complex128 as numpy 2.2 computes it on an FMA-capable host (checked bit for bit on 2e6
// random values, tests/golden/make_golden_slicers.py environment): big * sqrt(fma(r, r, 1)) with
// r = small / big -- NOT a correctly rounded hypot.  The nearest-point search compares these values,
// so (nearly) equidistant points resolve exactly as in the reference.
__device__ __forceinline__ double abs_np(double a, double b)
{
    double x = __builtin_fabs(a), y = __builtin_fabs(b);
    if (x < y) { const double t = x; x = y; y = t; }
    if (x == 0.0) return 0.0;
    const double r = y / x;
    return x * __builtin_sqrt(__builtin_fma(r, r, 1.0));
}

// np.argmin(np.abs(points - sample)): the first of the nearest points.  abs_np costs a division and a square
// root per point; squared distances (four operations) decide all but near-ties: a point whose squared distance
// exceeds the smallest by more than 1e-12 of it cannot win or tie under abs_np (both are the true distance to a
// few 1e-16), so only points inside that margin go through abs_np, in index order with the strict < of the
// reference -- the same index as the full scan, bit for bit.  (NaN samples: every comparison is false, index 0,
// np.argmin's answer for an all-NaN row.)
template <int NPF>
__device__ __forceinline__ int nearest_point(const d2v *pt, int n_points, double sr, double si, bool live)
{
    double d2[NPF];
    double best2 = __builtin_inf();
    int idx = 0;
#pragma unroll
    for (int j = 0; j < NPF; j++) {
        const double ex = pt[j].x - sr, ey = pt[j].y - si;
        d2[j] = __builtin_fma(ex, ex, ey * ey);
        const bool lt = (j < n_points) & (d2[j] < best2);
        best2 = lt ? d2[j] : best2;
        idx = lt ? j : idx;
    }
    const double lim = best2 * (1.0 + 1e-12);
    int ncand = 0;
#pragma unroll
    for (int j = 0; j < NPF; j++) ncand += ((j < n_points) & (d2[j] <= lim)) ? 1 : 0;
    // (lanes whose result nobody reads -- padding lanes, lanes past their burst's end: all-zero samples, from which
    //  every point of a PSK ring is equally far -- do not send the wave through the exact comparison)
    if (__any(live && ncand > 1)) {
        if (ncand > 1) {
            // (unrolled with static indices: a run-time index would send pt[] and d2[] to scratch memory)
            double best = __builtin_inf();
#pragma unroll
            for (int j = 0; j < NPF; j++) {
                const double dist = abs_np(pt[j].x - sr, pt[j].y - si);          // np.abs(d - sample)
                const bool lt = (j < n_points) & (d2[j] <= lim) & (dist < best);
                best = lt ? dist : best;
                idx = lt ? j : idx;
            }
        }
    }
    return idx;
}

__device__ __forceinline__ int nearest_point_lds(const d2v *tab, int n_points, double sr, double si)
{
    double best2 = __builtin_inf();
    int idx = 0;
    for (int j = 0; j < n_points; j++) {
        const double ex = tab[j].x - sr, ey = tab[j].y - si;
        const double d2 = __builtin_fma(ex, ex, ey * ey);
        const bool lt = d2 < best2;
        best2 = lt ? d2 : best2;
        idx = lt ? j : idx;
    }
    const double lim = best2 * (1.0 + 1e-12);
    double best = __builtin_inf();
    int first = -1;
    for (int j = 0; j < n_points; j++) {
        const double ex = tab[j].x - sr, ey = tab[j].y - si;
        if (!(__builtin_fma(ex, ex, ey * ey) <= lim)) continue;
        if (first < 0) { first = j; continue; }                                  // a lone candidate needs no abs_np
        if (best == __builtin_inf()) {
            best = abs_np(tab[first].x - sr, tab[first].y - si);
            idx = first;
        }
        const double dist = abs_np(ex, ey);
        if (dist < best) { best = dist; idx = j; }
    }
    return idx;
}

template <typename IN, typename OS, typename OC, int KIND>
__global__ __launch_bounds__(64) void fll_slicer_kernel(const IN *__restrict__ x, const int *__restrict__ lens,
                                                        int len_all, int B, int rows,
                                                        const d2v *__restrict__ points, int n_points, double kp,
                                                        double ki, OC *__restrict__ out, OS *__restrict__ freqs,
                                                        OS *__restrict__ efs, unsigned char *__restrict__ dec,
                                                        double *__restrict__ freq_final)
{
    __shared__ d2v tab[64];
    const int lane = threadIdx.x, tile = blockIdx.x, b = tile * 64 + lane;
    if (lane < n_points) tab[lane] = points[lane];
    __syncthreads();
    // (tables of up to 16 points -- every constellation of the reference -- are held in registers: the search
    //  below is straight-line code; longer tables are read from LDS point by point)
    constexpr int NPF = 16;
    d2v pt[NPF];
#pragma unroll
    for (int j = 0; j < NPF; j++) pt[j] = (KIND >= 2 && j < n_points) ? points[j] : d2v{0.0, 0.0};
    const int len = b < B ? (lens ? lens[b] : len_all) : 0;
    int max_len = len;
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(max_len, d); max_len = o > max_len ? o : max_len; }
    dspm::MathK K;
    for (int i = 0; i < dspm::K_COUNT; i++) K.v[i] = k_nco_table[i];
    const size_t base = (size_t)tile * rows * 64 + lane;
    double curr_freq = 0.0, integ = 0.0;
    for (int n = 0; n < max_len; n++) {
        const IN v = x[base + (size_t)n * 64];
        const bool live = n < len;
        const double xr = (double)v.x, xi = (double)v.y;
        const double phase = K.v[dspm::K_TWO_PI] * curr_freq * (double)n;       // :31
        double s, c;
        dspm::sincos_f64(K, phase, s, c);
        const double sr = xr * c + xi * s, si = xi * c - xr * s;                 // :32
        double dr, di;
        int idx;
        if constexpr (KIND == 0) {
            dr = sr > 0.0 ? 1.0 : (sr < 0.0 ? -1.0 : 0.0); di = 0.0;
            idx = sr > 0.0 ? 0 : (sr < 0.0 ? 1 : 2);
        } else if constexpr (KIND == 1) {
            dr = sr > 0.0 ? K.v[dspm::K_SCL] : (sr < 0.0 ? -K.v[dspm::K_SCL] : 0.0);
            di = si > 0.0 ? K.v[dspm::K_SCL] : (si < 0.0 ? -K.v[dspm::K_SCL] : 0.0);
            idx = (int)(sr < 0.0) + 2 * (int)(si < 0.0);
        } else {
            if (n_points <= 8) idx = nearest_point<8>(pt, n_points, sr, si, live);
            else if (n_points <= NPF) idx = nearest_point<NPF>(pt, n_points, sr, si, live);
            else idx = nearest_point_lds(tab, n_points, sr, si);
            dr = tab[idx].x; di = tab[idx].y;
        }
        const double zr = sr * dr + si * di, zi = si * dr - sr * di;             // sample * conj(decided)
        const double error = dspm::atan2_f64(K, zi, zr, false);                  // :70
        if (live) {
            integ += error * ki;                                                 // :71
            const double ef_n = error * kp + integ;                              // :72
            curr_freq += ef_n;                                                   // :74
            const size_t at = base + (size_t)n * 64;
            if (out) out[at] = OC{(OS)sr, (OS)si};
            if (freqs) freqs[at] = (OS)curr_freq;
            if (efs) efs[at] = (OS)ef_n;
            if (dec) dec[at] = (unsigned char)idx;
        }
    }
    if (b < B && freq_final) freq_final[b] = curr_freq;
}

}  // namespace

hipError_t launch_fll_class(const void *x, int c128, const int *lens, int len_all, int B, int rows, int detector,
                            double Kp, double Ki, double flim, void *out, void *errs, void *freqs, void *phases,
                            double *freq_final, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    const int nt = (B + 63) / 64;
    if (c128)
        hipLaunchKernelGGL((fll_class_kernel<d2v, double, d2v>), dim3(nt), dim3(64), 0, s, (const d2v *)x, lens, len_all,
                           B, rows, detector, Kp, Ki, flim, (d2v *)out, (double *)errs, (double *)freqs,
                           (double *)phases, freq_final);
    else
        hipLaunchKernelGGL((fll_class_kernel<f2v, float, f2v>), dim3(nt), dim3(64), 0, s, (const f2v *)x, lens, len_all,
                           B, rows, detector, Kp, Ki, flim, (f2v *)out, (float *)errs, (float *)freqs,
                           (float *)phases, freq_final);
    return hipGetLastError();
}

hipError_t launch_fll_slicer(const void *x, int c128, const int *lens, int len_all, int B, int rows, int kind,
                             const void *points, int n_points, double kp, double ki, void *out, void *freqs,
                             void *efs, unsigned char *dec, double *freq_final, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    const int nt = (B + 63) / 64;
#define DSP_SLICER_LAUNCH(KIND)                                                                                      \
    do {                                                                                                             \
        if (c128)                                                                                                    \
            hipLaunchKernelGGL((fll_slicer_kernel<d2v, double, d2v, KIND>), dim3(nt), dim3(64), 0, s, (const d2v *)x, \
                               lens, len_all, B, rows, (const d2v *)points, n_points, kp, ki, (d2v *)out,            \
                               (double *)freqs, (double *)efs, dec, freq_final);                                     \
        else                                                                                                         \
            hipLaunchKernelGGL((fll_slicer_kernel<f2v, float, f2v, KIND>), dim3(nt), dim3(64), 0, s, (const f2v *)x,  \
                               lens, len_all, B, rows, (const d2v *)points, n_points, kp, ki, (f2v *)out,            \
                               (float *)freqs, (float *)efs, dec, freq_final);                                       \
    } while (0)
    if (kind == 0) DSP_SLICER_LAUNCH(0);
    else if (kind == 1) DSP_SLICER_LAUNCH(1);
    else DSP_SLICER_LAUNCH(2);
#undef DSP_SLICER_LAUNCH
    return hipGetLastError();
}

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

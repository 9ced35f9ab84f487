// cfo_kernels.hip -- coarse carrier-frequency estimators and the pre-derotation that sit between
// the burst cut and the matched filter.
//
//   cfo_preamble_kernel      diff_method.py:267-292, worker.py:288-294: s = cut[::sps][:L],
//                            p = s * conj(preamble), phases = angle(p), np.unwrap, then
//                              (1) mean(diff(unwrapped))            (2) polyfit(n, unwrapped, 1)[0]
//                              (3) angle(mean(p[1:] * conj(p[:-1])))        each / (2 pi sps)
//   cfo_fourth_power_kernel  estimate_freq.py:4-58: z = s**4, angle(mean z[i] conj(z[i-1])) / (8 pi)
//   derotate_cut_kernel      worker.py:287,298 / diff_method.py:296-298: cut B windows out of one
//                            signal and multiply by exp(-1j 2 pi f_b (arange + index0_b)), complex128
//
// One wave per burst for the preamble estimators (the unwrap is a prefix sum over the wave), one
// workgroup per row for the fourth-power sum, one thread per sample for the derotation.  float64
// throughout, complex products as numpy forms them (four products, two sums, no FMA); sums are
// tree-ordered where numpy's are sequential or pairwise, which moves results by a few 1e-16 relative.
#include "dsp_internal.h"

#pragma clang fp contract(off)

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr double kPi = 3.141592653589793;
constexpr double kTwoPi = 6.283185307179586;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename IN>
__global__ __launch_bounds__(64) void cfo_preamble_kernel(const IN *__restrict__ x, const int64_t *__restrict__ offsets,
                                                          int sps, const d2v *__restrict__ pre, int L,
                                                          double *__restrict__ out)
{
    extern __shared__ double up[];                    // [L] unwrapped phases of this burst
    const int b = blockIdx.x, lane = threadIdx.x;
    const IN *s = x + offsets[b];
    double carry = 0.0;                               // corrections accumulated by earlier groups of 64
    double ph_prev = 0.0, pr_prev = 0.0, pi_prev = 0.0;   // last element of the earlier group
    double qr = 0.0, qi = 0.0;                        // this lane's share of sum p[i] conj(p[i-1])
    for (int base = 0; base < L; base += 64) {
        const int i = base + lane;
        const bool live = i < L;
        double pr = 0.0, pi = 0.0, ph = 0.0;
        if (live) {
            const IN v = s[(int64_t)i * sps];
            const d2v c = pre[i];
            const double xr = (double)v.x, xi = (double)v.y;
            pr = xr * c.x + xi * c.y;                 // s * conj(c)
            pi = xi * c.x - xr * c.y;
            ph = atan2(pi, pr);
        }
        double lr = __shfl_up(pr, 1), li = __shfl_up(pi, 1), lph = __shfl_up(ph, 1);
        if (lane == 0) { lr = pr_prev; li = pi_prev; lph = ph_prev; }
        double corr = 0.0;
        if (live && i > 0) {
            // np.unwrap: fold the jump into [-pi, pi) (python-style modulo), keep small jumps as they are
            const double d = ph - lph;
            double m = fmod(d + kPi, kTwoPi);
            if (m < 0.0) m += kTwoPi;
            m -= kPi;
            if (m == -kPi && d > 0.0) m = kPi;
            corr = fabs(d) < kPi ? 0.0 : m - d;
            qr += pr * lr + pi * li;                  // p[i] * conj(p[i-1])
            qi += pi * lr - pr * li;
        }
        double sc = corr;                             // inclusive prefix sum of the corrections
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(sc, o);
            if (lane >= o) sc += t;
        }
        sc += carry;
        if (live) up[i] = ph + sc;
        carry = __shfl(sc, 63);
        ph_prev = __shfl(ph, 63); pr_prev = __shfl(pr, 63); pi_prev = __shfl(pi, 63);
    }
    __syncthreads();
    double su = 0.0, sd = 0.0;
    for (int i = lane; i < L; i += 64) {
        su += up[i];
        if (i > 0) sd += up[i] - up[i - 1];
    }
    su = wave_sum(su); sd = wave_sum(sd);
    const double ubar = su / (double)L, nbar = 0.5 * (double)(L - 1);
    double sxy = 0.0, sxx = 0.0;
    for (int i = lane; i < L; i += 64) {
        const double dn = (double)i - nbar;
        sxy += dn * (up[i] - ubar);
        sxx += dn * dn;
    }
    sxy = wave_sum(sxy); sxx = wave_sum(sxx);
    qr = wave_sum(qr); qi = wave_sum(qi);
    if (lane == 0) {
        const double den = kTwoPi * (double)sps, m = (double)(L - 1);
        out[(size_t)b * 3 + 0] = sd / m / den;
        out[(size_t)b * 3 + 1] = sxy / sxx / den;
        out[(size_t)b * 3 + 2] = atan2(qi / m, qr / m) / den;
    }
}

__device__ __forceinline__ d2v cmul(d2v a, d2v b) { return d2v{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

template <typename IN>
__global__ __launch_bounds__(256) void cfo_fourth_power_kernel(const IN *__restrict__ x, int64_t stride,
                                                               const int *__restrict__ lens, int len_all,
                                                               int64_t num_samples, double *__restrict__ out)
{
    __shared__ double red[2][4];
    const int b = blockIdx.x;
    const int len = lens ? lens[b] : len_all;
    const int64_t n = num_samples < len ? num_samples : len;          // :32
    if (n < 2) {                                                       // :33-34
        if (threadIdx.x == 0) out[b] = 0.0;
        return;
    }
    const IN *s = x + (size_t)b * stride;
    double ar = 0.0, ai = 0.0;
    for (int64_t i = 1 + threadIdx.x; i < n; i += 256) {
        const d2v a = {(double)s[i].x, (double)s[i].y}, p = {(double)s[i - 1].x, (double)s[i - 1].y};
        const d2v a2 = cmul(a, a), p2 = cmul(p, p);
        const d2v z = cmul(a2, a2), w = cmul(p2, p2);                  // s**4: two squarings (:37)
        ar += z.x * w.x + z.y * w.y;                                   // z[i] * conj(z[i-1])   (:42)
        ai += z.y * w.x - z.x * w.y;
    }
    ar = wave_sum(ar); ai = wave_sum(ai);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = ar; red[1][w] = ai; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = (double)(n - 1);
        const double tr = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double ti = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        out[b] = atan2(ti / m, tr / m) / (8.0 * kPi);                  // :45-54
    }
}

template <typename IN>
__global__ __launch_bounds__(256) void derotate_cut_kernel(const IN *__restrict__ x, const int64_t *__restrict__ offsets,
                                                           const int64_t *__restrict__ index0,
                                                           const double *__restrict__ f, int f_stride,
                                                           int64_t n_cut, d2v *__restrict__ out, int64_t out_stride,
                                                           unsigned nblk)
{
    const int b = blockIdx.x / nblk;                 // nblk workgroups per burst, one flat grid
    const int64_t i = (blockIdx.x % nblk) * (int64_t)256 + threadIdx.x;
    if (i >= n_cut) return;
    const int64_t off = offsets ? offsets[b] : 0;
    const int64_t i0 = index0 ? index0[b] : off;
    const double w = kTwoPi * f[(size_t)b * f_stride];                // (2 pi) * f_rel, then * n
    const double th = w * (double)(i + i0);
    double sn, cs;
    sincos(th, &sn, &cs);
    const IN v = x[off + i];
    const double xr = (double)v.x, xi = (double)v.y;
    out[(size_t)b * out_stride + i] = d2v{xr * cs + xi * sn, xi * cs - xr * sn};   // x * (cos - j sin)
}

}  // namespace

hipError_t launch_cfo_preamble(const void *x, bool c128, const int64_t *offsets, int B, int sps, const void *pre,
                               int L, double *out, hipStream_t s)
{
    const size_t lds = (size_t)L * 8;
    if (c128)
        hipLaunchKernelGGL(cfo_preamble_kernel<d2v>, dim3(B), dim3(64), lds, s, (const d2v *)x, offsets, sps,
                           (const d2v *)pre, L, out);
    else
        hipLaunchKernelGGL(cfo_preamble_kernel<f2v>, dim3(B), dim3(64), lds, s, (const f2v *)x, offsets, sps,
                           (const d2v *)pre, L, out);
    return hipGetLastError();
}

hipError_t launch_cfo_fourth_power(const void *x, bool c128, int64_t stride, const int *lens, int len_all, int B,
                                   int64_t num_samples, double *out, hipStream_t s)
{
    if (c128)
        hipLaunchKernelGGL(cfo_fourth_power_kernel<d2v>, dim3(B), dim3(256), 0, s, (const d2v *)x, stride, lens,
                           len_all, num_samples, out);
    else
        hipLaunchKernelGGL(cfo_fourth_power_kernel<f2v>, dim3(B), dim3(256), 0, s, (const f2v *)x, stride, lens,
                           len_all, num_samples, out);
    return hipGetLastError();
}

hipError_t launch_derotate_cut(const void *x, bool c128, const int64_t *offsets, const int64_t *index0,
                               const double *f, int f_stride, int B, int64_t n_cut, void *out, int64_t out_stride,
                               hipStream_t s)
{
    const unsigned nblk = (unsigned)((n_cut + 255) / 256);
    if ((uint64_t)nblk * (uint64_t)B > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 grid(nblk * (unsigned)B);
    if (c128)
        hipLaunchKernelGGL(derotate_cut_kernel<d2v>, grid, dim3(256), 0, s, (const d2v *)x, offsets, index0, f,
                           f_stride, n_cut, (d2v *)out, out_stride, nblk);
    else
        hipLaunchKernelGGL(derotate_cut_kernel<f2v>, grid, dim3(256), 0, s, (const f2v *)x, offsets, index0, f,
                           f_stride, n_cut, (d2v *)out, out_stride, nblk);
    return hipGetLastError();
}

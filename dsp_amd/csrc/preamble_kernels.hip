// preamble_kernels.hip -- differential-correlation preamble search.
//
// find_preamble_offset (diff_method.py:92-162; the same loops in
// worker.Processing.find_preamble_offset_with_interpolate_dots, worker.py:180-245):
//   d_pre = pre[1:] * conj(pre[:-1])                                   :118
//   for phase in range(sps):  s = x[phase::sps];  d = s[1:] * conj(s[:-1])   :122-126
//   conv[phase][k] = sum(d[k:k+L] * conj(d_pre)) / np.std(d[k:k+L])   (0 where std == 0)  :139-148
// stored as complex64 (:137).  In the reference this O(N L) Python loop with one np.std per
// window is the receiver's wall-time hog; here every (phase, k) is one thread.  Arithmetic is
// float64 on the samples as given (complex multiplies as numpy does them: four products, two
// sums, no FMA); the window statistics are the exact two-pass np.std.
#include "dsp_internal.h"

#pragma clang fp contract(off)

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

template <typename IN>
__global__ __launch_bounds__(256) void diff_corr_kernel(const IN *__restrict__ x, int64_t n, int sps,
                                                        const d2v *__restrict__ dpre, int L,
                                                        f2v *__restrict__ conv, int64_t kstride)
{
    extern __shared__ d2v pre_s[];                    // conj(d_pre)
    for (int j = threadIdx.x; j < L; j += 256) pre_s[j] = d2v{dpre[j].x, -dpre[j].y};
    __syncthreads();
    const int phase = blockIdx.y;
    const int64_t k = blockIdx.x * (int64_t)256 + threadIdx.x;
    // symbols of this phase: x[phase::sps] has ns entries, d has ns - 1
    const int64_t ns = n > phase ? (n - phase + sps - 1) / sps : 0;
    const int64_t nd = ns - 1;
    if (k >= kstride) return;
    f2v out = {0.f, 0.f};
    if (k + L <= nd) {
        const IN *s = x + phase + k * sps;
        double cr = 0.0, ci = 0.0, mr = 0.0, mi = 0.0;
        double ar = (double)s[0].x, ai = (double)s[0].y;
        for (int j = 0; j < L; j++) {
            const IN vb = s[(int64_t)(j + 1) * sps];
            const double br = (double)vb.x, bi = (double)vb.y;
            // d = b * conj(a)
            const double dr = br * ar + bi * ai, di = bi * ar - br * ai;
            const d2v p = pre_s[j];
            cr += dr * p.x - di * p.y;
            ci += dr * p.y + di * p.x;
            mr += dr; mi += di;
            ar = br; ai = bi;
        }
        mr /= (double)L; mi /= (double)L;
        double m2 = 0.0;
        ar = (double)s[0].x; ai = (double)s[0].y;
        for (int j = 0; j < L; j++) {
            const IN vb = s[(int64_t)(j + 1) * sps];
            const double br = (double)vb.x, bi = (double)vb.y;
            const double dr = (br * ar + bi * ai) - mr, di = (bi * ar - br * ai) - mi;
            m2 += dr * dr + di * di;
            ar = br; ai = bi;
        }
        const double Y = sqrt(m2 / (double)L);
        if (Y > 0.0) out = f2v{(float)(cr / Y), (float)(ci / Y)};
    }
    conv[(int64_t)phase * kstride + k] = out;
}

}  // namespace

hipError_t launch_diff_corr(const void *x, int c128, int64_t n, int sps, const void *dpre, int L, void *conv,
                            int64_t kstride, hipStream_t s)
{
    if (kstride <= 0 || sps <= 0) return hipSuccess;
    dim3 grid((unsigned)((kstride + 255) / 256), sps), block(256);
    const size_t lds = (size_t)L * 16;
    if (c128) hipLaunchKernelGGL(diff_corr_kernel<d2v>, grid, block, lds, s, (const d2v *)x, n, sps, (const d2v *)dpre, L,
                                 (f2v *)conv, kstride);
    else hipLaunchKernelGGL(diff_corr_kernel<f2v>, grid, block, lds, s, (const f2v *)x, n, sps, (const d2v *)dpre, L,
                            (f2v *)conv, kstride);
    return hipGetLastError();
}

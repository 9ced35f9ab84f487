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

// Workgroup = DC_OUT consecutive sample positions i (all sps phases interleaved: phase = i % sps,
// k = i / sps), thread = DC_PER of them spaced 256 apart.  D[i] = x[i + sps] * conj(x[i]) -- the
// differential product of EVERY phase at once, since d_phase[k] = D[phase + k sps] -- is formed once per
// sample from coalesced loads and kept in LDS; conv[phase][k] = sum_j D[i + j sps] conj(d_pre[j]) / std
// is then a strided FIR over LDS with the exact two-pass np.std of the same L values.  The sums run
// in j order (as the reference's np.sum / the previous thread-per-output kernel did).
constexpr int DC_PER = 4, DC_OUT = 256 * DC_PER;

template <typename IN>
__global__ __launch_bounds__(256) void diff_corr_kernel(const IN *__restrict__ x, int64_t n, int sps,
                                                        const d2v *__restrict__ dpre, int L,
                                                        f2v *__restrict__ conv, int64_t kstride)
{
    extern __shared__ d2v smem_dc[];
    d2v *pre_s = smem_dc;                             // [L] conj(d_pre)
    d2v *D = smem_dc + L;                             // [DC_OUT + (L - 1) * sps]
    const int64_t i0 = blockIdx.x * (int64_t)DC_OUT;
    const int nD = DC_OUT + (L - 1) * sps;
    for (int j = threadIdx.x; j < L; j += 256) pre_s[j] = d2v{dpre[j].x, -dpre[j].y};
    for (int e = threadIdx.x; e < nD; e += 256) {
        const int64_t i = i0 + e;
        d2v d = {0.0, 0.0};
        if (i + sps < n) {
            const IN va = x[i], vb = x[i + sps];
            const double ar = (double)va.x, ai = (double)va.y, br = (double)vb.x, bi = (double)vb.y;
            d = d2v{br * ar + bi * ai, bi * ar - br * ai};            // b * conj(a)
        }
        D[e] = d;
    }
    __syncthreads();
    double cr[DC_PER], ci[DC_PER], mr[DC_PER], mi[DC_PER], m2[DC_PER];
#pragma unroll
    for (int o = 0; o < DC_PER; o++) { cr[o] = ci[o] = mr[o] = mi[o] = m2[o] = 0.0; }
    const int t = threadIdx.x;
    for (int j = 0; j < L; j++) {
        const d2v p = pre_s[j];
#pragma unroll
        for (int o = 0; o < DC_PER; o++) {
            const d2v d = D[t + 256 * o + j * sps];
            cr[o] += d.x * p.x - d.y * p.y;
            ci[o] += d.x * p.y + d.y * p.x;
            mr[o] += d.x; mi[o] += d.y;
        }
    }
#pragma unroll
    for (int o = 0; o < DC_PER; o++) { mr[o] /= (double)L; mi[o] /= (double)L; }
    for (int j = 0; j < L; j++) {
#pragma unroll
        for (int o = 0; o < DC_PER; o++) {
            const d2v d = D[t + 256 * o + j * sps];
            const double dr = d.x - mr[o], di = d.y - mi[o];
            m2[o] += dr * dr + di * di;
        }
    }
#pragma unroll
    for (int o = 0; o < DC_PER; o++) {
        const int64_t i = i0 + t + 256 * o;
        const int phase = (int)(i % sps);
        const int64_t k = i / sps;
        if (k >= kstride) continue;
        // symbols of this phase: x[phase::sps] has ns entries, d has ns - 1
        const int64_t ns = n > phase ? (n - phase + sps - 1) / sps : 0;
        const int64_t nd = ns - 1;
        f2v out = {0.f, 0.f};
        if (k + L <= nd) {
            const double Y = sqrt(m2[o] / (double)L);
            if (Y > 0.0) out = f2v{(float)(cr[o] / Y), (float)(ci[o] / Y)};
        }
        conv[(int64_t)phase * kstride + k] = out;
    }
}

// diff_corr_win_kernel: the same sums with the LDS reads shared between a thread's outputs.  A thread owns
// DW_KB CONSECUTIVE k of one phase, so the D values of its outputs overlap: at tap j output o needs
// d_phase[k0 + o + j], and a register window of DW_KB values, one new value per tap, serves all of them
// (3 L / DW_KB + 2 LDS reads per output instead of 2.25 L: diff_corr_kernel is bound by its LDS reads).
// D is kept per phase, Dp[phase][k] (row length odd, DW_KB odd: the lanes of a wave are consecutive
// k-blocks of one phase, DW_KB words apart, and spread over all banks).  Workgroup = kblk = 256 / sps
// k-blocks x sps phases = positions [K0 sps, (K0 + kblk DW_KB) sps).  Every sum runs in j order with the
// expressions of diff_corr_kernel: the same bits.
constexpr int DW_KB = 7;

template <typename IN>
__global__ __launch_bounds__(256) void diff_corr_win_kernel(const IN *__restrict__ x, int64_t n, int sps,
                                                            const d2v *__restrict__ dpre, int L,
                                                            f2v *__restrict__ conv, int64_t kstride, int kblk, int KP)
{
    constexpr int KB = DW_KB;
    extern __shared__ d2v smem_dc[];
    d2v *pre_s = smem_dc;                             // [L] conj(d_pre)
    d2v *Dp = smem_dc + L;                            // [sps][KP], KP >= kblk KB + L (one value past the last window is loaded, never used)
    const int nk = kblk * KB;                         // k per phase of this workgroup
    const int64_t K0 = (int64_t)blockIdx.x * nk;
    const int64_t i0 = K0 * sps;
    for (int j = threadIdx.x; j < L; j += 256) pre_s[j] = d2v{dpre[j].x, -dpre[j].y};
    const int ne = sps * (nk + L);                    // position order: coalesced loads
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int64_t i = i0 + e;
        const int kk = e / sps, p = e - kk * sps;
        d2v d = {0.0, 0.0};
        if (i + sps < n) {
            const IN va = x[i], vb = x[i + sps];
            const double ar = (double)va.x, ai = (double)va.y, br = (double)vb.x, bi = (double)vb.y;
            d = d2v{br * ar + bi * ai, bi * ar - br * ai};            // b * conj(a)
        }
        Dp[p * KP + kk] = d;
    }
    __syncthreads();
    const int phase = threadIdx.x / kblk, kb = threadIdx.x - phase * kblk;
    if (phase >= sps) return;                         // 256 is not a multiple of sps: the last threads idle
    const d2v *row = Dp + phase * KP + kb * KB;
    double cr[KB], ci[KB], mr[KB], mi[KB], m2[KB];
    d2v w[KB];
#pragma unroll
    for (int o = 0; o < KB; o++) { cr[o] = ci[o] = mr[o] = mi[o] = m2[o] = 0.0; w[o] = row[o]; }
    // tap j + jj: output o reads window slot (o + jj) % KB = D value j + jj + o; slot jj is then refilled
#define DW_TAP1(jj)                                                          \
    {                                                                        \
        const d2v pj = pre_s[j + jj];                                        \
        _Pragma("unroll") for (int o = 0; o < KB; o++) {                     \
            const d2v d = w[(o + jj) % KB];                                  \
            cr[o] += d.x * pj.x - d.y * pj.y;                                \
            ci[o] += d.x * pj.y + d.y * pj.x;                                \
            mr[o] += d.x; mi[o] += d.y;                                      \
        }                                                                    \
        w[jj] = row[j + jj + KB];                                            \
    }
    int j = 0;
    for (; j + KB <= L; j += KB) {
#pragma unroll
        for (int jj = 0; jj < KB; jj++) DW_TAP1(jj)
    }
#pragma unroll
    for (int jj = 0; jj < KB; jj++)
        if (j + jj < L) DW_TAP1(jj)
#undef DW_TAP1
#pragma unroll
    for (int o = 0; o < KB; o++) { mr[o] /= (double)L; mi[o] /= (double)L; w[o] = row[o]; }
#define DW_TAP2(jj)                                                          \
    {                                                                        \
        _Pragma("unroll") for (int o = 0; o < KB; o++) {                     \
            const d2v d = w[(o + jj) % KB];                                  \
            const double dr = d.x - mr[o], di = d.y - mi[o];                 \
            m2[o] += dr * dr + di * di;                                      \
        }                                                                    \
        w[jj] = row[j + jj + KB];                                            \
    }
    for (j = 0; j + KB <= L; j += KB) {
#pragma unroll
        for (int jj = 0; jj < KB; jj++) DW_TAP2(jj)
    }
#pragma unroll
    for (int jj = 0; jj < KB; jj++)
        if (j + jj < L) DW_TAP2(jj)
#undef DW_TAP2
    // symbols of this phase: x[phase::sps] has ns entries, d has ns - 1
    const int64_t ns = n > phase ? (n - phase + sps - 1) / sps : 0;
    const int64_t nd = ns - 1;
#pragma unroll
    for (int o = 0; o < KB; o++) {
        const int64_t k = K0 + kb * KB + o;
        if (k >= kstride) continue;
        f2v out = {0.f, 0.f};
        if (k + L <= nd) {
            const double Y = sqrt(m2[o] / (double)L);
            if (Y > 0.0) out = f2v{(float)(cr[o] / Y), (float)(ci[o] / Y)};
        }
        conv[(int64_t)phase * kstride + k] = out;
    }
}

// Fallback for preambles whose differential window does not fit LDS ((L - 1) * sps > ~9000): one
// thread per (phase, k) straight from global memory.
template <typename IN>
__global__ __launch_bounds__(256) void diff_corr_direct_kernel(const IN *__restrict__ x, int64_t n, int sps,
                                                        const d2v *__restrict__ dpre, int L,
                                                        f2v *__restrict__ conv, int64_t kstride)
{
    extern __shared__ d2v pre_direct[];
    d2v *pre_s = pre_direct;                    // conj(d_pre)
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

template <typename IN>
static hipError_t launch_diff_corr_win(const IN *x, int64_t n, int sps, const d2v *dpre, int L, f2v *conv, int64_t kstride,
                                       int kblk, int KP, size_t lds, hipStream_t s)
{
    const int64_t nk = (int64_t)kblk * DW_KB, nblk = (kstride + nk - 1) / nk;
    if (nblk > 0x7fffffff) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)diff_corr_win_kernel<IN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(diff_corr_win_kernel<IN>, dim3((unsigned)nblk), dim3(256), lds, s, x, n, sps, dpre, L, conv, kstride, kblk, KP);
    return hipGetLastError();
}

hipError_t launch_diff_corr(const void *x, int c128, int64_t n, int sps, const void *dpre, int L, void *conv,
                            int64_t kstride, int window, hipStream_t s)
{
    if (kstride <= 0 || sps <= 0) return hipSuccess;
    if (window && sps <= 256) {                                 // register-window kernel where its LDS rows fit
        const int kblk = 256 / sps;
        const int KP = (kblk * DW_KB + L) | 1;                   // odd row length
        const size_t lds_w = ((size_t)L + (size_t)sps * KP) * 16;
        if (lds_w <= 144 * 1024)
            return c128 ? launch_diff_corr_win((const d2v *)x, n, sps, (const d2v *)dpre, L, (f2v *)conv, kstride, kblk, KP, lds_w, s)
                        : launch_diff_corr_win((const f2v *)x, n, sps, (const d2v *)dpre, L, (f2v *)conv, kstride, kblk, KP, lds_w, s);
    }
    const int64_t total = kstride * sps;                       // positions i = phase + k sps to cover
    const int64_t nblk = (total + DC_OUT - 1) / DC_OUT;
    if (nblk > 0x7fffffff) return hipErrorInvalidValue;
    const size_t lds = (size_t)(L + DC_OUT + (size_t)(L - 1) * sps) * 16;
    if (lds > 144 * 1024) {                                    // long preamble: thread per (phase, k)
        dim3 g2((unsigned)((kstride + 255) / 256), sps);
        if (c128) hipLaunchKernelGGL(diff_corr_direct_kernel<d2v>, g2, dim3(256), (size_t)L * 16, s, (const d2v *)x, n, sps,
                                     (const d2v *)dpre, L, (f2v *)conv, kstride);
        else hipLaunchKernelGGL(diff_corr_direct_kernel<f2v>, g2, dim3(256), (size_t)L * 16, s, (const f2v *)x, n, sps,
                                (const d2v *)dpre, L, (f2v *)conv, kstride);
        return hipGetLastError();
    }
    dim3 grid((unsigned)nblk), block(256);
    if (c128) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)diff_corr_kernel<d2v>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(diff_corr_kernel<d2v>, grid, block, lds, s, (const d2v *)x, n, sps, (const d2v *)dpre, L,
                           (f2v *)conv, kstride);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)diff_corr_kernel<f2v>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(diff_corr_kernel<f2v>, grid, block, lds, s, (const f2v *)x, n, sps, (const d2v *)dpre, L,
                           (f2v *)conv, kstride);
    }
    return hipGetLastError();
}

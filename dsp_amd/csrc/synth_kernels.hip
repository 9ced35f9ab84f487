// synth_kernels.hip -- on-device synthetic QPSK burst generator for benchmarks and
// full-size tests (the recipe of SURVEY.md 8(d) / worker.py:339-427 with a counter-based
// RNG, so that 65 536+ bursts need neither host generation nor PCIe):
//   symbols uniform over (+-1 +-1j)/sqrt(2) -> zero-stuff x sps -> TX RRC 'same'
//   -> fractional delay tau ~ U(-0.5, 0.5) by linear interpolation (np.interp form)
//   -> CFO exp(j(2 pi f n + phi)), f ~ U(-5e-4, 5e-4), phi ~ U(0, 2 pi)
//   -> complex AWGN at Es/N0 dB -> complex64, burst-major [B][n_sym * sps].
// Every random number is splitmix64(splitmix64(seed + burst) + (stream << 40) + index), so a
// test can recompute the transmitted symbols on the host (dsp_amd/synth.py: device_symbols).
#include "dsp_internal.h"

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rnd(uint64_t key, uint64_t stream, uint64_t idx)
{
    return splitmix64(key + (stream << 40) + idx);
}
__device__ __forceinline__ double u01(uint64_t h) { return (double)((h >> 11) + 1) * (1.0 / 9007199254740992.0); }

__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, int64_t b0, int B, int n_sym, int sps,
                                                    const double *__restrict__ taps, int M, float sigma,
                                                    f2v *__restrict__ out)
{
    __shared__ float h[512];
    for (int i = threadIdx.x; i < M; i += 256) h[i] = (float)taps[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int N = n_sym * sps;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const uint64_t key = splitmix64(seed + (uint64_t)(b0 + b));
    const double tau = u01(rnd(key, 2, 0)) - 0.5;
    const double f = (u01(rnd(key, 2, 1)) - 0.5) * 1.0e-3;
    const double phi = u01(rnd(key, 2, 2)) * 6.283185307179586;
    // np.interp(t, t - tau, tx): position n + tau on the tx grid, clamped at both ends
    double pos = (double)n + tau;
    pos = pos < 0.0 ? 0.0 : (pos > (double)(N - 1) ? (double)(N - 1) : pos);
    int k = (int)pos;
    if (k > N - 2) k = N - 2;
    const float frac = (float)(pos - (double)k);
    const int start = (M - 1) / 2;
    float txr[2] = {0.f, 0.f}, txi[2] = {0.f, 0.f};
    for (int t = 0; t < 2; t++) {
        const int m = k + t + start;                 // tx[k+t] = sum_j h[j] u[m - j], u[4i] = sym[i]
        int j = m % sps;                             // first tap with (m - j) % sps == 0
        for (; j < M; j += sps) {
            const int i = (m - j) / sps;
            if (m - j < 0) break;
            if (i >= n_sym) continue;
            const unsigned q = (unsigned)(rnd(key, 1, (uint64_t)i) & 3);
            const float sr = (q & 1) ? -0.70710678f : 0.70710678f;
            const float si = (q & 2) ? -0.70710678f : 0.70710678f;
            txr[t] += h[j] * sr;
            txi[t] += h[j] * si;
        }
    }
    const float xr = txr[0] * (1.f - frac) + txr[1] * frac;
    const float xi = txi[0] * (1.f - frac) + txi[1] * frac;
    double ang = 6.283185307179586 * f * (double)n + phi;
    ang -= 6.283185307179586 * floor(ang * 0.15915494309189535);
    float sn, cs;
    __sincosf((float)ang, &sn, &cs);
    // AWGN: (N(0,sigma) + j N(0,sigma)) / sqrt(2)   (worker.py:425-427)
    const double u1 = u01(rnd(key, 3, (uint64_t)n)), u2 = u01(rnd(key, 4, (uint64_t)n));
    const float r = sqrtf(-2.f * __logf((float)u1)) * sigma * 0.70710678f;
    float ns, nc;
    __sincosf(6.2831853f * (float)u2, &ns, &nc);
    out[(size_t)b * N + n] = f2v{xr * cs - xi * sn + r * nc, xr * sn + xi * cs + r * ns};
}

// One continuous QPSK stream (config 5): the burst recipe without ends, a timing offset that
// drifts slowly (tau0 + drift * n samples) and one carrier offset; int16 I,Q (amplitude `amp`
// counts per unit) or complex64.  Symbol i of the stream is rnd(splitmix64(seed), 1, i) & 3.
template <bool I16>
__global__ __launch_bounds__(256) void synth_stream_kernel(uint64_t seed, int64_t n_total, int sps,
                                                           const double *__restrict__ taps, int M, float sigma,
                                                           double cfo, double tau0, double drift, float amp,
                                                           void *__restrict__ out)
{
    __shared__ float h[512];
    for (int i = threadIdx.x; i < M; i += 256) h[i] = (float)taps[i];
    __syncthreads();
    const uint64_t key = splitmix64(seed);
    const int start = (M - 1) / 2;
    for (int64_t n = blockIdx.x * (int64_t)256 + threadIdx.x; n < n_total; n += (int64_t)gridDim.x * 256) {
        double pos = (double)n + tau0 + drift * (double)n;
        pos = pos < 0.0 ? 0.0 : pos;
        const int64_t k = (int64_t)pos;
        const float frac = (float)(pos - (double)k);
        float txr[2] = {0.f, 0.f}, txi[2] = {0.f, 0.f};
        for (int t = 0; t < 2; t++) {
            const int64_t m = k + t + start;
            int j = (int)(m % sps);
            for (; j < M; j += sps) {
                if (m - j < 0) break;
                const uint64_t i = (uint64_t)((m - j) / sps);
                const unsigned q = (unsigned)(rnd(key, 1, i) & 3);
                txr[t] += h[j] * ((q & 1) ? -0.70710678f : 0.70710678f);
                txi[t] += h[j] * ((q & 2) ? -0.70710678f : 0.70710678f);
            }
        }
        const float xr = txr[0] * (1.f - frac) + txr[1] * frac;
        const float xi = txi[0] * (1.f - frac) + txi[1] * frac;
        double cyc = cfo * (double)n;                     // carrier phase in cycles, reduced exactly
        cyc -= floor(cyc);
        float sn, cs;
        __sincosf((float)(6.283185307179586 * cyc), &sn, &cs);
        const double u1 = u01(rnd(key, 3, (uint64_t)n)), u2 = u01(rnd(key, 4, (uint64_t)n));
        const float r = sqrtf(-2.f * __logf((float)u1)) * sigma * 0.70710678f;
        float ns, nc;
        __sincosf(6.2831853f * (float)u2, &ns, &nc);
        const float yr = xr * cs - xi * sn + r * nc, yi = xr * sn + xi * cs + r * ns;
        if (I16) {
            float a = rintf(yr * amp), b = rintf(yi * amp);
            a = fminf(fmaxf(a, -32767.f), 32767.f);
            b = fminf(fmaxf(b, -32767.f), 32767.f);
            ((short2 *)out)[n] = make_short2((short)a, (short)b);
        } else {
            ((f2v *)out)[n] = f2v{yr, yi};
        }
    }
}

}  // namespace

hipError_t launch_synth_stream(uint64_t seed, int64_t n_total, int sps, const double *taps, int M, float sigma,
                               double cfo, double tau0, double drift, int i16, float amp, void *out, hipStream_t s)
{
    if (n_total <= 0) return hipSuccess;
    int64_t nb = (n_total + 255) / 256;
    if (nb > 65536) nb = 65536;
    if (i16) hipLaunchKernelGGL(synth_stream_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, seed, n_total, sps, taps, M,
                                sigma, cfo, tau0, drift, amp, out);
    else hipLaunchKernelGGL(synth_stream_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, seed, n_total, sps, taps, M,
                            sigma, cfo, tau0, drift, amp, out);
    return hipGetLastError();
}

hipError_t launch_synth(uint64_t seed, int64_t b0, int B, int n_sym, int sps, const double *taps, int M,
                        float sigma, void *out, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    dim3 grid((n_sym * sps + 255) / 256, B), block(256);
    hipLaunchKernelGGL(synth_kernel, grid, block, 0, s, seed, b0, B, n_sym, sps, taps, M, sigma, (f2v *)out);
    return hipGetLastError();
}

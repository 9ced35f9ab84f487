// Micro-benchmark: does the fp64 MFMA (v_mfma_f64_16x16x4_f64) run beside fp64 VALU FMAs on gfx950, or do the
// two share one fp64 datapath?  MI355X quotes the same 78.6 TFLOP/s for fp64 vector and fp64 matrix; a matched
// filter that fed both pipes at once would be worth building only if their rates ADD.
//   hipcc -O3 --offload-arch=gfx950 -w -o mfma_f64 mfma_f64.hip && ./mfma_f64
// One workgroup of 8 waves (two per SIMD).  mode 0: every wave issues MFMAs; 1: every wave issues FMAs;
// 2: waves 0-3 MFMAs, waves 4-7 FMAs (one of each per SIMD); 3: every wave alternates 1 MFMA + NF FMAs.
// Output: s_memtime ticks per wave for REP * iters units, and ticks per MFMA / per FMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 32

template <int MODE, int NF>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *cyc, int iters, double a, double b)
{
    const int wv = threadIdx.x >> 6;
    d4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wv < 4);
    const bool do_f = MODE == 1 || MODE == 3 || (MODE == 2 && wv >= 4);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (do_m) {
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n v_mfma_f64_16x16x4_f64 %1, %2, %3, %1"
                             : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
            }
            if (do_f) {
#pragma unroll
                for (int q = 0; q < NF; q++)
                    asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[threadIdx.x] = c0.x + c0.y + c0.z + c0.w + c1.x + c1.w + x0 + x1 + x2 + x3;
    if ((threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

template <int MODE, int NF>
static void run(const char *name, double *d_out, unsigned long long *d_cyc, int n_m, int n_f)
{
    const int iters = 200;
    unsigned long long h[8];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<MODE, NF>), dim3(1), dim3(512), 0, 0, d_out, d_cyc, iters, 1.0000001, 1e-9);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
    double units = (double)iters * REP;
    printf("%-46s", name);
    for (int w = 0; w < 8; w++) printf(" %7.1f", h[w] / units);
    printf("   ticks per unit per wave (unit = %d MFMA + %d FMA)\n", n_m, n_f);
}

int main()
{
    double *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_out, 512 * 8); hipMalloc(&d_cyc, 64);
    run<0, 4>("all waves: 2 MFMA f64 16x16x4", d_out, d_cyc, 2, 0);
    run<1, 4>("all waves: 16 FMA f64", d_out, d_cyc, 0, 16);
    run<1, 8>("all waves: 32 FMA f64", d_out, d_cyc, 0, 32);
    run<2, 4>("waves 0-3: 2 MFMA | waves 4-7: 16 FMA", d_out, d_cyc, 2, 16);
    run<2, 8>("waves 0-3: 2 MFMA | waves 4-7: 32 FMA", d_out, d_cyc, 2, 32);
    run<3, 4>("all waves: 2 MFMA + 16 FMA interleaved", d_out, d_cyc, 2, 16);
    run<3, 8>("all waves: 2 MFMA + 32 FMA interleaved", d_out, d_cyc, 2, 32);
    return 0;
}

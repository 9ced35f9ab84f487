// copy_bw.hip -- what a plain device-to-device copy reaches on this box, in a few shapes:
//   hipcc -O3 --offload-arch=gfx950 -o copy_bw copy_bw.hip && ./copy_bw [GiB=4]
// Prints read+write TB/s per variant.  Context for bench.py's same-box copy figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// grid-stride, one float4 per iteration
__global__ __launch_bounds__(256) void copy_gs(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// grid-stride, U float4 in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_un(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) b[i] = a[i];
}
// each workgroup owns one contiguous chunk
template <bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n, size_t per)
{
    const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        f4v v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)4095, n = bytes / 16;
    f4v *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %.3f TB/s (read + write)\n", name, 2.0 * bytes * 5 / (ms * 1e-3) / 1e12);
    };
    for (int g : {2048, 8192, 32768, 131072}) {
        char nm[64]; snprintf(nm, 64, "grid-stride x1, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL(copy_gs, dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    for (int g : {2048, 8192}) {
        char nm[64];
        snprintf(nm, 64, "grid-stride x4, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL((copy_un<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "grid-stride x8, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL((copy_un<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "grid-stride x4 nt, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL((copy_un<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "grid-stride x8 nt, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL((copy_un<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    for (int g : {2048, 16384}) {
        char nm[64];
        const size_t per = (n + g - 1) / g;
        snprintf(nm, 64, "chunk per WG, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL((copy_chunk<false>), dim3(g), dim3(256), 0, 0, a, b, n, per); });
        snprintf(nm, 64, "chunk per WG nt, %d WGs", g);
        time(nm, [&] { hipLaunchKernelGGL((copy_chunk<true>), dim3(g), dim3(256), 0, 0, a, b, n, per); });
    }
    time("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    hipFree(a); hipFree(b);
    return 0;
}

// Memory-only replica of the recurrence kernel's traffic: per tile one wave streams the input
// rows through an LDS ring with LDS-DMA, and the waves store output rows of the real sizes.
// No arithmetic: how fast can the memory system go with this access pattern and this few waves?
//   hipcc -O3 --offload-arch=gfx950 -w -o mem_pattern mem_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NW>   // vmcnt to leave in flight
__device__ __forceinline__ void waitv() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NW) : "memory"); }

// mode bits: 1 = read stream (DMA), 2 = G-wave stores (8+4+4 B), 4 = F-wave stores (8+4+4+1 B)
__global__ __launch_bounds__(128) void k(const uint8_t *x, uint8_t *rec, uint8_t *err, uint8_t *mu, uint8_t *sig,
                                         uint8_t *fr, uint8_t *ef, uint8_t *dec, int in_rows, int out_rows,
                                         int n_sym, int mode, int T, int spin, int var)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6, tile = blockIdx.x;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    float acc = lane;
    // var & 32: the seven output streams of a tile lie back to back in ONE buffer (rec), 2112 bytes per row
    // in all -- a tile then touches 3 two-MiB pages instead of 7 (does address translation limit the rate?)
    const bool grouped = (var & 32) != 0;
    const size_t tb = (size_t)out_rows * 2112;
    if (role == 0) {
        const uint8_t *xt = x + (size_t)tile * in_rows * 512 + lane * 16;
        const uint8_t *rt = grouped ? rec + tile * tb : rec + (size_t)tile * out_rows * 512,
                      *et = grouped ? rec + tile * tb + (size_t)out_rows * 512 : err + (size_t)tile * out_rows * 256,
                      *mt = grouped ? rec + tile * tb + (size_t)out_rows * 768 : mu + (size_t)tile * out_rows * 256;
        int F = 0;   // pieces (1 KiB = 2 rows) issued
        for (int it0 = 0; it0 < n_sym; it0 += T) {
            if (mode & 1) {
                if ((mode & 2) && !(var & 24)) { if (var & 4) waitv<6>(); else waitv<18>(); } else if (var & 24) waitv<8>(); else waitv<0>();
                for (int p = 0; p < 2 * T; p += 4, F += 4) {   // 2T pieces per block of T symbols (4 rows each)
                    const uint8_t *g = xt + (size_t)F * 1024;
                    unsigned off = lds_base + (unsigned)((F & 31) * 1024);
                    if (var & 2)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:1024 nt\n\tglobal_load_lds_dwordx4 %0, off offset:2048 nt\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:3072 nt" :: "v"(g), "s"(off) : "memory", "m0");
                    else
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\t"
                                 "global_load_lds_dwordx4 %0, off offset:3072" :: "v"(g), "s"(off) : "memory", "m0");
                }
            }
            for (int t = 0; t < T; t++) {
                const int it = it0 + t;
                for (int s = 0; s < spin; s++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
                if ((mode & 2) && (var & 24)) {
                    if ((it & 3) == 3) {
                        const int r0 = it - 3;
                        if (var & 16) {      // packed: 4 symbols per lane contiguous
                            f4v w = {acc, acc, acc, acc};
                            unsigned oa = lane * 32 + (r0 >> 2) * 2048, ob = lane * 16 + (r0 >> 2) * 1024;
                            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(oa), "v"(w), "s"(rt) : "memory");
                            asm volatile("global_store_dwordx4 %0, %1, %2 offset:16" :: "v"(oa), "v"(w), "s"(rt) : "memory");
                            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(ob), "v"(w), "s"(et) : "memory");
                            asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(ob), "v"(w), "s"(mt) : "memory");
                        } else {
                            for (int q = 0; q < 4; q++) {
                                f2v v = {acc, acc};
                                unsigned o8 = lane * 8 + (r0 + q + 1) * 512, o4 = lane * 4 + (r0 + q) * 256;
                                asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(o8), "v"(v), "s"(rt) : "memory");
                                asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(et) : "memory");
                                asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(mt) : "memory");
                            }
                        }
                    }
                } else
                if (mode & 2) {
                    f2v v = {acc, acc};
                    unsigned o8 = lane * 8 + (it + 1) * 512, o4 = lane * 4 + it * 256;
                    if (var & 4) {
                        f4v w = {acc, acc, acc, acc};
                        unsigned o16 = lane * 16 + it * 1024;
                        if (var & 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(o16), "v"(w), "s"(rt) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(o16), "v"(w), "s"(rt) : "memory");
                    } else if (var & 1) {
                    asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(o8), "v"(v), "s"(rt) : "memory");
                    asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(o4), "v"(acc), "s"(et) : "memory");
                    asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(o4), "v"(acc), "s"(mt) : "memory");
                    } else {
                    asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(o8), "v"(v), "s"(rt) : "memory");
                    asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(et) : "memory");
                    asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(mt) : "memory");
                    }
                }
            }
        }
    } else {
        const uint8_t *st = grouped ? rec + tile * tb + (size_t)out_rows * 1024 : sig + (size_t)tile * out_rows * 512,
                      *ft = grouped ? rec + tile * tb + (size_t)out_rows * 1536 : fr + (size_t)tile * out_rows * 256,
                      *et = grouped ? rec + tile * tb + (size_t)out_rows * 1792 : ef + (size_t)tile * out_rows * 256,
                      *dt = grouped ? rec + tile * tb + (size_t)out_rows * 2048 : dec + (size_t)tile * out_rows * 64;
        for (int it = 0; it < n_sym; it++) {
            for (int s = 0; s < spin; s++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
            if ((mode & 4) && (var & 24)) {
                if ((it & 3) == 3) {
                    const int r0 = it - 3;
                    unsigned q = lane & 3;
                    if (var & 16) {
                        f4v w = {acc, acc, acc, acc};
                        unsigned oa = lane * 32 + (r0 >> 2) * 2048, ob = lane * 16 + (r0 >> 2) * 1024, oc = lane * 4 + (r0 >> 2) * 256;
                        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(oa), "v"(w), "s"(st) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, %2 offset:16" :: "v"(oa), "v"(w), "s"(st) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(ob), "v"(w), "s"(ft) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(ob), "v"(w), "s"(et) : "memory");
                        asm volatile("global_store_dword %0, %1, %2" :: "v"(oc), "v"(q), "s"(dt) : "memory");
                    } else {
                        for (int k = 0; k < 4; k++) {
                            f2v v = {acc, acc};
                            unsigned o8 = lane * 8 + (r0 + k + 1) * 512, o4 = lane * 4 + (r0 + k + 1) * 256, o1 = lane + (r0 + k + 1) * 64;
                            asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(o8), "v"(v), "s"(st) : "memory");
                            asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(ft) : "memory");
                            asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(et) : "memory");
                            asm volatile("global_store_byte %0, %1, %2" :: "v"(o1), "v"(q), "s"(dt) : "memory");
                        }
                    }
                }
            } else
            if (mode & 4) {
                f2v v = {acc, acc};
                unsigned o8 = lane * 8 + (it + 1) * 512, o4 = lane * 4 + (it + 1) * 256, o1 = lane + (it + 1) * 64;
                unsigned q = lane & 3;
                if (var & 4) {
                    f4v w = {acc, acc, acc, acc};
                    unsigned o16 = lane * 16 + it * 1024;
                    if (var & 1) {
                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(o16), "v"(w), "s"(st) : "memory");
                    asm volatile("global_store_byte %0, %1, %2 nt" :: "v"(o1), "v"(q), "s"(dt) : "memory");
                    } else {
                    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(o16), "v"(w), "s"(st) : "memory");
                    asm volatile("global_store_byte %0, %1, %2" :: "v"(o1), "v"(q), "s"(dt) : "memory");
                    }
                } else if (var & 1) {
                asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(o8), "v"(v), "s"(st) : "memory");
                asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(o4), "v"(acc), "s"(ft) : "memory");
                asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(o4), "v"(acc), "s"(et) : "memory");
                asm volatile("global_store_byte %0, %1, %2 nt" :: "v"(o1), "v"(q), "s"(dt) : "memory");
                } else {
                asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(o8), "v"(v), "s"(st) : "memory");
                asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(ft) : "memory");
                asm volatile("global_store_dword %0, %1, %2" :: "v"(o4), "v"(acc), "s"(et) : "memory");
                asm volatile("global_store_byte %0, %1, %2" :: "v"(o1), "v"(q), "s"(dt) : "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) dec[0] = 1;
}

// plain streaming copy for comparison: every thread moves 16 B per step, grid-stride
__global__ __launch_bounds__(256) void copy_k(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 65536;
    const int ntiles = B / 64, n_sym = 2048;
    const int in_rows = argc > 2 ? atoi(argv[2]) : 8192, out_rows = argc > 3 ? atoi(argv[3]) : 2056;
    if (in_rows < 8192 || out_rows < n_sym + 2) { printf("in_rows >= 8192 and out_rows >= %d required\n", n_sym + 2); return 1; }
    uint8_t *x, *o[7];
    const size_t esz[7] = {16, 4, 4, 16, 4, 4, 1};
    CHECK(hipMalloc(&x, (size_t)ntiles * in_rows * 512 + 65536));
    CHECK(hipMemset(x, 1, (size_t)ntiles * in_rows * 512));
    for (int i = 0; i < 7; i++) CHECK(hipMalloc(&o[i], (size_t)ntiles * out_rows * 64 * esz[i] * (i == 0 ? 3 : 1)));   // (o[0] also holds the grouped layout)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int lds = 40960;
    const double bytes_r = (double)B * 8192 * 8, bytes_g = (double)B * 2048 * 16, bytes_f = (double)B * 2048 * 17;
    printf("B=%d tiles=%d in_rows=%d out_rows=%d\n", B, ntiles, in_rows, out_rows);
    {
        uint8_t *y;
        const size_t nb = (size_t)ntiles * in_rows * 512;
        CHECK(hipMalloc(&y, nb));
        for (int blocks : {2048, 8192, 32768}) {
            for (int r = 0; r < 2; r++) hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const f4v *)x, (f4v *)y, nb / 16);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const f4v *)x, (f4v *)y, nb / 16);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            printf("float4 copy of %.2f GB, %5d blocks: %7.3f ms  %6.2f TB/s (read + write)\n", nb / 1e9, blocks, ms, 2.0 * nb / ms / 1e9);
        }
        CHECK(hipFree(y));
    }
    for (int var : {0, 2, 3, 32}) { const int spin = 0;   // 2: nt loads, 3: nt loads + nt stores, 32: a tile's output streams contiguous
        for (int mode : {1, 6, 7}) {
            for (int r = 0; r < 2; r++) {
                hipLaunchKernelGGL(k, dim3(ntiles), dim3(128), lds, 0, x, o[0], o[1], o[2], o[3], o[4], o[5], o[6],
                                   in_rows, out_rows, n_sym, mode, 6, spin, var);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            const int n = 5;
            for (int r = 0; r < n; r++)
                hipLaunchKernelGGL(k, dim3(ntiles), dim3(128), lds, 0, x, o[0], o[1], o[2], o[3], o[4], o[5], o[6],
                                   in_rows, out_rows, n_sym, mode, 6, spin, var);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
            double by = (mode & 1 ? bytes_r : 0) + (mode & 2 ? bytes_g : 0) + (mode & 4 ? bytes_f : 0);
            printf("var %d spin %3d mode %d (%s%s%s): %7.3f ms  %6.2f TB/s\n", var, spin, mode, mode & 1 ? "R" : "-", mode & 2 ? "G" : "-",
                   mode & 4 ? "F" : "-", ms, by / ms / 1e9);
        }
    }
    return 0;
}

// Does the LDS-DMA destination base (M0) reach beyond 64 KiB of a workgroup's LDS on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_dma_range.hip -o tools/ubench/lds_dma_range && tools/ubench/lds_dma_range
// One wave; for each offset: global_load_lds_dwordx4 of 1 KiB to LDS byte address `off`, then the wave reads
// LDS at `off` with ds_read and writes what it finds to `out`.  The LDS is cleared to a marker first, so
// "landed", "not landed" and "landed elsewhere" (searched over the whole allocation) can be told apart.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(const uint4 *src, unsigned off, unsigned lds_bytes, uint4 *out, unsigned *found)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x;
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    for (unsigned i = lane * 4; i < lds_bytes; i += 256) *(lds_u32 *)(size_t)(base + i) = 0xdeadbeefu;
    __syncthreads();
    const void *g = (const void *)(src + lane);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)"
                 :: "v"(g), "s"(base + off) : "memory", "m0");
    __syncthreads();
    uint4 v;
    v.x = *(lds_u32 *)(size_t)(base + off + lane * 16);
    v.y = *(lds_u32 *)(size_t)(base + off + lane * 16 + 4);
    v.z = *(lds_u32 *)(size_t)(base + off + lane * 16 + 8);
    v.w = *(lds_u32 *)(size_t)(base + off + lane * 16 + 12);
    out[lane] = v;
    // where did lane 0's first word (0x1000) land, if anywhere?
    if (lane == 0) {
        unsigned where = 0xffffffffu;
        for (unsigned i = 0; i < lds_bytes; i += 4)
            if (*(lds_u32 *)(size_t)(base + i) == 0x1000u) { where = i; break; }
        *found = where;
    }
}

int main()
{
    const unsigned lds_bytes = 160 * 1024;
    std::vector<uint4> h(64);
    for (unsigned i = 0; i < 64; i++) h[i] = make_uint4(0x1000 + i, 0x2000 + i, 0x3000 + i, 0x4000 + i);
    uint4 *src, *out; unsigned *found;
    CK(hipMalloc(&src, 1024)); CK(hipMalloc(&out, 1024)); CK(hipMalloc(&found, 4));
    CK(hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    for (unsigned off : {0u, 32768u, 64512u, 65536u, 66560u, 98304u, 131072u, 162816u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), lds_bytes, 0, src, off, lds_bytes, out, found);
        CK(hipDeviceSynchronize());
        std::vector<uint4> r(64); unsigned w = 0;
        CK(hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&w, found, 4, hipMemcpyDeviceToHost));
        int ok = 0;
        for (unsigned i = 0; i < 64; i++) ok += (r[i].x == h[i].x && r[i].y == h[i].y && r[i].z == h[i].z && r[i].w == h[i].w);
        printf("M0 = LDS base + %6u: %2d / 64 lanes landed there; lane 0's word found at byte %d\n", off, ok, (int)w);
    }
    return 0;
}

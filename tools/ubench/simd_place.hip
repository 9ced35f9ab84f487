// Where does the dispatcher put the waves of small workgroups?  128-thread workgroups with a large LDS
// allocation (so that only a few fit a CU), each wave records its HW_ID (SIMD, CU, SE) and XCC_ID and
// then idles long enough for the whole grid to be resident together.
//   hipcc --offload-arch=gfx950 -O2 -o simd_place simd_place.hip && ./simd_place [threads=128] [lds_kb=64] [grid=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <string>

__global__ void place_k(unsigned *out, int spin)
{
    extern __shared__ char lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(32);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw; out[2 * w + 1] = xcc;
        if (spin < 0) lds[threadIdx.x] = 1;
    }
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 128, lds_kb = argc > 2 ? atoi(argv[2]) : 64;
    const int grid = argc > 3 ? atoi(argv[3]) : 512, wpb = threads / 64;
    unsigned *d; std::vector<unsigned> h((size_t)2 * grid * wpb);
    hipMalloc(&d, h.size() * 4);
    hipFuncSetAttribute((const void *)place_k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipLaunchKernelGGL(place_k, dim3(grid), dim3(threads), lds_kb * 1024, 0, d, 20000000 /* 100 MHz ticks: 0.2 s */ / 100);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::string> cu;         // (xcc, se, sh, cu) -> list of "block.wave@simd"
    std::map<std::string, int> pattern;
    for (int b = 0; b < grid; b++)
        for (int w = 0; w < wpb; w++) {
            const unsigned hw = h[2 * (b * wpb + w)], xcc = h[2 * (b * wpb + w) + 1] & 15;
            const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            char t[32]; snprintf(t, sizeof t, "%u", simd);
            cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid] += t;
        }
    for (auto &kv : cu) pattern[kv.second]++;
    printf("threads %d lds %d KB grid %d: %zu CUs used; SIMD ids of the waves on one CU (in block order) -> CUs with that pattern\n",
           threads, lds_kb, grid, cu.size());
    for (auto &kv : pattern) printf("  %-24s %d\n", kv.first.c_str(), kv.second);
    return 0;
}

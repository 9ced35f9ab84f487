// Micro-benchmark: cost of VALU instructions on gfx950, latency (one wave per SIMD, dependent
// chain) and throughput (four waves per SIMD).  Cycles are s_memtime ticks per instruction.
//   hipcc -O3 --offload-arch=gfx950 -w -o issue_latency issue_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define REP 64
#define KERNEL(NAME, N_INSTR, ASM)                                                              \
    __global__ __launch_bounds__(64) void NAME(double *out, unsigned long long *cyc, int iters, \
                                               double a, double b)                              \
    {                                                                                           \
        double x0 = threadIdx.x * 1e-3 + 0.5, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;            \
        float f0 = threadIdx.x + 0.5f, f1 = f0 + 1;                                             \
        int i0 = threadIdx.x, i1 = i0 + 7;                                                      \
        float fa = (float)a, fb = (float)b;                                                     \
        unsigned long long m = 0x5555555555555555ull;                                           \
        asm volatile("s_mov_b64 vcc, %0" ::"s"(m) : "vcc");                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        for (int i = 0; i < iters; i++) {                                                       \
            _Pragma("unroll") for (int r = 0; r < REP; r++) {                                   \
                asm volatile(ASM                                                                \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(f0), "+v"(f1),      \
                               "+v"(i0), "+v"(i1)                                               \
                             : "v"(a), "v"(b), "v"(fa), "v"(fb), "s"(m)                         \
                             : "vcc", "s20", "s21", "s22", "s23", "scc");                     \
            }                                                                                   \
        }                                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                     \
        out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + f0 + f1 + i0 + i1;             \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                        \
    }
// operands: %0-%3 doubles x0..x3, %4 %5 floats, %6 %7 ints, %8 a %9 b (double), %10 %11 (float), %12 sgpr pair
KERNEL(k_fma64_dep, 1, "v_fma_f64 %0, %0, %8, %9")
KERNEL(k_fma64_2, 2, "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9")
KERNEL(k_fma64_4, 4, "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9")
KERNEL(k_add64_dep, 1, "v_add_f64 %0, %0, %8")
KERNEL(k_mul64_dep, 1, "v_mul_f64 %0, %0, %8")
KERNEL(k_mul64_4, 4, "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8")
KERNEL(k_fma32_dep, 1, "v_fma_f32 %4, %4, %10, %11")
KERNEL(k_fma32_2, 2, "v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11")
KERNEL(k_cnd_vcc_dep, 1, "v_cndmask_b32 %4, %4, %5, vcc")
KERNEL(k_cnd_vcc_2, 2, "v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %11, vcc")
KERNEL(k_cnd_sgpr_dep, 1, "v_cndmask_b32_e64 %4, %4, %5, %12")
KERNEL(k_cnd_sgpr_2, 2, "v_cndmask_b32_e64 %4, %4, %10, %12\n v_cndmask_b32_e64 %5, %5, %11, %12")
KERNEL(k_mov_dep, 1, "v_mov_b32 %4, %4")
KERNEL(k_and_dep, 1, "v_and_b32 %6, %6, %7")
KERNEL(k_and_2, 2, "v_and_b32 %6, %6, %7\n v_xor_b32 %7, %7, %4")
KERNEL(k_addu_dep, 1, "v_add_u32 %6, %6, %7")
KERNEL(k_bfi_dep, 1, "v_bfi_b32 %6, %7, %6, %4")
KERNEL(k_lshladd64, 1, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL(k_cvt_f64_i32, 2, "v_cvt_f64_i32 %0, %6\n v_cvt_i32_f64 %6, %0")
KERNEL(k_cvt_f64_f32, 2, "v_cvt_f64_f32 %0, %4\n v_cvt_f32_f64 %4, %0")
KERNEL(k_cvt64_4, 4, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5")
KERNEL(k_fract64, 1, "v_fract_f64 %0, %0")
KERNEL(k_rndne64, 1, "v_rndne_f64 %0, %0")
KERNEL(k_ldexp64, 1, "v_ldexp_f64 %0, %0, 1")
KERNEL(k_rcp64, 1, "v_rcp_f64 %0, %0")
KERNEL(k_cmp64_cnd, 2, "v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc")
KERNEL(k_cmp64_only, 1, "v_cmp_gt_f64 vcc, %0, %1")
KERNEL(k_cmp32_only, 1, "v_cmp_gt_f32 vcc, %4, %5")
KERNEL(k_cmp64_sgpr, 1, "v_cmp_gt_f64 s[20:21], %0, %1")
KERNEL(k_mul_lo, 1, "v_mul_lo_u32 %6, %6, %7")
KERNEL(k_fma64_snop, 2, "v_fma_f64 %0, %0, %8, %9\n s_nop 0")
KERNEL(k_fma64_fma32, 2, "v_fma_f64 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11")
KERNEL(k_pk_fma32, 1, "v_pk_fma_f32 %0, %0, %8, %9")
KERNEL(k_ds_read, 1, "ds_read_b64 %1, %6\n s_waitcnt lgkmcnt(0)")
// mask round trips VALU -> SALU -> VALU, branches, votes
KERNEL(k_cmp_sand_cnd, 3, "v_cmp_gt_f64 s[20:21], %0, %1\n s_and_b64 s[20:21], s[20:21], %12\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]")
KERNEL(k_cmp_cnd_e64, 2, "v_cmp_gt_f64 s[20:21], %0, %1\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]")
KERNEL(k_cmp_cnd_dep, 2, "v_cmp_gt_f32 s[20:21], %4, %5\n v_cndmask_b32_e64 %4, %4, %5, s[20:21]")
KERNEL(k_cmp_vote_nt, 3, "v_cmp_gt_f64 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 1f\n1:")
KERNEL(k_cmp_vccz_nt, 2, "v_cmp_lt_f64 vcc, %0, %0\n s_cbranch_vccnz 1f\n1:")
KERNEL(k_branch_taken, 2, "s_branch 1f\n v_mov_b32 %4, %4\n1: v_mov_b32 %5, %5")
KERNEL(k_readfirstlane, 2, "v_readfirstlane_b32 s20, %6\n v_add_u32 %6, s20, %6")
KERNEL(k_salu_dep, 1, "s_add_u32 s20, s20, 1")
KERNEL(k_salu_2, 2, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1")
KERNEL(k_fma64_salu2, 2, "v_fma_f64 %0, %0, %8, %9\n s_add_u32 s20, s20, 1")
KERNEL(k_fma64_3salu, 4, "v_fma_f64 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1")
KERNEL(k_saveexec, 3, "v_cmp_gt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]")
KERNEL(k_ds_write_read, 2, "ds_write_b64 %6, %1\n ds_read_b64 %1, %6\n s_waitcnt lgkmcnt(0)")
KERNEL(k_ds_read6, 6, "ds_read_b64 %0, %6\n ds_read_b64 %1, %6 offset:512\n ds_read_b64 %2, %6 offset:1024\n ds_read_b64 %3, %6 offset:1536\n ds_read_b64 %0, %6 offset:2048\n ds_read_b64 %1, %6 offset:2560\n s_waitcnt lgkmcnt(0)")
KERNEL(k_waitcnt_only, 1, "s_waitcnt lgkmcnt(0)")
KERNEL(k_snop, 1, "s_nop 0")
KERNEL(k_cvt_i32_dep, 2, "v_cvt_i32_f64 %6, %0\n v_and_b32 %6, %6, %7")

// round 3: what makes v_cndmask slow (19 ticks with vcc above)
KERNEL(k_cnd_vcc_e64, 1, "v_cndmask_b32_e64 %4, %4, %5, vcc")
KERNEL(k_cmp_4cnd_vcc, 5, "v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %11, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %6, vcc")
KERNEL(k_cmp_4cnd_sgpr, 5, "v_cmp_gt_f64 s[20:21], %0, %1\n v_cndmask_b32_e64 %4, %4, %10, s[20:21]\n v_cndmask_b32_e64 %5, %5, %11, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n v_cndmask_b32_e64 %7, %7, %6, s[20:21]")
KERNEL(k_cmp_1cnd_3fma, 5, "v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %10, vcc\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9")
KERNEL(k_2cnd_vcc_2fma, 4, "v_cndmask_b32 %4, %4, %10, vcc\n v_fma_f64 %1, %1, %8, %9\n v_cndmask_b32 %5, %5, %11, vcc\n v_fma_f64 %2, %2, %8, %9")
KERNEL(k_bfi_2, 2, "v_bfi_b32 %6, %7, %6, %4\n v_bfi_b32 %5, %7, %5, %4")
KERNEL(k_minmax64, 2, "v_min_f64 %0, %0, %8\n v_max_f64 %1, %1, %9")
KERNEL(k_maxu_ashr, 2, "v_max_u32 %6, %6, %7\n v_ashrrev_i32 %7, 31, %7")
KERNEL(k_addco, 2, "v_add_co_u32 %6, vcc, %6, %7\n v_addc_co_u32 %7, vcc, %7, %6, vcc")
KERNEL(k_and_or, 2, "v_and_or_b32 %6, %6, %7, %4\n v_lshl_add_u32 %7, %7, 9, %5")
KERNEL(k_cvt_f32_f64_2, 2, "v_cvt_f32_f64 %4, %0\n v_cvt_f32_f64 %5, %1")
KERNEL(k_ds_write64, 1, "ds_write_b64 %6, %0")
KERNEL(k_ds_read8_nowait, 8, "ds_read_b64 %0, %6\n ds_read_b64 %1, %6 offset:512\n ds_read_b64 %2, %6 offset:1024\n ds_read_b64 %3, %6 offset:1536\n ds_read_b64 %0, %6 offset:2048\n ds_read_b64 %1, %6 offset:2560\n ds_read_b64 %2, %6 offset:3072\n ds_read_b64 %3, %6 offset:3584")
KERNEL(k_vote_sgpr, 3, "v_cmp_gt_f64 s[20:21], %0, %1\n s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc1 1f\n1:")

typedef void (*kern_t)(double *, unsigned long long *, int, double, double);
struct Case { const char *name; kern_t k; int n; };

int main()
{
    std::vector<Case> cases = {
#define C(K, N) {#K, K, N}
        C(k_fma64_dep, 1), C(k_fma64_2, 2), C(k_fma64_4, 4), C(k_add64_dep, 1), C(k_mul64_dep, 1), C(k_mul64_4, 4),
        C(k_fma32_dep, 1), C(k_fma32_2, 2), C(k_cnd_vcc_dep, 1), C(k_cnd_vcc_2, 2), C(k_cnd_sgpr_dep, 1),
        C(k_cnd_sgpr_2, 2), C(k_mov_dep, 1), C(k_and_dep, 1), C(k_and_2, 2), C(k_addu_dep, 1), C(k_bfi_dep, 1),
        C(k_lshladd64, 1), C(k_cvt_f64_i32, 2), C(k_cvt_f64_f32, 2), C(k_cvt64_4, 4), C(k_fract64, 1), C(k_rndne64, 1),
        C(k_ldexp64, 1), C(k_rcp64, 1), C(k_cmp64_cnd, 2), C(k_cmp64_only, 1), C(k_cmp32_only, 1),
        C(k_cmp64_sgpr, 1), C(k_mul_lo, 1), C(k_fma64_snop, 2), C(k_fma64_fma32, 2),
        C(k_pk_fma32, 1), C(k_ds_read, 1), C(k_cmp_sand_cnd, 3), C(k_cmp_cnd_e64, 2), C(k_cmp_cnd_dep, 2),
        C(k_cmp_vote_nt, 3), C(k_cmp_vccz_nt, 2), C(k_branch_taken, 2), C(k_readfirstlane, 2), C(k_salu_dep, 1),
        C(k_salu_2, 2), C(k_fma64_salu2, 2), C(k_fma64_3salu, 4), C(k_saveexec, 3), C(k_ds_write_read, 2),
        C(k_ds_read6, 6), C(k_waitcnt_only, 1), C(k_snop, 1), C(k_cvt_i32_dep, 2),
        C(k_cnd_vcc_e64, 1), C(k_cmp_4cnd_vcc, 5), C(k_cmp_4cnd_sgpr, 5), C(k_cmp_1cnd_3fma, 5), C(k_2cnd_vcc_2fma, 4),
        C(k_bfi_2, 2), C(k_minmax64, 2), C(k_maxu_ashr, 2), C(k_addco, 2), C(k_and_or, 2), C(k_cvt_f32_f64_2, 2),
        C(k_ds_write64, 1), C(k_ds_read8_nowait, 8), C(k_vote_sgpr, 3)};
    const int iters = 1000;
    const int nbs[3] = {1024, 2048, 4096};        // 1, 2, 4 waves per SIMD on 256 CUs
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&cyc, 4096 * 8);
    unsigned long long *h = (unsigned long long *)malloc(4096 * 8);
    printf("%-18s %28s %28s %28s\n", "", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    printf("%-18s %28s %28s %28s\n", "kernel", "cyc/instr (wave | SIMD)", "cyc/instr (wave | SIMD)", "cyc/instr (wave | SIMD)");
    for (auto &c : cases) {
        printf("%-18s", c.name);
        for (int w = 0; w < 3; w++) {
            int nb = nbs[w];
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(c.k, dim3(nb), dim3(64), 0, 0, out, cyc, iters, 0.999, 1e-3);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < nb; i++) s += (double)h[i];
            s /= nb;
            double per = s / ((double)iters * REP * c.n);
            printf("        %8.2f | %8.2f  ", per, per / (1 << w));
        }
        printf("\n");
    }
    return 0;
}

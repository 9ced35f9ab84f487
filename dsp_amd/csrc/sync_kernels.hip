// sync_kernels.hip -- the receiver synchronisation recurrences on gfx950.
//
// One workgroup owns one tile of 64 bursts, one burst per lane, and walks the Gardner timing
// loop (gardner2.py:63-98 or hw1/gardner.py:45-72) and the decision-directed FLL
// (fll_func.py:14-26) symbol by symbol with all loop state in VGPRs (float64).
//   sync_split_kernel  two waves per tile: wave 0 = timing loop, wave 1 = FLL + slicer, coupled
//                      by a double-buffered LDS queue (the default when both loops run);
//   sync_kernel        one wave does everything (single loops, Lagrange interpolator, FLL only).
//
// Memory system design (CDNA4):
//   * the tile's input is ONE contiguous stream of 64-lane rows; the timing wave streams it
//     into a private LDS ring with LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction,
//     up to four pieces per M0 setup, no VGPR staging) a block of symbols ahead of use;
//   * each lane then picks the rows its own fractional delay needs out of the ring with
//     ds_read (bank-conflict free: lane l always reads column l);
//   * outputs leave as full-row stores (64 lanes x 1/4/8/16 B contiguous).
//   The DMA and the stores are issued through inline asm so that their count in the in-order
//   vmcnt queue is exact: once per block the wave waits with s_waitcnt vmcnt(N), N = the
//   stores issued since the previous block's DMA, which retires that DMA while leaving the
//   stores in flight.
// Control flow per symbol (wave-uniform votes pick the path, every input stays correct):
//   steady   all lanes active, window clear of the burst ends: one vote, no commit selects;
//   fast     the same arithmetic committed per lane by select;
//   generic  per-lane code: mu wrap loops (symbol slips), rows outside the ring from global memory.
// This file is compiled nine times (Makefile): part 0 holds every kernel with the reference's linear
// interpolator, the one-wave kernels and the launch logic; parts 1-6 (-DDSP_SYNC_PART=n) only the
// two-wave kernel with the Lagrange interpolator inside the timing loop (gardner2 / gardner_ted, half_len
// 1, 2, 4 / 3, 6, 8 / 5, 7), part 7 the two-wave kernel without its FLL (the timing loop on its own), part 8
// the fused kernel with a third live wave that feeds the ring, so that the sets of instantiations build
// side by side.
#ifndef DSP_SYNC_PART
#define DSP_SYNC_PART 0
#endif
// dsp_math.h's bit select as one v_bfi_b32 (instead of the compiler's v_and + v_and_or / v_cndmask): measured per
// part on one box (tools/ab_libs.sh, library built with -DDSP_NO_BFI against the product).  Part 0: fll_func alone
// 0.68 -> 0.60 ms at 16 384 bursts, the fused two-wave kernel at 65 536 bursts 1.847 -> 1.840; part 7 (the timing
// loop alone) 0.72 -> 0.69; part 8 (three live waves, <= 1 tile per CU) 0.749 -> 0.761 at 4096 bursts, 0.764 ->
// 0.774 at 16 384: slower, so that part keeps the compiler's selects, as do the Lagrange parts (not measured).
#if DSP_SYNC_PART != 0 && DSP_SYNC_PART != 7 && !defined(DSP_NO_BFI)
#define DSP_NO_BFI
#endif
#if DSP_SYNC_PART != 0 && DSP_SYNC_PART != 7 && !defined(DSP_NO_BFI_T)
#define DSP_NO_BFI_T
#endif
// ... except the FLL wave of part 8 on complex64 tiles, where it does pay (4096 bursts 0.746 -> 0.724 ms; complex128
// tiles 0.760 -> 0.779 with it, and the timing wave's pick3 0.746 -> 0.757): fll_spec_block's FSEL
#if DSP_SYNC_PART == 8 && !defined(DSP_P8_NO_BFI_F)
#define DSP_P8_BFI_F 1
#else
#define DSP_P8_BFI_F 0
#endif

#include "dsp_internal.h"
#include "sync_lane.h"


#pragma clang fp contract(off)

using dspm::cd;

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));
typedef double d2v __attribute__((ext_vector_type(2)));

// ---- asm VMEM helpers (exactly one vector-memory instruction each) ---------
// Cache-policy bits for A/B builds: -DDSP_DMA_NT appends `nt` to every LDS-DMA load, -DDSP_ST_NT to the
// hot paths' stores through wave-uniform bases (the product builds neither: no gain measured)
#ifdef DSP_DMA_NT
#define DSP_DMA_MOD " nt"
#else
#define DSP_DMA_MOD ""
#endif
#ifdef DSP_ST_NT
#define DSP_ST_MOD " nt"
#else
#define DSP_ST_MOD ""
#endif
__device__ __forceinline__ void lds_dma_16B(const void *g, unsigned lds_off)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" DSP_DMA_MOD
                 :: "v"(g), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma_pair(const void *g, unsigned lds_off)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" DSP_DMA_MOD "\n\tglobal_load_lds_dwordx4 %0, off offset:1024" DSP_DMA_MOD
                 :: "v"(g), "s"(lds_off) : "memory", "m0");
}
// n (1..4) consecutive 1 KiB pieces with one M0 / one address: the instruction offset moves
// the global and the LDS address together
__device__ __forceinline__ void lds_dma_group(const void *g, unsigned lds_off, int n)
{
#define DMA0 "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" DSP_DMA_MOD
#define DMAN(o) "\n\tglobal_load_lds_dwordx4 %0, off offset:" #o DSP_DMA_MOD
    switch (n) {
        case 1: asm volatile(DMA0 :: "v"(g), "s"(lds_off) : "memory", "m0"); break;
        case 2: asm volatile(DMA0 DMAN(1024) :: "v"(g), "s"(lds_off) : "memory", "m0"); break;
        case 3: asm volatile(DMA0 DMAN(1024) DMAN(2048) :: "v"(g), "s"(lds_off) : "memory", "m0"); break;
        default: asm volatile(DMA0 DMAN(1024) DMAN(2048) DMAN(3072) :: "v"(g), "s"(lds_off) : "memory", "m0"); break;
    }
#undef DMA0
#undef DMAN
}
__device__ __forceinline__ void st_b8(uint8_t *p, unsigned v)
{
    asm volatile("global_store_byte %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
// `keep` false -> the lane stores zeros (rows past a burst's last symbol read as zero)
__device__ __forceinline__ void st_val(float *p, double v, bool keep = true)
{
    float f = (float)v;
    f = keep ? f : 0.0f;
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(f) : "memory");
}
__device__ __forceinline__ void st_val(double *p, double v, bool keep = true)
{
    v = keep ? v : 0.0;
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_cplx(f2v *p, double re, double im, bool keep = true)
{
    f2v v = {(float)re, (float)im};
    v.x = keep ? v.x : 0.0f;
    v.y = keep ? v.y : 0.0f;
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
// A store of more than 64 bits reads its data VGPRs for a few cycles after it issues: a VALU write to
// them needs 2 wait states after the store on gfx940+ (1 before).  The compiler's hazard recogniser
// pads its own stores but cannot see inside inline asm, so the pad is part of the asm text here
// (without it the low dword of the real part came out overwritten in lanes 12-15 of every 16).
#define DSP_WIDE_STORE_PAD "\n\ts_nop 1"
__device__ __forceinline__ void st_cplx(d2v *p, double re, double im, bool keep = true)
{
    d2v v = {keep ? re : 0.0, keep ? im : 0.0};
    asm volatile("global_store_dwordx4 %0, %1, off" DSP_WIDE_STORE_PAD :: "v"(p), "v"(v) : "memory");
}

// stores through a wave-uniform base (SGPR pair) + 32-bit lane offset: no 64-bit VALU address math.
// HAZARD (the cause of round 2's two GPU-box faults): a VMEM instruction must not read an SGPR within
// 5 wait states of a VALU write to it.  The compiler keeps that distance for its own instructions but
// cannot see inside asm text, and under register pressure it restores a spilled base with
// `v_readlane_b32 s1, v127, 37` DIRECTLY in front of the asm: the store then goes through a
// half-restored 64-bit address.  The guard is therefore part of the asm text: five wait states in front
// of the store (DSP_SST_GUARD 2, the default), or -- 1 -- a SALU copy of the base into a scalar temporary
// of the asm's own (SALU reads of VALU-written SGPRs are interlocked, a VMEM read of a SALU-written SGPR
// needs no wait).  Measured at B = 65536 / 32768 with one guard per store: copy +0.5 % / +1.9 %, wait
// states +1.0 % / +4.2 % (profiles/r03_sst_guard_ab.txt).  The copy is cheaper, but the compiler may hand
// the asm a temporary that a VALU instruction wrote a few instructions earlier (a dead v_readlane target):
// harmless as far as the ISA text goes, yet outside the rule LLVM itself applies (ANY VALU write of the
// register inside the window counts), which is the rule tools/isa_lint.py enforces on every build
// flavour.  The hot paths group their stores behind ONE wait (GroupStore below), so the default is the
// form that needs no argument.
#ifndef DSP_SST_GUARD
#define DSP_SST_GUARD 2
#endif
#if DSP_SST_GUARD == 1
#define DSP_SST(OPC, TAIL)                                                                          \
    unsigned long long tb;                                                                          \
    asm volatile("s_mov_b64 %0, %3\n\t" OPC " %1, %2, %0" DSP_ST_MOD TAIL : "=&s"(tb) : "v"(voff), "v"(v), "s"(base) : "memory")
#elif DSP_SST_GUARD == 2
#define DSP_SST(OPC, TAIL)                                                                          \
    asm volatile("s_nop 4\n\t" OPC " %0, %1, %2" DSP_ST_MOD TAIL :: "v"(voff), "v"(v), "s"(base) : "memory")
#else   // 0: no guard -- round 2's text, for A/B timing only; the lint decides whether such a build may exist
#define DSP_SST(OPC, TAIL)                                                                          \
    asm volatile(OPC " %0, %1, %2" DSP_ST_MOD TAIL :: "v"(voff), "v"(v), "s"(base) : "memory")
#endif
__device__ __forceinline__ void sst(const void *base, unsigned voff, float v) { DSP_SST("global_store_dword", ""); }
__device__ __forceinline__ void sst(const void *base, unsigned voff, double v) { DSP_SST("global_store_dwordx2", ""); }
__device__ __forceinline__ void sst(const void *base, unsigned voff, f2v v) { DSP_SST("global_store_dwordx2", ""); }
__device__ __forceinline__ void sst(const void *base, unsigned voff, d2v v) { DSP_SST("global_store_dwordx4", DSP_WIDE_STORE_PAD); }
__device__ __forceinline__ void sst_b8(const void *base, unsigned voff, unsigned v) { DSP_SST("global_store_byte", ""); }
#undef DSP_SST

// Several stores through wave-uniform bases in ONE asm statement: every base is an input of the
// statement, so whatever restores the compiler needs sit in front of it, and the five wait states
// are spent once for the group.  (One statement per store with its own guard costs the timing wave
// 2 % at B <= 32768: four extra instructions per symbol on a lone wave.)
// Operand order: [voff_c, cplx, base_c,] voff_s, s0, base0 [, s1, base1 [, s2, base2]]; dec: voff_b, byte, base.
template <typename OC> struct GroupStore;
#define DSP_GS_IMPL(OC_T, OS_T, OCOP, OSOP, PAD)                                                                  \
    template <> struct GroupStore<OC_T> {                                                                         \
        static __device__ __forceinline__ void c_s_s(unsigned oc, OC_T c, const void *bc, unsigned os, OS_T a,   \
                                                     const void *ba, OS_T b, const void *bb)                    \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OCOP " %0, %1, %2" DSP_ST_MOD "\n\t" OSOP " %3, %4, %5" DSP_ST_MOD "\n\t" OSOP " %3, %6, %7" DSP_ST_MOD PAD        \
                         :: "v"(oc), "v"(c), "s"(bc), "v"(os), "v"(a), "s"(ba), "v"(b), "s"(bb) : "memory");      \
        }                                                                                                         \
        static __device__ __forceinline__ void c_s_s_s(unsigned oc, OC_T c, const void *bc, unsigned os, OS_T a, \
                                                       const void *ba, OS_T b, const void *bb, OS_T d,          \
                                                       const void *bd)                                           \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OCOP " %0, %1, %2" DSP_ST_MOD "\n\t" OSOP " %3, %4, %5" DSP_ST_MOD "\n\t" OSOP " %3, %6, %7" DSP_ST_MOD "\n\t"        \
                         OSOP " %3, %8, %9" DSP_ST_MOD PAD                                                                    \
                         :: "v"(oc), "v"(c), "s"(bc), "v"(os), "v"(a), "s"(ba), "v"(b), "s"(bb), "v"(d), "s"(bd)  \
                         : "memory");                                                                             \
        }                                                                                                         \
        static __device__ __forceinline__ void s_s(unsigned os, OS_T a, const void *ba, OS_T b, const void *bb)  \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OSOP " %0, %1, %2" DSP_ST_MOD "\n\t" OSOP " %0, %3, %4" DSP_ST_MOD                                   \
                         :: "v"(os), "v"(a), "s"(ba), "v"(b), "s"(bb) : "memory");                                \
        }                                                                                                         \
        static __device__ __forceinline__ void s(unsigned os, OS_T a, const void *ba)                            \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OSOP " %0, %1, %2" DSP_ST_MOD :: "v"(os), "v"(a), "s"(ba) : "memory");                \
        }                                                                                                         \
        static __device__ __forceinline__ void c_s_s_b(unsigned oc, OC_T c, const void *bc, unsigned os, OS_T a, \
                                                       const void *ba, OS_T b, const void *bb, unsigned ob,     \
                                                       unsigned q, const void *bq)                               \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OCOP " %0, %1, %2" DSP_ST_MOD "\n\t" OSOP " %3, %4, %5" DSP_ST_MOD "\n\t" OSOP " %3, %6, %7" DSP_ST_MOD "\n\t"        \
                         "global_store_byte %8, %9, %10" DSP_ST_MOD PAD                                                       \
                         :: "v"(oc), "v"(c), "s"(bc), "v"(os), "v"(a), "s"(ba), "v"(b), "s"(bb), "v"(ob), "v"(q), \
                            "s"(bq) : "memory");                                                                  \
        }                                                                                                         \
        static __device__ __forceinline__ void c_s_b(unsigned oc, OC_T c, const void *bc, unsigned os, OS_T a,   \
                                                     const void *ba, unsigned ob, unsigned q, const void *bq)    \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OCOP " %0, %1, %2" DSP_ST_MOD "\n\t" OSOP " %3, %4, %5" DSP_ST_MOD "\n\t"                                   \
                         "global_store_byte %6, %7, %8" DSP_ST_MOD PAD                                                        \
                         :: "v"(oc), "v"(c), "s"(bc), "v"(os), "v"(a), "s"(ba), "v"(ob), "v"(q), "s"(bq) : "memory");  \
        }                                                                                                         \
        static __device__ __forceinline__ void c_b(unsigned oc, OC_T c, const void *bc, unsigned ob, unsigned q, \
                                                   const void *bq)                                               \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OCOP " %0, %1, %2" DSP_ST_MOD "\n\tglobal_store_byte %3, %4, %5" DSP_ST_MOD PAD               \
                         :: "v"(oc), "v"(c), "s"(bc), "v"(ob), "v"(q), "s"(bq) : "memory");                       \
        }                                                                                                         \
        static __device__ __forceinline__ void s_b(unsigned os, OS_T a, const void *ba, unsigned ob, unsigned q, \
                                                   const void *bq)                                               \
        {                                                                                                         \
            asm volatile("s_nop 4\n\t" OSOP " %0, %1, %2" DSP_ST_MOD "\n\tglobal_store_byte %3, %4, %5" DSP_ST_MOD                         \
                         :: "v"(os), "v"(a), "s"(ba), "v"(ob), "v"(q), "s"(bq) : "memory");                       \
        }                                                                                                         \
        static __device__ __forceinline__ void b(unsigned ob, unsigned q, const void *bq)                        \
        {                                                                                                         \
            asm volatile("s_nop 4\n\tglobal_store_byte %0, %1, %2" DSP_ST_MOD :: "v"(ob), "v"(q), "s"(bq) : "memory");       \
        }                                                                                                         \
    }
DSP_GS_IMPL(f2v, float, "global_store_dwordx2", "global_store_dword", "");
DSP_GS_IMPL(d2v, double, "global_store_dwordx4", "global_store_dwordx2", DSP_WIDE_STORE_PAD);
#undef DSP_GS_IMPL

template <typename T>
__device__ __forceinline__ T lds_ld(unsigned byte_addr)
{
    // (through size_t: the host pass of the compiler takes every pointer for 64 bits wide and warns otherwise)
    return *(const __attribute__((address_space(3))) T *)(size_t)byte_addr;
}
template <typename T>
__device__ __forceinline__ void lds_st(unsigned byte_addr, T v)
{
    *(__attribute__((address_space(3))) T *)(size_t)byte_addr = v;
}

// s_waitcnt takes its count as an immediate; the count here is a per-launch constant (stores issued per
// DMA block).  A switch over 64 cases compiles to a tree of ~25 scalar branches that a lone wave walks at
// ~20 cycles each, once per DMA block (measured: 200 cycles per SYMBOL went into advance()).  The host
// therefore rounds the count DOWN to a multiple of four above 8 (waiting for a few more of the block's
// oldest stores: stricter, never laxer) and the kernel decides in at most five uniform branches.
#define WV(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void wait_vmcnt(int n)
{
    if (n >= 32) {
        if (n >= 48) { if (n >= 56) { if (n >= 60) WV(60); else WV(56); } else { if (n >= 52) WV(52); else WV(48); } }
        else { if (n >= 40) { if (n >= 44) WV(44); else WV(40); } else { if (n >= 36) WV(36); else WV(32); } }
    } else if (n >= 16) {
        if (n >= 24) { if (n >= 28) WV(28); else WV(24); } else { if (n >= 20) WV(20); else WV(16); }
    } else if (n >= 8) {
        if (n >= 12) WV(12); else WV(8);
    } else {
        if (n >= 4) { if (n >= 6) { if (n >= 7) WV(7); else WV(6); } else { if (n >= 5) WV(5); else WV(4); } }
        else { if (n >= 2) { if (n >= 3) WV(3); else WV(2); } else { if (n >= 1) WV(1); else WV(0); } }
    }
}
#undef WV

// transcendental constants: loaded once per lane through an opaque index so that they
// stay in VGPRs (see dsp_math.h)
__device__ const double k_math_table[dspm::K_COUNT] = DSP_MATH_TABLE_INIT;

__device__ __forceinline__ dspm::MathK load_math_consts()
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    dspm::MathK k;
#pragma unroll
    for (int i = 0; i < dspm::K_COUNT; i++) k.v[i] = k_math_table[z + i];
    // An empty asm is a use: the compiler's s_waitcnt for these loads lands HERE.  Left to the first real use
    // it can land inside a block loop (the gardner_ted kernels have no FLL step in front of theirs), and as the
    // asm stores are invisible to the compiler's count it comes out as vmcnt(0) there: every symbol then waits
    // for all of the wave's stores (16 384 gardner_ted bursts, all outputs: 0.99 ms instead of 0.78).
#pragma unroll
    for (int i = 0; i < dspm::K_COUNT; i++) asm volatile("" : "+v"(k.v[i]));
    return k;
}

template <typename IN> struct InT;
template <> struct InT<f2v> { typedef float S; typedef float OS; typedef f2v OC; enum { RPP = 2 }; };
template <> struct InT<d2v> { typedef double S; typedef double OS; typedef d2v OC; enum { RPP = 1 }; };

// ---------------------------------------------------------------------------
// One wave's view of its tile: the input stream, the LDS ring and its window.
template <typename IN>
struct Tile {
    const IN *xt;                 // this tile's rows in global memory
    const unsigned char *ring;    // LDS ring base (generic pointer)
    unsigned lds_base;            // LDS byte address of the ring
    int lane, RM, R, in_rows;
    int col;                      // the column this lane reads (its own; a passenger lane reads another burst's)
    int guard;                    // rows behind the ring that mirror its first rows (0 or 8)
    int win_lo, win_hi;           // rows [win_lo, win_hi) are valid in the ring
    int F;                        // DMA front: rows [0, F) have been issued
    double scale;

    // generic access: ring if the row is inside the window, else global memory
    __device__ __forceinline__ cd fetch(int row) const
    {
        IN v;
        if ((unsigned)(row - win_lo) < (unsigned)(win_hi - win_lo)) {
            v = ((const IN *)ring)[(row & RM) * 64 + col];
        } else {
            int r = row < 0 ? 0 : (row >= in_rows ? in_rows - 1 : row);
            v = xt[(size_t)r * 64 + col];
            asm volatile("" : "+v"(v));          // keep the load's wait inside this branch
        }
        cd z;
        z.re = (double)v.x * scale;
        z.im = (double)v.y * scale;
        return z;
    }
    // straight-line access, row known to be inside the window; SC: apply the per-burst scale
    template <bool SC>
    __device__ __forceinline__ cd lds_at(int row) const
    {
        IN v = ((const IN *)ring)[(row & RM) * 64 + col];
        cd z;
        z.re = SC ? (double)v.x * scale : (double)v.x;
        z.im = SC ? (double)v.y * scale : (double)v.y;
        return z;
    }
    // issue the DMA pieces for rows [F, F_new).  Blocks are whole numbers of 1 KiB pieces and the ring is a
    // power of two of them, so PAIRS of pieces that start on an even piece never cross the ring's end: two
    // pieces per M0 / address, one uniform loop without a switch (the four-piece groups of rounds 1-2 cost a
    // four-way branch per group and the same number of VMEM instructions).
    __device__ __forceinline__ void issue(int F_new)
    {
        constexpr int RPP = InT<IN>::RPP;
        F_new = F_new < in_rows ? F_new : in_rows;
        int r = F;
        const unsigned char *gp = (const unsigned char *)(xt + (size_t)r * 64) + lane * 16;
        if (r < F_new && ((r / RPP) & 1)) {              // an odd first piece goes alone
            lds_dma_16B(gp, lds_base + (unsigned)((r & RM) * 64 * (int)sizeof(IN)));
            r += RPP; gp += 1024;
        }
        for (; r + 2 * RPP <= F_new; r += 2 * RPP, gp += 2048)
            lds_dma_pair(gp, lds_base + (unsigned)((r & RM) * 64 * (int)sizeof(IN)));
        for (; r < F_new; r += RPP, gp += 1024)          // (a last piece, whole or -- at the tile's end -- partial)
            lds_dma_16B(gp, lds_base + (unsigned)((r & RM) * 64 * (int)sizeof(IN)));
        F = F_new > F ? F_new : F;
    }
    // start of DMA block `blk` (T iterations of `step` rows each)
    // Guard rows: ring slots [0, guard) copied to slots [R, R + guard) once the DMA that filled them has
    // landed, every lane its own column.  A lane's rows of one symbol (at most guard - 1 rows apart) then
    // lie at consecutive LDS addresses even when they wrap: one address and immediate offsets instead of a
    // masked address per row.  [lo, hi) = the rows that have just landed.
    __device__ __forceinline__ void mirror(int lo, int hi)
    {
        if (!guard || hi <= lo) return;
        const int s0 = lo & RM;
        if (s0 >= guard && s0 + (hi - lo) <= R) return;          // the range touches no slot below `guard`
        typedef __attribute__((address_space(3))) IN lds_in;
        lds_in *base = (lds_in *)(size_t)lds_base + lane;
        IN v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = base[j * 64];
#pragma unroll
        for (int j = 0; j < 8; j++) base[(R + j) * 64] = v[j];
    }
    __device__ __forceinline__ void advance(int blk, int step, int T, int C, int nwait)
    {
        if (blk > 0) wait_vmcnt(nwait);     // the previous block's DMA has landed
        mirror(win_hi, F);
        win_hi = F;
        issue(step * (blk + 2) * T + C);
        win_lo = F - R > 0 ? F - R : 0;
    }
    // the same bookkeeping when another wave issues and awaits the DMA (SyncParams::dma_fll)
    __device__ __forceinline__ void advance_nodma(int blk, int step, int T, int C)
    {
        win_hi = F;
        int F_new = step * (blk + 2) * T + C;
        F_new = F_new < in_rows ? F_new : in_rows;
        F = F_new > F ? F_new : F;
        win_lo = F - R > 0 ? F - R : 0;
    }
    __device__ __forceinline__ void prologue_nodma(int step, int T, int C)
    {
        const int F_new = step * T + C;
        F = F_new < in_rows ? F_new : in_rows;
        win_hi = F;
    }
    __device__ __forceinline__ void prologue(int step, int T, int C)
    {
        issue(step * T + C);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mirror(0, F);
        win_hi = F;
    }
};

// row 0 of a tile's windows: tiles lie x_tile_rows rows apart (== in_rows for separate tiles; fewer when
// consecutive tiles' windows overlap in one buffer, segment-parallel mode), tile 0 may live elsewhere
template <typename IN>
__device__ __forceinline__ const IN *tile_rows(const SyncParams &P, int tile)
{
    if (tile == 0 && P.x_tile0) return (const IN *)P.x_tile0;
    return (const IN *)P.x + (ptrdiff_t)tile * P.x_tile_rows * 64;
}

template <typename IN>
struct FetchFn {
    const Tile<IN> &t;
    __device__ __forceinline__ cd operator()(int row) const { return t.fetch(row); }
};

struct TimingConst {
    int len, sps, n_symbols;
    double spsd, lend, halfd, alpha;
};

// One pass of the timing loop body for every lane of the wave.  Fast path: no
// lane needs the wrap loops and all rows are inside the ring window -> straight
// line code, results committed by select.  Otherwise the generic per-lane code.
template <typename IN, int VARIANT, bool LAGR, bool SC>
__device__ __forceinline__ dspm::GOut timing_iter(const Tile<IN> &tl, dspm::GState &gs, const TimingConst &tc,
                                                  const dspm::LagrangeW *Lp, dspm::TrAcc *tr = nullptr)
{
    dspm::GOut go; go.produced = false; go.rec.re = 0.0; go.rec.im = 0.0; go.err = 0.0; go.mu = 0.0;
    const int len = tc.len, sps = tc.sps;
    const double spsd = tc.spsd, lend = tc.lend, alpha = tc.alpha;
    const int win_lo = tl.win_lo, win_hi = tl.win_hi;
    bool fast = !LAGR;
    bool ok = false;
    double mu = gs.mu, i_a = 0.0, i_b = 0.0, i_c = 0.0;
    int r_a = 0, r_b = 0, r_c = 0;
    if (!LAGR) {
        if (VARIANT == DSP_VARIANT_GARDNER2) {
            // gardner2.py:63-80 without the wrap loops
            const bool cont = gs.active && (gs.k <= len - sps);
            const double kd = (double)gs.k;
            const bool odd = cont && !(mu < spsd && mu >= -spsd);   // wrap needed, or NaN
            i_a = (kd - spsd) + mu;           // idx_early
            i_b = (kd - tc.halfd) + mu;       // idx_mid
            i_c = kd + mu;                    // idx_late
            ok = cont && !(i_a < 0.0) && !(i_c + 1.0 >= lend);
            r_a = __double2int_rz(i_a); r_b = __double2int_rz(i_b); r_c = __double2int_rz(i_c);
            const bool inwin = (r_a >= win_lo) && (r_c + 1 < win_hi);
            fast = !__any(odd || (ok && !inwin));
        } else {
            // hw1/gardner.py:45-53
            const bool cont = gs.active && (gs.k < tc.n_symbols);
            const double base = (double)(gs.k * sps);
            i_a = base + mu;                                   // idx_k
            i_c = (double)((gs.k + 1) * sps) + mu;             // idx_next
            i_b = base + mu + tc.halfd;                        // idx_mid
            ok = cont && !(i_c + 1.0 >= lend || i_b + 1.0 >= lend);
            r_a = __double2int_rd(i_a); r_b = __double2int_rd(i_b); r_c = __double2int_rd(i_c);   // np.floor
            const bool inwin = (r_a >= win_lo) && (r_c + 1 < win_hi) && (r_a >= 0);
            fast = !__any(ok && !inwin);
        }
    }
    DSP_TR(tr, 1);
    if (fast) {
        cd a0 = tl.template lds_at<SC>(r_a), a1 = tl.template lds_at<SC>(r_a + 1);
        cd b0 = tl.template lds_at<SC>(r_b), b1 = tl.template lds_at<SC>(r_b + 1);
        cd c0 = tl.template lds_at<SC>(r_c), c1 = tl.template lds_at<SC>(r_c + 1);
        DSP_TR(tr, 2);
        // the three indices are >= 0 on every lane that commits: idx - trunc(idx) = fract(idx)
        cd xa = dspm::lerp(a0, a1, dspm::fract_pos(i_a));
        cd xb = dspm::lerp(b0, b1, dspm::fract_pos(i_b));
        cd xc = dspm::lerp(c0, c1, dspm::fract_pos(i_c));
        const double e = dspm::ted_error(xa, xb, xc);
        cd rec; rec.re = 0.0; rec.im = 0.0;
        double mu_n, mu_hist;
        if (VARIANT == DSP_VARIANT_GARDNER2) {
            mu_n = mu - alpha * e;                              // :91
            mu_hist = mu_n;
            const double i_o = (double)gs.k + mu_n;             // :95
            const int r_o = __double2int_rz(i_o);
            const int hi_lim = win_hi < len ? win_hi : len;
            const bool ofast = (i_o > -1.0e9 && i_o < 1.0e9) && r_o >= win_lo && r_o + 1 < hi_lim;
            DSP_TR(tr, 3);
            // mu moves by alpha*e per symbol: the output instant almost always lies between the
            // same two samples as the late one, which are already in registers
            const bool same = ofast && r_o == r_c;
            if (!__any(ok && !same)) {
                rec = dspm::lerp(c0, c1, i_o - (double)r_o);
            } else if (!__any(ok && !ofast)) {
                cd o0 = tl.template lds_at<SC>(r_o), o1 = tl.template lds_at<SC>(r_o + 1);
                DSP_TR(tr, 4);
                rec = dspm::lerp(o0, o1, i_o - (double)r_o);
            } else if (ok) {
                rec = dspm::g2_interp(i_o, len, (const dspm::LagrangeW *)nullptr, FetchFn<IN>{tl});
            }
            if (ok) gs.k += sps;                                // :98
        } else {
            mu_hist = mu;                                       // :63
            rec = xa;                                           // :72
            mu_n = mu + alpha * e;                              // :66
            mu_n = mu_n < -spsd ? -spsd : mu_n;                 // :69
            mu_n = mu_n > spsd ? spsd : mu_n;
            if (ok) gs.k += 1;
        }
        gs.mu = ok ? mu_n : mu;
        gs.active = ok;
        go.produced = ok;          // err / mu / rec of a lane that did not produce are never
        go.err = e;                // consumed: the stores zero them by predicate
        go.mu = mu_hist;
        go.rec = rec;
    } else {
        if (VARIANT == DSP_VARIANT_GARDNER2)
            go = dspm::gardner2_iter(gs, len, sps, alpha, Lp, FetchFn<IN>{tl});
        else
            go = dspm::ted_iter(gs, len, sps, tc.n_symbols, alpha, Lp, FetchFn<IN>{tl});
    }
    return go;
}

// Steady-state gardner2.py iteration: every lane of the wave is active, the ring window is
// past the burst start (win_lo > 0, so idx_early >= 1) and ends at least 2*sps + 1 rows before
// the shortest burst's end (so k <= len - sps and idx_late + 1 < len hold for every lane), and
// the caller has checked the output capacity for the whole block.  What remains per lane is
// the wrap test on mu and the window test on the rows; one vote, then straight-line code with
// no per-lane commit selects.  Returns false (state untouched) when a lane fails the vote.
template <typename IN, bool SC>
__device__ __forceinline__ bool g2_steady(const Tile<IN> &tl, dspm::GState &gs, const TimingConst &tc,
                                          cd &rec, double &err, double &pos)
{
    const double mu = gs.mu, spsd = tc.spsd;
    const double kd = (double)gs.k;
    const double i_a = (kd - spsd) + mu;                 // gardner2.py:76-78
    const double i_b = (kd - tc.halfd) + mu;
    const double i_c = kd + mu;
    const int r_a = __double2int_rz(i_a), r_b = __double2int_rz(i_b), r_c = __double2int_rz(i_c);
    const bool in = (mu < spsd) & (mu >= -spsd) & (r_a >= tl.win_lo) & (r_c + 1 < tl.win_hi);
    if (!__all(in)) return false;
    const cd a0 = tl.template lds_at<SC>(r_a), a1 = tl.template lds_at<SC>(r_a + 1);
    const cd b0 = tl.template lds_at<SC>(r_b), b1 = tl.template lds_at<SC>(r_b + 1);
    const cd c0 = tl.template lds_at<SC>(r_c), c1 = tl.template lds_at<SC>(r_c + 1);
    const cd xa = dspm::lerp(a0, a1, dspm::fract_pos(i_a));
    const cd xb = dspm::lerp(b0, b1, dspm::fract_pos(i_b));
    const cd xc = dspm::lerp(c0, c1, dspm::fract_pos(i_c));
    const double e = dspm::ted_error(xa, xb, xc);        // :87
    const double mu_n = mu - tc.alpha * e;               // :91
    const double i_o = kd + mu_n;                        // :95
    const int r_o = __double2int_rz(i_o);
    if (__all(r_o == r_c)) {                             // i_o in [r_c, r_c + 1): samples in registers
        rec = dspm::lerp(c0, c1, dspm::fract_pos(i_o));
    } else if (__all((i_o > 0.0) & (i_o < 1.0e9) & (r_o >= tl.win_lo) & (r_o + 1 < tl.win_hi))) {
        rec = dspm::lerp(tl.template lds_at<SC>(r_o), tl.template lds_at<SC>(r_o + 1), dspm::fract_pos(i_o));
    } else {
        rec = dspm::g2_interp(i_o, tc.len, (const dspm::LagrangeW *)nullptr, FetchFn<IN>{tl});
    }
    err = e;
    pos = i_o;
    gs.mu = mu_n;
    gs.k += tc.sps;                                      // :98
    return true;
}

// ---------------------------------------------------------------------------
// Speculative steady segment (round 3): n consecutive gardner2.py iterations as ONE branch-free
// instruction stream, checked once at its end.
//
// A lone wave issues a dependent instruction every ~8 cycles, an independent one every ~5, and a wave
// vote (v_cmp -> s_cmp -> s_cbranch) stalls it for ~58: g2_steady above pays two votes per symbol and
// its branches cut the symbol into basic blocks the scheduler cannot overlap.  Here nothing inside the
// segment depends on a vote.  Every lane assumes what is true of a locked loop -- mu inside
// [-sps, sps), the three instants in one binade (then idx_mid = idx_early + (sps - sps // 2) and idx_late =
// idx_early + sps hold EXACTLY, so one truncation and one fraction serve all three), the rows inside
// the ring window, |alpha * e| < 1 (then the output instant lies within one sample of the late one) --
// and records the evidence in four accumulators.  One vote at the end of the segment; if any lane's
// assumption failed, the caller restores (mu, k) and runs the same symbols through the per-symbol code,
// whose stores and queue entries overwrite the segment's.  Values are those of g2_steady bit for bit
// (same products and sums in the same order).
//   * the output sample (k + mu_new) is fetched by per-lane address one symbol late: its LDS latency and
//     its conversions fall into the next symbol's chain instead of extending this one;
//   * stores go out as one guarded group per symbol (GroupStore); MASK = the outputs present
//     (1 rec, 2 err, 4 mu, 8 pos), -1 = decided at run time per store.
struct SpecBases { const unsigned char *rec_t, *err_t, *mu_t, *pos_t; };

template <typename IN, bool SC>
__device__ __forceinline__ cd widen(const IN v, const double scale)
{
    cd z;
    z.re = SC ? (double)v.x * scale : (double)v.x;
    z.im = SC ? (double)v.y * scale : (double)v.y;
    return z;
}

// lt ? a : (gt ? b : c), bit by bit through masks (see dsp_math.h on v_cndmask)
__device__ __forceinline__ f2v pick3(unsigned lt, unsigned gt, f2v a, f2v b, f2v c)
{
    f2v r;
    r.x = __uint_as_float(dspm::bsel_t(lt, __float_as_uint(a.x), dspm::bsel_t(gt, __float_as_uint(b.x), __float_as_uint(c.x))));
    r.y = __uint_as_float(dspm::bsel_t(lt, __float_as_uint(a.y), dspm::bsel_t(gt, __float_as_uint(b.y), __float_as_uint(c.y))));
    return r;
}
__device__ __forceinline__ d2v pick3(unsigned lt, unsigned gt, d2v a, d2v b, d2v c)
{
    d2v r;
    r.x = dspm::bsel64_t(lt, a.x, dspm::bsel64_t(gt, b.x, c.x));
    r.y = dspm::bsel64_t(lt, a.y, dspm::bsel64_t(gt, b.y, c.y));
    return r;
}

template <typename IN, bool SC, int MASK, bool GUARD>
__device__ __forceinline__ bool g2_spec_segment(const Tile<IN> &tl, dspm::GState &gs, const TimingConst &tc,
                                                const int n, const int it0, const int roff, const SpecBases &sb,
                                                const unsigned lane_s, const unsigned lane_c, const unsigned q_addr)
{
    typedef typename InT<IN>::OS OS;
    typedef typename InT<IN>::OC OC;
    typedef GroupStore<OC> GS;
    constexpr unsigned ROWB = 64u * (unsigned)sizeof(IN);
    const unsigned rmask = (unsigned)tl.R * ROWB - 1u;
    // (the caller has checked that the ring starts on a multiple of its size: row bits, column bits and
    //  base bits do not overlap, so one v_and_or forms an address)
    const unsigned colb = tl.lds_base + (unsigned)tl.col * (unsigned)sizeof(IN);
    const double alpha = tc.alpha, spsd = tc.spsd, scale = tl.scale;
    const int win_lo = tl.win_lo;
    const int wlim = tl.win_hi - win_lo - (tc.sps + 3);  // rows r_a .. r_a + sps + 2 inside [win_lo, win_hi)
    // idx_mid = k - sps // 2 + mu (:74, integer half) lies sps - sps // 2 rows behind idx_early = k - sps + mu:
    // 2 of 4, but 2 of 3 and 3 of 5 (sps / 2 rows, as this line read until the round-3 soak run, is right for
    // even sps only: tests/test_gpu_sync.py::test_spec_segments_odd_sps)
    const unsigned midb = (unsigned)(tc.sps - tc.sps / 2) * ROWB, lateb = (unsigned)tc.sps * ROWB;
    double mu = gs.mu;
    double kd = (double)gs.k, kd4 = kd - spsd;
    unsigned acc_x = 0u, acc_w = 0u, acc_mu = 0u, acc_e = 0u;
    unsigned row_s = (unsigned)it0 * 64u * (unsigned)sizeof(OS) + lane_s;
    unsigned row_c = (unsigned)(it0 + roff) * 64u * (unsigned)sizeof(OC) + lane_c;
    unsigned qa = q_addr;
    const unsigned mu_lim = (unsigned)__double2hiint(spsd);     // |mu| < sps <=> hi(|mu|) < hi(sps): sps is a small integer
    IN e0, e1, m0, m1, lm, l0, l1, l2;                   // rows early, early+1, mid, mid+1, late-1 .. late+2
    double i_a = 0.0;
    int r_c = 0;
    unsigned ev_mu = 0u;                                 // the evidence of the symbol whose rows are in flight
    unsigned hi_c = 0u, w_cur = 0u;

    auto issue = [&]() {                                 // indices, evidence, the eight row loads
        i_a = kd4 + mu;                                  // gardner2.py:73
        const double i_c = kd + mu;                      // :75
        const int r_a = __double2int_rz(i_a);
        r_c = r_a + tc.sps;
        hi_c = (unsigned)__double2hiint(i_c);
        w_cur = (unsigned)(r_a - win_lo);
        ev_mu = (unsigned)__double2hiint(mu) & 0x7fffffffu;                // NaN counts as too large
        const unsigned rb = (unsigned)r_a * ROWB;
        if (GUARD) {
            // guard rows behind the ring (Tile::mirror): the rows of one symbol are contiguous in LDS, one
            // masked address and immediate offsets (the compiler pairs them into ds_read2st64)
            const unsigned a0 = (rb & rmask) | colb;
            const unsigned am = a0 + midb, al = a0 + lateb;
            e0 = lds_ld<IN>(a0);
            e1 = lds_ld<IN>(a0 + ROWB);
            m0 = lds_ld<IN>(am);
            m1 = lds_ld<IN>(am + ROWB);
            lm = lds_ld<IN>(al - ROWB);
            l0 = lds_ld<IN>(al);
            l1 = lds_ld<IN>(al + ROWB);
            l2 = lds_ld<IN>(al + 2u * ROWB);
        } else {
            e0 = lds_ld<IN>((rb & rmask) | colb);
            e1 = lds_ld<IN>(((rb + ROWB) & rmask) | colb);
            m0 = lds_ld<IN>(((rb + midb) & rmask) | colb);
            m1 = lds_ld<IN>(((rb + midb + ROWB) & rmask) | colb);
            lm = lds_ld<IN>(((rb + lateb - ROWB) & rmask) | colb);
            l0 = lds_ld<IN>(((rb + lateb) & rmask) | colb);
            l1 = lds_ld<IN>(((rb + lateb + ROWB) & rmask) | colb);
            l2 = lds_ld<IN>(((rb + lateb + 2u * ROWB) & rmask) | colb);
        }
    };

    issue();
    // The instants grow from symbol to symbol (k by sps, mu by less than one sample -- acc_e), so the
    // binade and the window have to be looked at for the first and the last symbol only.
    const unsigned hi_first = (unsigned)__double2hiint(i_a);
    acc_w = w_cur;
    unsigned hi_last = 0u, w_last = 0u;
    for (int s = 0; s < n; s++) {
        // ---- TED and loop update (the loop-carried chain) ----
        hi_last = hi_c; w_last = w_cur;
        acc_mu = ev_mu > acc_mu ? ev_mu : acc_mu;
        const double f = dspm::fract_pos(i_a);           // == fract(idx_mid) == fract(idx_late), see above
        const cd xa = dspm::lerp(widen<IN, SC>(e0, scale), widen<IN, SC>(e1, scale), f);
        const cd xb = dspm::lerp(widen<IN, SC>(m0, scale), widen<IN, SC>(m1, scale), f);
        const cd xc = dspm::lerp(widen<IN, SC>(l0, scale), widen<IN, SC>(l1, scale), f);
        const double e = dspm::ted_error(xa, xb, xc);    // :87
        const double ae = alpha * e;
        const double mu_n = mu - ae;                     // :91
        const unsigned ah = (unsigned)__double2hiint(ae) & 0x7fffffffu;
        acc_e = ah > acc_e ? ah : acc_e;
        const double i_o = kd + mu_n;                    // :95
        // |alpha*e| < 1: the output instant lies within one sample of the late one, so its two samples
        // are among the four rows around it -- a per-lane select on the raw rows, no second LDS trip
        const int d = __double2int_rz(i_o) - r_c;
        const unsigned m_lt = (unsigned)(d >> 31), m_gt = (unsigned)((0 - d) >> 31);    // all ones: d < 0 / d > 0
        const IN o0 = pick3(m_lt, m_gt, lm, l1, l0);
        const IN o1 = pick3(m_lt, m_gt, l0, l2, l1);
        const double fo = dspm::fract_pos(i_o);
        mu = mu_n;
        kd += spsd; kd4 += spsd;                         // :98
        // ---- the next symbol's rows fly while this one's outputs are formed ----
        issue();
        const cd rec = dspm::lerp(widen<IN, SC>(o0, scale), widen<IN, SC>(o1, scale), fo);
        const OC rv = {(OS)rec.re, (OS)rec.im};
        if (MASK == 7) GS::c_s_s(row_c, rv, sb.rec_t, row_s, (OS)e, sb.err_t, (OS)mu_n, sb.mu_t);
        else if (MASK == 15) GS::c_s_s_s(row_c, rv, sb.rec_t, row_s, (OS)e, sb.err_t, (OS)mu_n, sb.mu_t, (OS)i_o, sb.pos_t);
        else if (MASK == 12) GS::s_s(row_s, (OS)mu_n, sb.mu_t, (OS)i_o, sb.pos_t);
        else if (MASK == 4) GS::s(row_s, (OS)mu_n, sb.mu_t);
        else if (MASK != 0) {
            if (sb.rec_t) sst(sb.rec_t, row_c, rv);
            if (sb.err_t) sst(sb.err_t, row_s, (OS)e);
            if (sb.mu_t) sst(sb.mu_t, row_s, (OS)mu_n);
            if (sb.pos_t) sst(sb.pos_t, row_s, (OS)i_o);
        }
        const d2v qv = {rec.re, rec.im};
        lds_st<d2v>(qa, qv);
        row_s += 64u * (unsigned)sizeof(OS); row_c += 64u * (unsigned)sizeof(OC); qa += 1024u;
    }
    // (the rows issued for symbol n are read and dropped: ring addresses are always valid)
    acc_x = hi_first ^ hi_last;
    acc_w = w_last > acc_w ? w_last : acc_w;
    const bool bad = (acc_x >> 20) != 0u || acc_w > (unsigned)wlim || acc_mu >= mu_lim || acc_e >= 0x3ff00000u;
    if (__any(bad)) return false;
    gs.mu = mu;
    gs.k += tc.sps * n;
    return true;
}

// The same for the gardner_ted variant (hw1/gardner.py:45-72), even samples-per-symbol counts: the loop has no
// symbol slips (mu is clipped to [-sps, sps] instead) and puts out x_k, the sample at the OLD mu, which the TED
// needs anyway -- no output interpolation.  Assumptions on the evidence words: the clip inactive (|mu| < sps
// before and after every update), idx_k and idx_next in one binade (then idx_mid = idx_k + sps/2 -- a whole
// number of samples for even sps -- and idx_next = idx_k + sps hold exactly: one floor and one fraction serve all
// three; odd sps put the mid instant half a sample off and stay with the per-symbol code), the rows inside the
// ring window, |delta_mu * e| < 1 (so that the instants grow from symbol to symbol and the first and the last
// bound them all).  idx_k >= win_lo > 0, so np.floor is the truncation.  The caller guarantees that every lane
// has at least n more symbols to run (k + n <= (len - 2 sps) // sps; :52's break cannot fire before that for
// sps >= 2).  Same values as ted_iter / timing_iter bit for bit; state untouched when the vote fails.
template <typename IN, bool SC, int MASK, bool GUARD>
__device__ __forceinline__ bool ted_spec_segment(const Tile<IN> &tl, dspm::GState &gs, const TimingConst &tc,
                                                 const int n, const int it0, const SpecBases &sb,
                                                 const unsigned lane_s, const unsigned lane_c, const unsigned q_addr)
{
    typedef typename InT<IN>::OS OS;
    typedef typename InT<IN>::OC OC;
    typedef GroupStore<OC> GS;
    constexpr unsigned ROWB = 64u * (unsigned)sizeof(IN);
    const unsigned rmask = (unsigned)tl.R * ROWB - 1u;
    const unsigned colb = tl.lds_base + (unsigned)tl.col * (unsigned)sizeof(IN);
    const double delta = tc.alpha, spsd = tc.spsd, scale = tl.scale;
    const int win_lo = tl.win_lo;
    const int wlim = tl.win_hi - win_lo - (tc.sps + 2);  // rows r_k .. r_k + sps + 1 inside [win_lo, win_hi)
    const unsigned midb = (unsigned)(tc.sps / 2) * ROWB, lateb = (unsigned)tc.sps * ROWB;
    double mu = gs.mu;
    double base = (double)(gs.k * tc.sps), basen = (double)((gs.k + 1) * tc.sps);     // :47-48, whole numbers
    unsigned acc_w = 0u, acc_mu = 0u, acc_e = 0u;
    unsigned row_s = (unsigned)it0 * 64u * (unsigned)sizeof(OS) + lane_s;
    unsigned row_c = (unsigned)it0 * 64u * (unsigned)sizeof(OC) + lane_c;            // (row of symbol it: ROFF = 0)
    unsigned qa = q_addr;
    const unsigned mu_lim = (unsigned)__double2hiint(spsd);
    IN e0, e1, m0, m1, l0, l1;                           // rows k, k+1, mid, mid+1, next, next+1
    double i_a = 0.0;
    unsigned ev_mu = 0u, hi_c = 0u, w_cur = 0u;

    auto issue = [&]() {
        i_a = base + mu;                                 // idx_k, hw1/gardner.py:47
        const double i_c = basen + mu;                   // idx_next, :48
        const int r_a = __double2int_rz(i_a);
        hi_c = (unsigned)__double2hiint(i_c);
        w_cur = (unsigned)(r_a - win_lo);
        ev_mu = (unsigned)__double2hiint(mu) & 0x7fffffffu;
        const unsigned rb = (unsigned)r_a * ROWB;
        if (GUARD) {
            const unsigned a0 = (rb & rmask) | colb;
            const unsigned am = a0 + midb, al = a0 + lateb;
            e0 = lds_ld<IN>(a0);
            e1 = lds_ld<IN>(a0 + ROWB);
            m0 = lds_ld<IN>(am);
            m1 = lds_ld<IN>(am + ROWB);
            l0 = lds_ld<IN>(al);
            l1 = lds_ld<IN>(al + ROWB);
        } else {
            e0 = lds_ld<IN>((rb & rmask) | colb);
            e1 = lds_ld<IN>(((rb + ROWB) & rmask) | colb);
            m0 = lds_ld<IN>(((rb + midb) & rmask) | colb);
            m1 = lds_ld<IN>(((rb + midb + ROWB) & rmask) | colb);
            l0 = lds_ld<IN>(((rb + lateb) & rmask) | colb);
            l1 = lds_ld<IN>(((rb + lateb + ROWB) & rmask) | colb);
        }
    };

    issue();
    const unsigned hi_first = (unsigned)__double2hiint(i_a);
    acc_w = w_cur;
    unsigned hi_last = 0u, w_last = 0u;
    for (int s = 0; s < n; s++) {
        hi_last = hi_c; w_last = w_cur;
        acc_mu = ev_mu > acc_mu ? ev_mu : acc_mu;
        const double f = dspm::fract_pos(i_a);           // == the fraction of idx_mid and of idx_next, see above
        const cd xa = dspm::lerp(widen<IN, SC>(e0, scale), widen<IN, SC>(e1, scale), f);
        const cd xb = dspm::lerp(widen<IN, SC>(m0, scale), widen<IN, SC>(m1, scale), f);
        const cd xc = dspm::lerp(widen<IN, SC>(l0, scale), widen<IN, SC>(l1, scale), f);
        const double e = dspm::ted_error(xa, xb, xc);    // :61
        const double de = delta * e;
        const unsigned ah = (unsigned)__double2hiint(de) & 0x7fffffffu;
        acc_e = ah > acc_e ? ah : acc_e;
        const double mu_h = mu, pos = i_a;               // :63 mu_history holds the old mu; x_k sits at idx_k
        mu = mu + de;                                    // :66 (the clip of :69 is inactive: acc_mu)
        base += spsd; basen += spsd;
        issue();
        const OC rv = {(OS)xa.re, (OS)xa.im};            // :72
        if (MASK == 7) GS::c_s_s(row_c, rv, sb.rec_t, row_s, (OS)e, sb.err_t, (OS)mu_h, sb.mu_t);
        else if (MASK == 15) GS::c_s_s_s(row_c, rv, sb.rec_t, row_s, (OS)e, sb.err_t, (OS)mu_h, sb.mu_t, (OS)pos, sb.pos_t);
        else if (MASK == 12) GS::s_s(row_s, (OS)mu_h, sb.mu_t, (OS)pos, sb.pos_t);
        else if (MASK == 4) GS::s(row_s, (OS)mu_h, sb.mu_t);
        else if (MASK != 0) {
            if (sb.rec_t) sst(sb.rec_t, row_c, rv);
            if (sb.err_t) sst(sb.err_t, row_s, (OS)e);
            if (sb.mu_t) sst(sb.mu_t, row_s, (OS)mu_h);
            if (sb.pos_t) sst(sb.pos_t, row_s, (OS)pos);
        }
        const d2v qv = {xa.re, xa.im};
        lds_st<d2v>(qa, qv);
        row_s += 64u * (unsigned)sizeof(OS); row_c += 64u * (unsigned)sizeof(OC); qa += 1024u;
    }
    acc_mu = ev_mu > acc_mu ? ev_mu : acc_mu;            // the mu the segment leaves behind was not clipped either
    const unsigned acc_x = hi_first ^ hi_last;
    acc_w = w_last > acc_w ? w_last : acc_w;
    const bool bad = (acc_x >> 20) != 0u || acc_w > (unsigned)wlim || acc_mu >= mu_lim || acc_e >= 0x3ff00000u;
    if (__any(bad)) return false;
    gs.mu = mu;
    gs.k += n;
    return true;
}

// The FLL wave's side of it: a queue block whose symbols all came from speculative segments (every lane
// produced every symbol) through fll_step_fast, one vote at the end (a derotated sample with a component
// exactly zero, a quadrant index beyond int32: the caller then runs the block through fll_step from the
// saved state).  MASK: 1 sig, 2 freq, 4 ef, 8 dec; -1 = decided at run time per store (specialised: all four,
// sig + freq + dec, sig + dec, freq + dec, dec: the subsets the host layers ask for).
struct SpecBasesF { const unsigned char *sig_t, *freq_t, *ef_t, *dec_t; };

template <typename OS, typename OC, int MASK>
__device__ __forceinline__ bool fll_spec_block(const dspm::MathK &MK, const int n, const unsigned q_addr, const d2v x0, const int row0,
                                               const double i0, double &freq, double &integ, const double kp,
                                               const double ki, const SpecBasesF &sb, const unsigned lane)
{
    typedef GroupStore<OC> GS;
    double f = freq, g = integ;
    dspm::FllEvidence ev;
    dspm::fll_evidence_init(ev);
    double idx = (double)row0 + i0, idx4 = 4.0 * idx;
    unsigned off_c = (unsigned)row0 * 64u * (unsigned)sizeof(OC) + lane * (unsigned)sizeof(OC);
    unsigned off_s = (unsigned)row0 * 64u * (unsigned)sizeof(OS) + lane * (unsigned)sizeof(OS);
    unsigned off_b = (unsigned)row0 * 64u + lane;
    unsigned qa = q_addr;
    d2v x = x0;                                          // the block's first sample, read beside the block's flags
    for (int t = 0; t < n; t++) {
        qa += 1024u;
        const d2v xn = lds_ld<d2v>(qa);                  // next symbol's sample (behind the block: read, not used)
        constexpr int FSEL = (DSP_P8_BFI_F && sizeof(OC) == 8) ? 1 : DSP_BFI_DEFAULT;
        const dspm::FllOut o = dspm::fll_step_fast<FSEL>(MK, x.x, x.y, idx, idx4, f, g, kp, ki, ev);
        const OC sv = {(OS)o.sr, (OS)o.si};
        if (MASK == 15) GS::c_s_s_b(off_c, sv, sb.sig_t, off_s, (OS)o.freq, sb.freq_t, (OS)o.ef, sb.ef_t, off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK == 11) GS::c_s_b(off_c, sv, sb.sig_t, off_s, (OS)o.freq, sb.freq_t, off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK == 9) GS::c_b(off_c, sv, sb.sig_t, off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK == 10) GS::s_b(off_s, (OS)o.freq, sb.freq_t, off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK == 8) GS::b(off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK != 0) {
            if (sb.sig_t) sst(sb.sig_t, off_c, sv);
            if (sb.freq_t) sst(sb.freq_t, off_s, (OS)o.freq);
            if (sb.ef_t) sst(sb.ef_t, off_s, (OS)o.ef);
            if (sb.dec_t) sst_b8(sb.dec_t, off_b, (unsigned)o.quad);
        }
        idx += 1.0; idx4 += 4.0;
        off_c += 64u * (unsigned)sizeof(OC); off_s += 64u * (unsigned)sizeof(OS); off_b += 64u;
        x = xn;
    }
    if (__any(dspm::fll_evidence_bad(ev))) return false;
    freq = f; integ = g;
    return true;
}

// fll_func on its own (the FLL-only form of the one-wave kernel): the T symbols of a DMA block through
// fll_step_fast straight from the LDS ring -- the FLL wave's fll_spec_block with the ring in place of the
// hand-off queue (ring rows wrap: one masked address per symbol).  x_io: the sample of row0 on entry, of
// row0 + n on return.  The caller guarantees that rows row0 + 1 ... row0 + n are inside the ring window and
// inside every lane's burst, and that the output rows fit 32-bit offsets.  Returns false (state untouched;
// the caller runs the block through fll_step) when the evidence word says so.
template <typename IN, bool SC, int MASK>
__device__ __forceinline__ bool fll_ring_block(const dspm::MathK &MK, const Tile<IN> &tl, const int n, cd &x_io, const int row0,
                                               const double i0, double &freq, double &integ, const double kp,
                                               const double ki, const SpecBasesF &sb, const unsigned lane)
{
    typedef typename InT<IN>::OS OS;
    typedef typename InT<IN>::OC OC;
    typedef GroupStore<OC> GS;
    double f = freq, g = integ;
    dspm::FllEvidence ev;
    dspm::fll_evidence_init(ev);
    double idx = (double)row0 + i0, idx4 = 4.0 * idx;
    unsigned off_c = (unsigned)row0 * 64u * (unsigned)sizeof(OC) + lane * (unsigned)sizeof(OC);
    unsigned off_s = (unsigned)row0 * 64u * (unsigned)sizeof(OS) + lane * (unsigned)sizeof(OS);
    unsigned off_b = (unsigned)row0 * 64u + lane;
    const unsigned colb = tl.lds_base + (unsigned)tl.col * (unsigned)sizeof(IN);
    const unsigned rm = (unsigned)tl.RM;
    const double scale = tl.scale;
    unsigned r = (unsigned)row0;
    cd x = x_io;
    for (int t = 0; t < n; t++) {
        r += 1u;
        const IN vn = lds_ld<IN>(colb + (r & rm) * (64u * (unsigned)sizeof(IN)));   // next symbol's sample
        const dspm::FllOut o = dspm::fll_step_fast(MK, x.re, x.im, idx, idx4, f, g, kp, ki, ev);
        const OC sv = {(OS)o.sr, (OS)o.si};
        if (MASK == 15) GS::c_s_s_b(off_c, sv, sb.sig_t, off_s, (OS)o.freq, sb.freq_t, (OS)o.ef, sb.ef_t, off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK == 10) GS::s_b(off_s, (OS)o.freq, sb.freq_t, off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK == 8) GS::b(off_b, (unsigned)o.quad, sb.dec_t);
        else if (MASK != 0) {
            if (sb.sig_t) sst(sb.sig_t, off_c, sv);
            if (sb.freq_t) sst(sb.freq_t, off_s, (OS)o.freq);
            if (sb.ef_t) sst(sb.ef_t, off_s, (OS)o.ef);
            if (sb.dec_t) sst_b8(sb.dec_t, off_b, (unsigned)o.quad);
        }
        idx += 1.0; idx4 += 4.0;
        off_c += 64u * (unsigned)sizeof(OC); off_s += 64u * (unsigned)sizeof(OS); off_b += 64u;
        x = widen<IN, SC>(vn, scale);
    }
    if (__any(dspm::fll_evidence_bad(ev))) return false;
    freq = f; integ = g; x_io = x;
    return true;
}

// ---------------------------------------------------------------------------
// The timing loop with a (2H+1)-point Lagrange interpolator (hw1/test_lagrange_resampling.py:34-72) in
// place of interpolate(): steady-state iteration, straight-line code out of the LDS ring.
//   * basis values in product form, L_i(f) = w_i * prod_{j != i} (f - o_j), o = -H..H, from prefix and
//     suffix products: no division and no exact-node case (the barycentric form of lagrange_at divides
//     twice per tap and once per output; both evaluate the same polynomial, they agree to ~1e-15);
//   * early, mid and late instants differ by whole samples, so they share ONE set of basis values
//     (when their fractions are not bit-identical -- a power of two between them -- the iteration is
//     left to the per-lane code); the output instant k + mu_new needs a second set and, as long as it
//     falls between the same two samples as the late one, no further samples.
// Returns false (state untouched) when a lane fails the vote.
template <int H>
__device__ __forceinline__ void lagr_basis(double f, double (&L)[2 * H + 1])
{
    constexpr int T = 2 * H + 1;
    dspm::LagrangeW W;
    dspm::lagrange_weights(H, W);                       // compile-time constants
    double d[T], pre[T], suf[T];
#pragma unroll
    for (int i = 0; i < T; i++) d[i] = f - (double)(i - H);
    pre[0] = 1.0;
#pragma unroll
    for (int i = 1; i < T; i++) pre[i] = pre[i - 1] * d[i - 1];
    suf[T - 1] = 1.0;
#pragma unroll
    for (int i = T - 2; i >= 0; i--) suf[i] = suf[i + 1] * d[i + 1];
#pragma unroll
    for (int i = 0; i < T; i++) L[i] = W.w[i] * (pre[i] * suf[i]);
}

template <int H>
__device__ __forceinline__ cd lagr_dot(const double (&L)[2 * H + 1], const cd (&x)[2 * H + 1])
{
    cd z; z.re = 0.0; z.im = 0.0;
#pragma unroll
    for (int i = 0; i < 2 * H + 1; i++) {
        z.re = __builtin_fma(L[i], x[i].re, z.re);
        z.im = __builtin_fma(L[i], x[i].im, z.im);
    }
    return z;
}

template <int H, int OFF, int N>
__device__ __forceinline__ cd lagr_dot_at(const double (&L)[2 * H + 1], const cd (&x)[N])   // taps on x[OFF .. OFF + 2H]
{
    static_assert(OFF + 2 * H + 1 <= N, "window inside the samples");
    cd z; z.re = 0.0; z.im = 0.0;
#pragma unroll
    for (int i = 0; i < 2 * H + 1; i++) {
        z.re = __builtin_fma(L[i], x[OFF + i].re, z.re);
        z.im = __builtin_fma(L[i], x[OFF + i].im, z.im);
    }
    return z;
}

template <typename IN, bool SC, int H>
__device__ __forceinline__ bool g2_steady_lagr(const Tile<IN> &tl, dspm::GState &gs, const TimingConst &tc,
                                               const dspm::LagrangeW *Lp, cd &rec, double &err, double &pos)
{
    constexpr int T = 2 * H + 1;
    const double mu = gs.mu, spsd = tc.spsd;
    const double kd = (double)gs.k;
    const double i_a = (kd - spsd) + mu;                 // gardner2.py:76-78
    const double i_b = (kd - tc.halfd) + mu;
    const double i_c = kd + mu;
    const int r_a = __double2int_rz(i_a), r_b = __double2int_rz(i_b), r_c = __double2int_rz(i_c);
    const double f = dspm::fract_pos(i_a);
    const bool in = (mu < spsd) & (mu >= -spsd) & (r_a - H >= tl.win_lo) & (r_c + 1 + H < tl.win_hi) &
                    (int)(dspm::fract_pos(i_b) == f) & (int)(dspm::fract_pos(i_c) == f);
    if (!__all(in)) return false;
    double L[T];
    lagr_basis<H>(f, L);
    cd va, vb, vc, xc[T];
    if (tc.sps == 4) {
        // equal fractions: the instants are exactly 2 and 4 samples apart, so their 3 x (2H+1) samples are
        // 2H+5 different ones -- each read and widened once (same products in the same order: same bits)
        cd u[T + 4];
#pragma unroll
        for (int i = 0; i < T + 4; i++) u[i] = tl.template lds_at<SC>(r_a - H + i);
        va = lagr_dot_at<H, 0>(L, u); vb = lagr_dot_at<H, 2>(L, u); vc = lagr_dot_at<H, 4>(L, u);
#pragma unroll
        for (int i = 0; i < T; i++) xc[i] = u[4 + i];
    } else {
        cd xa[T], xb[T];
#pragma unroll
        for (int i = 0; i < T; i++) {
            xa[i] = tl.template lds_at<SC>(r_a - H + i);
            xb[i] = tl.template lds_at<SC>(r_b - H + i);
            xc[i] = tl.template lds_at<SC>(r_c - H + i);
        }
        va = lagr_dot<H>(L, xa); vb = lagr_dot<H>(L, xb); vc = lagr_dot<H>(L, xc);
    }
    const double e = dspm::ted_error(va, vb, vc);        // :87
    const double mu_n = mu - tc.alpha * e;               // :91
    const double i_o = kd + mu_n;                        // :95
    const int r_o = __double2int_rz(i_o);
    if (__all(r_o == r_c)) {                             // the late instant's samples serve
        double Lo[T];
        lagr_basis<H>(dspm::fract_pos(i_o), Lo);
        rec = lagr_dot<H>(Lo, xc);
    } else if (__all((i_o > 0.0) & (i_o < 1.0e9) & (r_o - H >= tl.win_lo) & (r_o + 1 + H < tl.win_hi))) {
        double Lo[T];
        lagr_basis<H>(dspm::fract_pos(i_o), Lo);
        cd xo[T];
#pragma unroll
        for (int i = 0; i < T; i++) xo[i] = tl.template lds_at<SC>(r_o - H + i);
        rec = lagr_dot<H>(Lo, xo);
    } else {
        rec = dspm::g2_interp(i_o, tc.len, Lp, FetchFn<IN>{tl});
    }
    err = e;
    pos = i_o;
    gs.mu = mu_n;
    gs.k += tc.sps;                                      // :98
    return true;
}

// sample position of the symbol an iteration just produced: k + mu with the updated mu
// (gardner2.py:95) / idx_k (hw1/gardner.py:47); the k in gs has already advanced
template <int VARIANT>
__device__ __forceinline__ double symbol_position(const dspm::GState &gs, const dspm::GOut &go, int sps)
{
    return VARIANT == DSP_VARIANT_TED ? (double)((gs.k - 1) * sps) + go.mu : (double)(gs.k - sps) + go.mu;
}

// FLL step for every lane, committed by select (no divergent control flow)
struct FllLane { double sr, si, ef, freq; unsigned quad; };
__device__ __forceinline__ FllLane fll_iter(const dspm::MathK &MK, bool pending, cd x, double idx,
                                            double &freq, double &integ, double kp, double ki,
                                            dspm::TrAcc *tr = nullptr)
{
    double nf = freq, ni = integ;
    dspm::FllOut fo = dspm::fll_step(MK, x.re, x.im, idx, nf, ni, kp, ki, pending, tr);
    freq = pending ? nf : freq;
    integ = pending ? ni : integ;
    FllLane o;
    o.sr = pending ? fo.sr : 0.0; o.si = pending ? fo.si : 0.0;
    o.ef = pending ? fo.ef : 0.0; o.freq = pending ? fo.freq : 0.0;
    o.quad = pending ? (unsigned)fo.quad : 0u;
    return o;
}

template <typename IN>
__device__ __forceinline__ void tile_init(Tile<IN> &tl, const SyncParams &P, unsigned char *ring,
                                          int tile, int lane, double scale)
{
    tl.xt = tile_rows<IN>(P, tile);
    tl.ring = ring;
    tl.lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)ring;
    tl.lane = lane; tl.col = lane; tl.R = P.ring_rows; tl.RM = P.ring_rows - 1; tl.in_rows = P.in_rows;
    tl.win_lo = 0; tl.win_hi = 0; tl.F = 0; tl.scale = scale; tl.guard = P.guard_rows;
}

__device__ __forceinline__ void timing_init(TimingConst &tc, dspm::GState &gs, const SyncParams &P,
                                            int variant, int len, bool lane_ok, int b)
{
    tc.len = len; tc.sps = P.sps; tc.spsd = (double)P.sps; tc.lend = (double)len; tc.alpha = P.alpha;
    tc.halfd = variant == DSP_VARIANT_TED ? tc.spsd / 2.0 : (double)(P.sps / 2);
    tc.n_symbols = 0;
    gs.mu = variant == DSP_VARIANT_GARDNER2 ? ((P.mu0_per && lane_ok) ? P.mu0_per[b] : P.mu0) : 0.0;
    gs.k = variant == DSP_VARIANT_GARDNER2 ? P.sps : 0;
    gs.active = lane_ok && len > 0;
    if (variant == DSP_VARIANT_TED) {
        int ns = len - 2 * P.sps;                       // (len - 2*sps)//sps, floor
        tc.n_symbols = ns >= 0 ? ns / P.sps : -((-ns + P.sps - 1) / P.sps);
        gs.active = lane_ok;
    }
}

// ---------------------------------------------------------------------------
// Unified kernel: one wave per tile does the timing loop and the FLL.
template <typename IN, int VARIANT, bool FLL, bool LAGR, bool SC>
__global__ __launch_bounds__(64) void sync_kernel(SyncParams P)
{
    typedef typename InT<IN>::OS OS;     // output scalar (float for c64 in, double for c128 in)
    typedef typename InT<IN>::OC OC;     // output complex

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int b = tile * 64 + lane;
    const bool lane_ok = b < P.B;
    int len = lane_ok ? (P.lens ? P.lens[b] : P.len_all) : 0;
    double scale = (lane_ok && P.scale) ? P.scale[b] : 1.0;
    // fll_func on its own: the padding lanes of a partly filled tile (63 of them when ONE burst is given) run
    // as clones of the tile's longest burst -- its column, scale and start values -- so that the block path
    // below, which needs every lane inside its burst, serves such tiles too; a clone reports nothing (its
    // output columns are padding), as the two-wave kernel's passenger lanes
    constexpr bool FBLK = VARIANT == DSP_VARIANT_NONE && FLL && !LAGR;
    bool pass = false;
    int src = lane;
    if constexpr (FBLK) {
        int lmax = len;
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(lmax, d); lmax = o > lmax ? o : lmax; }
        const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(lane_ok && len == lmax)) - 1);
        pass = !lane_ok && lmax > 0 && first >= 0;
        const double sc = __shfl(scale, first < 0 ? 0 : first);
        if (pass) { len = lmax; src = first; scale = sc; }
    }
    const int out_rows = P.out_rows, sps = P.sps, T = P.T;
    const int cap_rows = out_rows - 1;   // the last output row is scratch (uniform stores land there)
    const size_t obase = (size_t)tile * out_rows * 64 + lane;
    const bool ring = P.use_ring != 0;

    Tile<IN> tl;
    tile_init(tl, P, smem, tile, lane, scale);
    tl.col = src;
    if (ring) tl.prologue(sps, T, P.C);

    dspm::LagrangeW LW;
    const dspm::LagrangeW *Lp = nullptr;
    if (LAGR) { dspm::lagrange_weights(P.half_len, LW); Lp = &LW; }

    TimingConst tc;
    dspm::GState gs;
    timing_init(tc, gs, P, VARIANT, len, lane_ok, b);
    double freq = 0.0, integ = 0.0, i0 = 0.0;
    if (FLL && lane_ok) {
        if (P.fll_in) { freq = P.fll_in[2 * b]; integ = P.fll_in[2 * b + 1]; }
        if (P.i0) i0 = P.i0[b];
    }
    if constexpr (FBLK) {
        const double f_s = __shfl(freq, src), g_s = __shfl(integ, src), i_s = __shfl(i0, src);
        if (pass) { freq = f_s; integ = g_s; i0 = i_s; gs.active = true; }
    }
    const double kp = P.kp, ki = P.ki;
    dspm::MathK MK;
    if (FLL) MK = load_math_consts();

    OC *o_rec = (OC *)P.rec; OS *o_terr = (OS *)P.terr; OS *o_mu = (OS *)P.mu; OS *o_pos = (OS *)P.pos;
    OC *o_sig = (OC *)P.sig; OS *o_freq = (OS *)P.freq; OS *o_ef = (OS *)P.ef;
    uint8_t *o_dec = P.dec;

    // first symbol: recovered[0] = signal[0] (gardner2.py:59) / x[0] for the FLL-only form
    constexpr int ROFF = (VARIANT == DSP_VARIANT_TED) ? 1 : 0;   // FLL lags the TED by one row
    bool pending = false;
    cd rec_prev; rec_prev.re = 0.0; rec_prev.im = 0.0;
    int n_rec = 0;
    if (VARIANT != DSP_VARIANT_TED) {
        if (gs.active) { rec_prev = tl.fetch(0); pending = true; n_rec = 1; }
        if (VARIANT == DSP_VARIANT_GARDNER2 && o_rec) st_cplx(o_rec + obase, rec_prev.re, rec_prev.im);
    }
    if (VARIANT == DSP_VARIANT_NONE) gs.active = false;
    bool overflow = false;

    // fll_func on its own: whole DMA blocks through fll_ring_block while every lane of the tile is inside
    // its burst (one vote per block instead of the per-symbol path's per-lane selects and `easy` vote)
    int min_len = 0, fmask = 0;
    SpecBasesF fbases = {nullptr, nullptr, nullptr, nullptr};
    if constexpr (FBLK) {
        min_len = len;
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(min_len, d); min_len = o < min_len ? o : min_len; }
        min_len = __builtin_amdgcn_readfirstlane(min_len);
        if (!ring || P.spec == 0) min_len = 0;               // (no ring, or the A/B switch: per-symbol path only)
        fmask = (o_sig ? 1 : 0) | (o_freq ? 2 : 0) | (o_ef ? 4 : 0) | (o_dec ? 8 : 0);
        const size_t tile_el = (size_t)tile * out_rows * 64;
        fbases = {o_sig ? (const unsigned char *)(o_sig + tile_el) : nullptr,
                  o_freq ? (const unsigned char *)(o_freq + tile_el) : nullptr,
                  o_ef ? (const unsigned char *)(o_ef + tile_el) : nullptr,
                  o_dec ? (const unsigned char *)(o_dec + tile_el) : nullptr};
    }

    int it = 0;
    bool running = true;
    for (int blk = 0; running; blk++) {
        if (ring) tl.advance(blk, sps, T, P.C, P.nwait);
        if constexpr (FBLK) {
            // rows it + 1 ... it + T are read (the last one becomes the next block's first sample)
            if (it + T < min_len && it + 1 >= tl.win_lo && it + T < tl.win_hi && it + T <= cap_rows &&
                (unsigned long long)(it + T + 1) * 64ull * sizeof(OC) <= 0xffffffffull) {
                bool blk_done = false;
                switch (fmask) {
                    case 15: blk_done = fll_ring_block<IN, SC, 15>(MK, tl, T, rec_prev, it, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 10: blk_done = fll_ring_block<IN, SC, 10>(MK, tl, T, rec_prev, it, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 8: blk_done = fll_ring_block<IN, SC, 8>(MK, tl, T, rec_prev, it, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    default: blk_done = fll_ring_block<IN, SC, -1>(MK, tl, T, rec_prev, it, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                }
                if (blk_done) { it += T; n_rec += T; continue; }
            }
        }
        for (int t = 0; t < T; t++, it++) {
            bool more = (VARIANT == DSP_VARIANT_NONE) ? pending : (gs.active || (FLL && pending));
            if (!__any(more)) { running = false; break; }
            const int rec_row = (VARIANT == DSP_VARIANT_TED) ? it : it + 1;
            const int fll_row = it - ROFF;
            if (fll_row >= cap_rows) { running = false; break; }      // capacity (uniform)

            FllLane fo; fo.sr = 0.0; fo.si = 0.0; fo.ef = 0.0; fo.freq = 0.0; fo.quad = 0;
            if (FLL) fo = fll_iter(MK, pending, rec_prev, (double)fll_row + i0, freq, integ, kp, ki);

            dspm::GOut go; go.produced = false; go.rec.re = 0.0; go.rec.im = 0.0; go.err = 0.0; go.mu = 0.0;
            if (VARIANT == DSP_VARIANT_NONE) {
                const bool have = pending && (it + 1 < len);
                if (it + 1 >= tl.win_lo && it + 1 < tl.win_hi) {            // uniform
                    go.rec = tl.template lds_at<SC>(it + 1);
                } else if (have) {
                    go.rec = tl.fetch(it + 1);
                }
                go.produced = have;
            } else {
                if (gs.active && rec_row >= cap_rows) { gs.active = false; overflow = true; }
                go = timing_iter<IN, VARIANT, LAGR, SC>(tl, gs, tc, Lp);
            }
            // ---- stores: unconditional, one instruction per enabled output -
            const int frow = fll_row < 0 ? 0 : fll_row;
            const size_t fo_off = obase + (size_t)frow * 64;
            if (FLL) {
                if (o_sig) st_cplx(o_sig + fo_off, fo.sr, fo.si);
                if (o_freq) st_val(o_freq + fo_off, fo.freq);
                if (o_ef) st_val(o_ef + fo_off, fo.ef);
                if (o_dec) st_b8(o_dec + fo_off, fo.quad);
            }
            if (VARIANT != DSP_VARIANT_NONE) {
                const int rrow = rec_row < cap_rows ? rec_row : cap_rows;   // last row = scratch
                const size_t g_off = obase + (size_t)it * 64;
                if (o_rec) st_cplx(o_rec + obase + (size_t)rrow * 64, go.rec.re, go.rec.im, go.produced);
                if (o_terr) st_val(o_terr + g_off, go.err, go.produced);
                if (o_mu) st_val(o_mu + g_off, go.mu, go.produced);
                if (o_pos) st_val(o_pos + g_off, symbol_position<VARIANT>(gs, go, sps), go.produced);
            }
            pending = go.produced;
            rec_prev = go.rec;
            n_rec += go.produced ? 1 : 0;
        }
    }
    if (lane_ok) {
        if (VARIANT != DSP_VARIANT_NONE && gs.active) overflow = true;   // stopped by capacity
        if (FLL && pending) overflow = true;
        if (P.n_out) P.n_out[b] = n_rec;
        if (P.status) P.status[b] = overflow ? 1 : 0;
        if (FLL && P.fll_out) { P.fll_out[2 * b] = freq; P.fll_out[2 * b + 1] = integ; }
    }
}

// Wave priority that rotates with progress.  The SIMD arbiter serves the oldest wave of the
// highest priority first; with several tiles per SIMD that lets the first-launched waves race
// ahead and leaves the youngest to finish alone at one-wave speed.  Rotating the priority
// among the co-resident wave slots gives every wave the same share over time.
__device__ __forceinline__ void rotate_priority(unsigned slot, int phase)
{
    switch ((slot + (unsigned)phase) & 3u) {
        case 0: asm volatile("s_setprio 0"); break;
        case 1: asm volatile("s_setprio 1"); break;
        case 2: asm volatile("s_setprio 2"); break;
        default: asm volatile("s_setprio 3"); break;
    }
}
__device__ __forceinline__ unsigned wave_slot()
{
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));   // wave slot on its SIMD
    return hw;
}

// ---------------------------------------------------------------------------
// Split kernel: two waves per tile.  Wave 0 runs the timing loop (ring, DMA,
// rec/err/mu stores) and hands each recovered symbol to wave 1 through a
// double-buffered LDS queue; wave 1 runs the FLL, the slicer and their stores
// one queue block behind.  The two recurrences are independent chains, so
// giving each its own wave doubles the waves per SIMD for the same bursts and
// halves the per-wave instruction stream (the kernel is issue/latency bound:
// one wave issues at most one instruction per 4 cycles).
//   LDS: [ring][queue 2*TQ*64 complex128][flags 2*TQ*64 bytes][more[2]]
//   Block qb: wave 0 fills queue[qb&1], s_barrier; wave 1 drains it while wave 0
//   fills the other half.  Every block both waves execute exactly one barrier.
// NWV = waves launched per tile: 2, or 3 / 4 with idle waves for the dispatcher (below); separate
// instantiations so that the plain launch compiles exactly as it did without them (0.9 % at B = 65 536)
#ifndef DSP_SPLIT_MIN_WAVES
#define DSP_SPLIT_MIN_WAVES 1
#endif
#ifdef DSP_NOSPEC            // A/B builds: every instantiation without the speculative segments
#define DSP_SPECT_DEFAULT false
#else
#define DSP_SPECT_DEFAULT true
#endif
// FLLON = false: the timing loop on its own (gardner_timing_recovery / gardner_ted without fll_func behind
// them).  The second wave keeps its other job -- it issues, awaits and mirrors the ring's LDS-DMA for the
// timing wave -- and runs no FLL; the timing wave is the one of the fused kernel, speculative segments and
// all (part 7 of this file instantiates these; the one-wave kernel ran such a call on its per-symbol path).
// DMAW = true (with NWV = 3; part 8): three LIVE waves per tile -- timing, FLL, and a third that does nothing
// but feed the ring (what the FLL wave does beside its own chain in the two-wave launch: ~70 cycles per KiB
// issued, 4 KiB per symbol of complex128 at 4 sps).  For launches of at most one tile per CU, where the
// dispatcher puts the three waves on three SIMDs and the per-symbol latency of the slower chain is the
// launch's time.  The third wave takes part in every barrier of the FLL wave's schedule.
template <typename IN, int VARIANT, bool SC, int NWV, int H = 0, bool SPECT = DSP_SPECT_DEFAULT, bool FLLON = true,
          bool DMAW = false>
__global__ __launch_bounds__(64 * NWV, (H == 0 ? DSP_SPLIT_MIN_WAVES : 1)) void sync_split_kernel(SyncParams P)
{
    typedef typename InT<IN>::OS OS;
    typedef typename InT<IN>::OC OC;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    // Launched with 192 threads the middle wave leaves at once: it is there for the dispatcher.  The
    // waves of a workgroup go to the SIMDs in the order 0, 2, 1, 3 starting one position further for
    // every next workgroup on the CU (tools/ubench/simd_place.hip), so two 2-wave workgroups on a CU
    // land on SIMDs (0,2) and (2,1): one SIMD carries two recurrences, one idles.  With the spare
    // in between the live waves land on (0,1) and (2,3).  A wave that has ended counts as arrived
    // at every later s_barrier of its workgroup.
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (NWV == 3 && !DMAW && wv == 1) return;
    int role = DMAW ? wv : (wv ? 1 : 0);                                 // 0 = timing, 1 = FLL, 2 = ring feeder (DMAW)
    const int tile = blockIdx.x;
    const int b = tile * 64 + lane;
    const bool lane_ok = b < P.B;
    int len = lane_ok ? (P.lens ? P.lens[b] : P.len_all) : 0;
    // Passenger lanes (both timing loops; gardner_ted since the end of round 3: ONE burst through
    // hw1/gardner.py's loop, the drop-in call, stayed on the per-lane path for its 63 idle lanes).
    // The steady path needs every lane of the wave active, so one idle lane -- the padding lanes of a partly filled last tile, or an empty window of the capture
    // mode -- would keep its whole tile on the slower per-lane path and make that tile the launch's
    // straggler.  Such a lane runs as a CLONE of the tile's longest burst (same column, scale and start
    // values, so it does whatever that burst does and never holds the wave up) and nothing it computes
    // is reported: its output columns are padding (b >= B) or, for empty bursts, only when the caller
    // says that they are never read (idle_pass: the capture mode's stitch skips empty windows).
    bool pass = false;
    int src = lane;                                                      // the lane a passenger copies
    double scale = (lane_ok && P.scale) ? P.scale[b] : 1.0;
    if (VARIANT == DSP_VARIANT_GARDNER2 || VARIANT == DSP_VARIANT_TED) {
        int lmax = len;
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(lmax, d); lmax = o > lmax ? o : lmax; }
        pass = (!lane_ok || (P.idle_pass && len == 0)) && lmax > 0;
        const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(len == lmax)) - 1);
        const double sc = __shfl(scale, first);
        if (pass) { len = lmax; src = first; scale = sc; }
    }
    const int out_rows = P.out_rows, sps = P.sps, T = P.T, TQ = P.TQ;
    const int cap_rows = out_rows - 1;   // the last output row is scratch
    const size_t obase = (size_t)tile * out_rows * 64 + lane;
    const bool ring = P.use_ring != 0;

    unsigned char *q_base = smem + (size_t)(P.ring_rows + P.guard_rows) * 64 * sizeof(IN) * (ring ? 1 : 0);
    d2v *q_data = (d2v *)q_base;                                    // [2][TQ][64]
    unsigned char *q_flag = q_base + (size_t)2 * TQ * 64 * sizeof(d2v);   // [2][TQ][64]
    // explicit LDS pointer: a generic (flat) access would have to wait for vmcnt(0) as well
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int *q_more = (lds_int *)(q_flag + (size_t)2 * TQ * 64);               // [2]
    constexpr int ROFF = (VARIANT == DSP_VARIANT_TED) ? 0 : 1;      // FLL row of queue entry `it` = it + ROFF
    if (NWV == 4) {
        // Four waves: the dispatcher puts them on the CU's four SIMDs (order 0, 2, 1, 3 from wherever
        // it stands).  The workgroup draws a ticket from its CU's counter; the two waves on SIMD pair
        // {0,1} (even ticket) or {2,3} (odd) become the timing and the FLL wave, the other two leave.
        // Two workgroups on a CU draw consecutive tickets, so their live waves never share a SIMD,
        // whatever kernels ran before (the 192-thread launch relies on the dispatcher's position).
        lds_int *sel = q_more + 2;                                  // [0..3] SIMD of wave i, [4] ticket parity
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) sel[wv] = (int)((hw >> 4) & 3u);
        if (threadIdx.x == 0) {
            const unsigned cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            sel[4] = (int)(atomicAdd(P.cu_tickets + cu, 1u) & 1u);
        }
        __syncthreads();
        const int t = sel[4];
        int first = -1, second = -1, n = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((sel[i] >> 1) == t) { if (n == 0) first = i; else if (n == 1) second = i; n++; }
        if (n != 2) { first = 0; second = 2; }                      // not one wave per SIMD: any two will do
        first = __builtin_amdgcn_readfirstlane(first);
        second = __builtin_amdgcn_readfirstlane(second);
        if (wv != first && wv != second) return;
        role = wv == first ? 0 : 1;
        // self-check of the placement (dsp_set_option("place_ptr"), tests/test_gpu_fullsize.py)
        if (P.place && lane == 0 && role < 2) P.place[tile * 2 + role] = 0x80000000u | ((xcc & 15u) << 16) | (hw & 0xffffu);
    }
    const unsigned slot = wave_slot();
    int prio_n = P.prio_period, prio_c = 1, prio_ph = 0;
    if (P.prio_role) {                                  // (role is wave-uniform)
        if (role == 0) asm volatile("s_setprio 3"); else asm volatile("s_setprio 0");
    }
    // coarse timers of the trace build, kept in registers (the stamps build's TrAcc lives in scratch memory
    // and every stamp waits for vmcnt(0): with LDS-DMA in flight that measures the DMA, not the code)
#if defined(DSP_TRACE) && !defined(DSP_TRACE_STAMPS)
    unsigned long long tt_last = __builtin_amdgcn_s_memtime(), tt_a = 0, tt_b = 0, tt_c = 0, tt_d = 0, tt_e = 0, tt_f = 0;
#define DSP_TT(acc) do { const unsigned long long tt_now = __builtin_amdgcn_s_memtime(); acc += tt_now - tt_last; tt_last = tt_now; } while (0)
#else
#define DSP_TT(acc) ((void)0)
#endif
#ifdef DSP_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    dspm::TrAcc tr_acc;
    for (int i = 0; i < 16; i++) tr_acc.acc[i] = 0;
    tr_acc.last = tr_t0;
#ifdef DSP_TRACE_STAMPS
    dspm::TrAcc *trp = &tr_acc;          // segment stamps: a third build, slower than the real kernel
#else
    dspm::TrAcc *trp = nullptr;
#endif
#else
    dspm::TrAcc *trp = nullptr;
#endif

    if (role == 0) {
        // =================== timing wave ===================
        Tile<IN> tl;
        tile_init(tl, P, smem, tile, lane, scale);
        const bool dma_fll = ring && P.dma_fll != 0;    // the FLL wave feeds the ring (see its side below)
        if (dma_fll) {
            asm volatile("s_barrier" ::: "memory");     // B0: the first block's rows have landed
            tl.prologue_nodma(sps, T, P.C);
        } else if (ring) {
            tl.prologue(sps, T, P.C);
        }
        TimingConst tc;
        dspm::GState gs;
        timing_init(tc, gs, P, VARIANT, len, lane_ok, b);
        if (VARIANT == DSP_VARIANT_GARDNER2 || VARIANT == DSP_VARIANT_TED) {
            const double mu_src = __shfl(gs.mu, src);
            if (pass) { gs.active = true; gs.mu = mu_src; tl.col = src; }
        }
        dspm::LagrangeW LW;                             // H > 0: the per-lane code's barycentric weights
        const dspm::LagrangeW *Lp = nullptr;
        if (H > 0) { dspm::lagrange_weights(H, LW); Lp = &LW; }
        OC *o_rec = (OC *)P.rec; OS *o_terr = (OS *)P.terr; OS *o_mu = (OS *)P.mu; OS *o_pos = (OS *)P.pos;
        int n_rec = 0;
        if (VARIANT == DSP_VARIANT_GARDNER2) {
            cd r0; r0.re = 0.0; r0.im = 0.0;
            if (gs.active) { r0 = tl.fetch(0); n_rec = 1; }
            if (o_rec) st_cplx(o_rec + obase, r0.re, r0.im);
        }
        bool overflow = false;
        // wave-uniform facts for the steady-state path
        int min_len = len;
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(min_len, d); min_len = o < min_len ? o : min_len; }
        min_len = __builtin_amdgcn_readfirstlane(min_len);
        // (gardner_ted: the speculative segments are its whole steady path -- even samples-per-symbol counts)
        const bool ted_even = VARIANT == DSP_VARIANT_TED && sps >= 2 && (sps & 1) == 0;
        const bool steady_ok = ring && (VARIANT == DSP_VARIANT_GARDNER2 || (ted_even && H == 0 && P.spec != 0));
        int min_nsym = tc.n_symbols;                    // gardner_ted: symbols the tile's shortest burst runs
        if (VARIANT == DSP_VARIANT_TED) {
            for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(min_nsym, d); min_nsym = o < min_nsym ? o : min_nsym; }
            min_nsym = __builtin_amdgcn_readfirstlane(min_nsym);
        }
        bool all_active = __all(gs.active);
        const unsigned char *rec_t = (const unsigned char *)(o_rec + (size_t)tile * out_rows * 64);
        const unsigned char *err_t = (const unsigned char *)(o_terr + (size_t)tile * out_rows * 64);
        const unsigned char *mu_t = (const unsigned char *)(o_mu + (size_t)tile * out_rows * 64);
        const unsigned char *pos_t = (const unsigned char *)(o_pos + (size_t)tile * out_rows * 64);
        const unsigned lane_s = lane * (unsigned)sizeof(OS), lane_c = lane * (unsigned)sizeof(OC);
        int n_steady = 0;
        // speculative segments (g2_spec_segment): gardner2 with the linear interpolator, ring in use and
        // starting on a multiple of its size
        constexpr bool SPEC = (SPECT && H == 0 && (VARIANT == DSP_VARIANT_GARDNER2 || VARIANT == DSP_VARIANT_TED));
        const int tmask = (o_rec ? 1 : 0) | (o_terr ? 2 : 0) | (o_mu ? 4 : 0) | (o_pos ? 8 : 0);
        const bool spec_cfg = SPEC && ring && P.spec != 0 && (VARIANT == DSP_VARIANT_GARDNER2 || ted_even) &&
                              (tl.lds_base & ((unsigned)tl.R * 64u * (unsigned)sizeof(IN) - 1u)) == 0u;
        const SpecBases sbases = {o_rec ? rec_t : nullptr, o_terr ? err_t : nullptr, o_mu ? mu_t : nullptr,
                                  o_pos ? pos_t : nullptr};
        const unsigned q_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)q_base + (unsigned)lane * 16u;
        bool qb_spec = true;          // every symbol of the current queue block came from a speculative segment
        // DMA blocks (T iterations) and queue blocks (TQ iterations) tick independently
        int it = 0, tq = 0, buf = 0;
        bool more = true;
        for (int blk = 0; more; blk++) {
            if (prio_n && --prio_c == 0) { prio_c = prio_n; rotate_priority(slot, ++prio_ph); }
            DSP_TR(trp, 0);
            DSP_TT(tt_d);
            if (dma_fll) tl.advance_nodma(blk, sps, T, P.C);
            else if (ring) tl.advance(blk, sps, T, P.C, P.nwait);
            DSP_TT(tt_a);
            DSP_TR(trp, 9);
            // the steady path's stores address a row by a 32-bit byte offset from the tile's base: a
            // block whose last row would pass 4 GiB (row 2^22 of a complex128 output) takes the
            // size_t path below instead
            bool steady = steady_ok && all_active && tl.win_lo > 0 && it + T + ROFF < cap_rows &&
                          (unsigned long long)(it + T + ROFF + 1) * 64ull * sizeof(OC) <= 0xffffffffull &&
                          (VARIANT == DSP_VARIANT_GARDNER2 ? tl.win_hi + 2 * sps + 1 <= min_len
                                                           : (spec_cfg && it + T <= min_nsym));
            int t = 0;
            while (t < T) {
                // a segment ends with the DMA block or the queue block, whichever comes first
                const int n = T - t < TQ - tq ? T - t : TQ - tq;
                bool seg_done = false;
                if (SPEC && steady && spec_cfg && tl.win_hi - tl.win_lo >= sps + 3) {
                    const unsigned qa = q_lds + (unsigned)(buf * TQ + tq) * 1024u;
                    DSP_TR(trp, 10);
                    DSP_TT(tt_e);
#define DSP_SEG(M, G)                                                                                            \
    do {                                                                                                         \
        if constexpr (VARIANT == DSP_VARIANT_TED) seg_done = ted_spec_segment<IN, SC, M, G>(tl, gs, tc, n, it, sbases, lane_s, lane_c, qa); \
        else seg_done = g2_spec_segment<IN, SC, M, G>(tl, gs, tc, n, it, ROFF, sbases, lane_s, lane_c, qa);     \
    } while (0)
                    if (tl.guard) {
                        switch (tmask) {
                            case 7: DSP_SEG(7, true); break;
                            case 15: DSP_SEG(15, true); break;
                            case 12: DSP_SEG(12, true); break;
                            case 4: DSP_SEG(4, true); break;
                            case 0: DSP_SEG(0, true); break;
                            default: DSP_SEG(-1, true); break;
                        }
                    } else {
                        switch (tmask) {
                            case 7: DSP_SEG(7, false); break;
                            case 15: DSP_SEG(15, false); break;
                            case 12: DSP_SEG(12, false); break;
                            case 4: DSP_SEG(4, false); break;
                            case 0: DSP_SEG(0, false); break;
                            default: DSP_SEG(-1, false); break;
                        }
                    }
#undef DSP_SEG
                    DSP_TR(trp, 11);
                    DSP_TT(tt_b);
                    if (seg_done) {
                        if (!qb_spec)       // the block's other symbols carry per-lane flags: so must these
                            for (int i = 0; i < n; i++) q_flag[(buf * TQ + tq + i) * 64 + lane] = 1;
                        it += n; t += n; tq += n; n_steady += n;
                    }
                }
                if (!seg_done) {
                    if (qb_spec) {          // the FLL wave will read this block's flags: those of the segments behind us
                        for (int i = 0; i < tq; i++) q_flag[(buf * TQ + i) * 64 + lane] = 1;
                        qb_spec = false;
                    }
                    for (int i = 0; i < n; i++, t++, it++, tq++) {
                        DSP_TR(trp, 0);
                        bool done = false;
                        if (VARIANT == DSP_VARIANT_GARDNER2 && steady && (!SPEC || !spec_cfg)) {
                            cd rec; double e, ps;
                            if constexpr (H > 0) done = g2_steady_lagr<IN, SC, H>(tl, gs, tc, Lp, rec, e, ps);
                            else done = g2_steady<IN, SC>(tl, gs, tc, rec, e, ps);
                            if (done) {
                                DSP_TR(trp, 5);
                                const unsigned row_s = (unsigned)it * 64u * (unsigned)sizeof(OS);
                                const unsigned row_c = (unsigned)(it + ROFF) * 64u * (unsigned)sizeof(OC);
                                OC rv = {(OS)rec.re, (OS)rec.im};
                                if (o_rec) sst(rec_t, lane_c + row_c, rv);
                                if (o_terr) sst(err_t, lane_s + row_s, (OS)e);
                                if (o_mu) sst(mu_t, lane_s + row_s, (OS)gs.mu);
                                if (o_pos) sst(pos_t, lane_s + row_s, (OS)ps);
                                DSP_TR(trp, 6);
                                d2v qv = {rec.re, rec.im};
                                q_data[(buf * TQ + tq) * 64 + lane] = qv;
                                q_flag[(buf * TQ + tq) * 64 + lane] = 1;
                                n_steady++;
                            }
                        }
                        if (!done) {
                            const int rec_row = it + ROFF;
                            if (gs.active && rec_row >= cap_rows) { gs.active = false; overflow = true; }
                            dspm::GOut go = timing_iter<IN, VARIANT, (H > 0), SC>(tl, gs, tc, Lp, trp);
                            DSP_TR(trp, 5);
                            const int rrow = rec_row < cap_rows ? rec_row : cap_rows;
                            const int erow = it < cap_rows ? it : cap_rows;
                            const size_t g_off = obase + (size_t)erow * 64;
                            if (o_rec) st_cplx(o_rec + obase + (size_t)rrow * 64, go.rec.re, go.rec.im, go.produced);
                            if (o_terr) st_val(o_terr + g_off, go.err, go.produced);
                            if (o_mu) st_val(o_mu + g_off, go.mu, go.produced);
                            if (o_pos) st_val(o_pos + g_off, symbol_position<VARIANT>(gs, go, sps), go.produced);
                            DSP_TR(trp, 6);
                            d2v qv = {go.rec.re, go.rec.im};
                            q_data[(buf * TQ + tq) * 64 + lane] = qv;
                            q_flag[(buf * TQ + tq) * 64 + lane] = go.produced ? 1 : 0;
                            n_rec += go.produced ? 1 : 0;
                            if (steady) steady = __all(gs.active);
                        }
                        DSP_TR(trp, 7);
                    }
                }
                if (tq == TQ) {                        // hand the queue block over (uniform)
                    // (a block of speculative segments only: every lane was active and still is)
                    more = qb_spec ? true : __any(gs.active);
                    if (lane == 0) q_more[buf] = (more ? 1 : 0) | (qb_spec ? 2 : 0);
                    DSP_TT(tt_f);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    DSP_TT(tt_c);
                    DSP_TR(trp, 8);
                    tq = 0; buf ^= 1; qb_spec = true;
                    if (!more) break;
                }
            }
            all_active = __all(gs.active);
        }
        n_rec += n_steady;
        if (lane_ok) {
            if (P.n_out) P.n_out[b] = pass ? 0 : n_rec;
            if (P.status) P.status[b] = (overflow && !pass) ? 1 : 0;
        }
    } else {
        // =================== FLL wave ===================
        const IN *xt = tile_rows<IN>(P, tile);
        double freq = 0.0, integ = 0.0, i0 = 0.0;
        if (lane_ok) {
            if (P.fll_in) { freq = P.fll_in[2 * b]; integ = P.fll_in[2 * b + 1]; }
            if (P.i0) i0 = P.i0[b];
        }
        if (VARIANT == DSP_VARIANT_GARDNER2 || VARIANT == DSP_VARIANT_TED) {                           // a passenger starts where its model starts
            const double f_s = __shfl(freq, src), g_s = __shfl(integ, src), i_s = __shfl(i0, src);
            if (pass) { freq = f_s; integ = g_s; i0 = i_s; }
        }
        // (the start values come from memory: like the constants below they are "used" here, in front of the
        //  block loops, so that no s_waitcnt vmcnt for them ends up inside -- see load_math_consts)
        asm volatile("" : "+v"(freq), "+v"(integ), "+v"(i0));
        const double kp = P.kp, ki = P.ki;
        // (DMAW: role 1 runs the FLL and feeds nothing, role 2 feeds the ring and runs no FLL; wave-uniform)
        const bool compute = FLLON && (!DMAW || role == 1);
        dspm::MathK MK;
        if constexpr (FLLON) { if (compute) MK = load_math_consts(); }
        OC *o_sig = (OC *)P.sig; OS *o_freq = (OS *)P.freq; OS *o_ef = (OS *)P.ef;
        uint8_t *o_dec = P.dec;
        // dma_fll: this wave feeds the timing wave's ring.  It waits at the hand-over barriers a third of its
        // time (the timing wave is the longer chain), and issuing a DMA block costs the issuing wave ~70 cycles
        // per KiB, so the producer role moves here: with block b of the timing loop = queue blocks
        // [b k, (b + 1) k), k = T / TQ, the rows of block b + 1 are awaited (vmcnt over this wave's own stores
        // since the issue, counted at run time) and their guard rows mirrored BEFORE the barrier that ends
        // block b, and the rows of block b + 2 are issued right AFTER it -- the moments at which the timing
        // wave did both itself.  B0 hands over block 0.
        const bool dma_fll = ring && P.dma_fll != 0;
        const bool feeder = dma_fll && (!DMAW || role == 2);          // this wave feeds the ring
        const int n_fst = (o_sig ? 1 : 0) + (o_freq ? 1 : 0) + (o_ef ? 1 : 0) + (o_dec ? 1 : 0);
        const int kq = T / TQ;
        Tile<IN> tf;
        int landed = 0, nst = 0;
        if (feeder) {
            tile_init(tf, P, smem, tile, lane, 1.0);
            tf.prologue(sps, T, P.C);
            landed = tf.F;
            tf.issue(sps * 2 * T + P.C);
        }
        if (dma_fll) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // B0
        if (compute && VARIANT == DSP_VARIANT_GARDNER2) {       // recovered[0] = signal[0]
            const bool have = (lane_ok || pass) && len > 0;
            IN v = xt[src];
            cd r0; r0.re = (double)v.x * scale; r0.im = (double)v.y * scale;
            FllLane fo = fll_iter(MK, have, r0, i0, freq, integ, kp, ki);
            if (o_sig) st_cplx(o_sig + obase, fo.sr, fo.si);
            if (o_freq) st_val(o_freq + obase, fo.freq);
            if (o_ef) st_val(o_ef + obase, fo.ef);
            if (o_dec) st_b8(o_dec + obase, fo.quad);
            nst += n_fst;
        }
        int it = 0;
        prio_n = prio_n ? (prio_n * T / TQ > 0 ? prio_n * T / TQ : 1) : 0;     // same period in queue blocks
        prio_c = 1;
        constexpr bool SPEC = (SPECT && H == 0 && (VARIANT == DSP_VARIANT_GARDNER2 || VARIANT == DSP_VARIANT_TED));
        const int fmask = (o_sig ? 1 : 0) | (o_freq ? 2 : 0) | (o_ef ? 4 : 0) | (o_dec ? 8 : 0);
        const size_t tile_el = (size_t)tile * out_rows * 64;
        const SpecBasesF fbases = {o_sig ? (const unsigned char *)(o_sig + tile_el) : nullptr,
                                   o_freq ? (const unsigned char *)(o_freq + tile_el) : nullptr,
                                   o_ef ? (const unsigned char *)(o_ef + tile_el) : nullptr,
                                   o_dec ? (const unsigned char *)(o_dec + tile_el) : nullptr};
        const unsigned q_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)q_base + (unsigned)lane * 16u;
        for (int qb = 0;; qb++) {
            const int buf = qb & 1;
            if (prio_n && --prio_c == 0) { prio_c = prio_n; rotate_priority(slot, ++prio_ph); }
            DSP_TR(trp, 5);
            DSP_TT(tt_d);
            const bool dma_edge = feeder && (qb + 1) % kq == 0;     // this barrier ends a block of the timing loop
            if (dma_edge) {
                wait_vmcnt(nst);                                    // every DMA issued so far has landed
                tf.mirror(landed, tf.F);
                landed = tf.F;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            DSP_TT(tt_a);
            asm volatile("s_barrier" ::: "memory");
            DSP_TT(tt_c);
            if (dma_edge) {
                tf.issue(sps * ((qb + 1) / kq + 2) * T + P.C);
                nst = 0;
            }
            DSP_TR(trp, 0);
            const unsigned qa0 = q_lds + (unsigned)(buf * TQ) * 1024u;
            const int qfl = q_more[buf];
            if constexpr (!FLLON || DMAW) {                  // a wave that only feeds the ring (the timing loop on its own; DMAW's third wave)
                if (!compute) {
                    if (!(qfl & 1)) break;
                    continue;
                }
            }
            const d2v x0 = lds_ld<d2v>(qa0);                 // in flight while the flags decide the path
            const int more = qfl & 1;
            bool blk_done = false;
            if (SPEC && (qfl & 2) && it + TQ + ROFF < cap_rows &&
                (unsigned long long)(it + TQ + ROFF + 1) * 64ull * sizeof(OC) <= 0xffffffffull) {
                const unsigned qa = qa0;
                switch (fmask) {
                    case 15: blk_done = fll_spec_block<OS, OC, 15>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 11: blk_done = fll_spec_block<OS, OC, 11>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 9: blk_done = fll_spec_block<OS, OC, 9>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 10: blk_done = fll_spec_block<OS, OC, 10>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 8: blk_done = fll_spec_block<OS, OC, 8>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    case 0: blk_done = fll_spec_block<OS, OC, 0>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                    default: blk_done = fll_spec_block<OS, OC, -1>(MK, TQ, qa, x0, it + ROFF, i0, freq, integ, kp, ki, fbases, (unsigned)lane); break;
                }
                DSP_TR(trp, 12);
                DSP_TT(tt_b);
                nst += TQ * n_fst;                       // (a block that failed its vote is stored again below)
                if (blk_done) it += TQ;
            }
            if (!blk_done) {
                for (int t = 0; t < TQ; t++, it++) {
                    const int row = it + ROFF;
                    d2v qv = q_data[(buf * TQ + t) * 64 + lane];
                    // (a block the timing wave marked speculative carries no flags: every lane produced)
                    const bool pend = (qfl & 2) ? true : q_flag[(buf * TQ + t) * 64 + lane] != 0;
                    cd x; x.re = qv.x; x.im = qv.y;
                    DSP_TR(trp, 1);
                    if (row < cap_rows) {                                    // uniform
                        FllLane fo = fll_iter(MK, pend, x, (double)row + i0, freq, integ, kp, ki, trp);
                        const size_t off = obase + (size_t)row * 64;
                        if (o_sig) st_cplx(o_sig + off, fo.sr, fo.si);
                        if (o_freq) st_val(o_freq + off, fo.freq);
                        if (o_ef) st_val(o_ef + off, fo.ef);
                        if (o_dec) st_b8(o_dec + off, fo.quad);
                        nst += n_fst;
                    }
                    DSP_TR(trp, 5);
                }
            }
            if (!more) break;
        }
        if (compute && lane_ok && P.fll_out) { P.fll_out[2 * b] = freq; P.fll_out[2 * b + 1] = integ; }
    }
#ifdef DSP_TRACE
    if (P.trace && role < 2) {
        const unsigned long long tr_t1 = __builtin_amdgcn_s_memtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        if (lane == 0) {
            unsigned long long *t = P.trace + ((size_t)tile * 2 + role) * 20;
            t[0] = tr_t0; t[1] = tr_t1; t[2] = hw; t[3] = xcc;
            for (int i = 0; i < 16; i++) t[4 + i] = tr_acc.acc[i];
#if !defined(DSP_TRACE_STAMPS)
            t[4] = tt_a; t[5] = tt_b; t[6] = tt_c; t[7] = tt_d;     // advance / segments (blocks) / barrier / the rest
            t[8] = tt_e; t[9] = tt_f;                               // timing wave: block head -> segment, segment -> hand-over
#endif
        }
    }
#endif
}

#if DSP_SYNC_PART >= 1 && DSP_SYNC_PART <= 6
// The two-wave kernel with the Lagrange interpolator inside the timing loop, one translation unit per
// (variant, set of orders) so that they build side by side:
//   part 1  gardner2, half_len 1, 2, 4      part 2  gardner2, half_len 3, 6, 8      part 5  gardner2, half_len 5, 7
//   part 3  gardner_ted, half_len 1, 2, 4   part 4  gardner_ted, half_len 3, 6, 8   part 6  gardner_ted, half_len 5, 7
// gardner2: straight-line steady iteration (g2_steady_lagr), two waves per tile or four (two of them
// chosen by the CU ticket, see above).  gardner_ted: the per-lane iteration (ted_iter + lagrange_at) on
// the timing wave, the FLL on its own wave.
#if DSP_SYNC_PART == 1 || DSP_SYNC_PART == 3
#define DSP_LAGR_HA 1
#define DSP_LAGR_HB 2
#define DSP_LAGR_HC 4
#elif DSP_SYNC_PART == 2 || DSP_SYNC_PART == 4
#define DSP_LAGR_HA 3
#define DSP_LAGR_HB 6
#define DSP_LAGR_HC 8
#else
#define DSP_LAGR_HA 5
#define DSP_LAGR_HB 7
#define DSP_LAGR_HC 0
#endif
template <typename IN>
static hipError_t launch_split_lagr_t(const SyncParams &P, size_t lds, hipStream_t s)
{
    constexpr int VAR = (DSP_SYNC_PART <= 2 || DSP_SYNC_PART == 5) ? DSP_VARIANT_GARDNER2 : DSP_VARIANT_TED;
    const int ntiles = (P.B + 63) / 64;
    const bool four = VAR == DSP_VARIANT_GARDNER2 && P.spare_wave == 2 && P.cu_tickets;
    void (*k)(SyncParams) = nullptr;
    auto pick = [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        if constexpr (VAR == DSP_VARIANT_GARDNER2) {
            k = four ? sync_split_kernel<IN, VAR, true, 4, HH> : sync_split_kernel<IN, VAR, true, 2, HH>;
        } else {
            k = sync_split_kernel<IN, VAR, true, 2, HH>;
        }
    };
    switch (P.half_len) {
        case DSP_LAGR_HA: pick(std::integral_constant<int, DSP_LAGR_HA>{}); break;
        case DSP_LAGR_HB: pick(std::integral_constant<int, DSP_LAGR_HB>{}); break;
#if DSP_LAGR_HC
        case DSP_LAGR_HC: pick(std::integral_constant<int, DSP_LAGR_HC>{}); break;
#endif
        default: return hipErrorInvalidValue;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(ntiles), dim3(four ? 256 : 128), lds, s, P);
    return hipGetLastError();
}
#endif

#if DSP_SYNC_PART == 8
// Part 8: the fused kernel with three live waves per tile (sync_split_kernel<..., DMAW = true>, 192 threads):
// launches of at most one tile per CU.
template <typename IN>
static hipError_t launch_split_dmaw_t(const SyncParams &P, int variant, size_t lds, hipStream_t s)
{
    const int ntiles = (P.B + 63) / 64;
    void (*k)(SyncParams) =
        variant == DSP_VARIANT_GARDNER2
            ? (P.scale ? sync_split_kernel<IN, DSP_VARIANT_GARDNER2, true, 3, 0, DSP_SPECT_DEFAULT, true, true>
                       : sync_split_kernel<IN, DSP_VARIANT_GARDNER2, false, 3, 0, DSP_SPECT_DEFAULT, true, true>)
            : (P.scale ? sync_split_kernel<IN, DSP_VARIANT_TED, true, 3, 0, DSP_SPECT_DEFAULT, true, true>
                       : sync_split_kernel<IN, DSP_VARIANT_TED, false, 3, 0, DSP_SPECT_DEFAULT, true, true>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(ntiles), dim3(192), lds, s, P);
    return hipGetLastError();
}
#endif

#if DSP_SYNC_PART == 7
// Part 7: the two-wave kernel without its FLL (sync_split_kernel<..., FLLON = false>): linear interpolator,
// both timing loops, two waves per tile or four with the CU ticket (the three-wave launch, an option kept
// for A/B runs of the fused kernel, is served by the two-wave one here).
template <typename IN>
static hipError_t launch_split_nofll_t(const SyncParams &P, int variant, size_t lds, hipStream_t s)
{
    const int ntiles = (P.B + 63) / 64;
    const bool four = P.spare_wave == 2 && P.cu_tickets;
    void (*k)(SyncParams);
#define DSP_NOFLL_PICK(N)                                                                                              \
    k = variant == DSP_VARIANT_GARDNER2                                                                                \
            ? (P.scale ? sync_split_kernel<IN, DSP_VARIANT_GARDNER2, true, N, 0, DSP_SPECT_DEFAULT, false>             \
                       : sync_split_kernel<IN, DSP_VARIANT_GARDNER2, false, N, 0, DSP_SPECT_DEFAULT, false>)           \
            : (P.scale ? sync_split_kernel<IN, DSP_VARIANT_TED, true, N, 0, DSP_SPECT_DEFAULT, false>                  \
                       : sync_split_kernel<IN, DSP_VARIANT_TED, false, N, 0, DSP_SPECT_DEFAULT, false>)
    if (four) { DSP_NOFLL_PICK(4); } else { DSP_NOFLL_PICK(2); }
#undef DSP_NOFLL_PICK
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(ntiles), dim3(four ? 256 : 128), lds, s, P);
    return hipGetLastError();
}
#endif

}  // namespace

#define DSP_CAT2(a, b) a##b
#define DSP_CAT(a, b) DSP_CAT2(a, b)
#if DSP_SYNC_PART >= 1 && DSP_SYNC_PART <= 6
hipError_t DSP_CAT(launch_sync_split_lagr_part, DSP_SYNC_PART)(const SyncParams &P, bool c128, size_t lds, hipStream_t s)
{
    return c128 ? launch_split_lagr_t<d2v>(P, lds, s) : launch_split_lagr_t<f2v>(P, lds, s);
}
#elif DSP_SYNC_PART == 7
hipError_t launch_sync_split_nofll(const SyncParams &P, bool c128, int variant, size_t lds, hipStream_t s)
{
    return c128 ? launch_split_nofll_t<d2v>(P, variant, lds, s) : launch_split_nofll_t<f2v>(P, variant, lds, s);
}
#elif DSP_SYNC_PART == 8
hipError_t launch_sync_split_dmaw(const SyncParams &P, bool c128, int variant, size_t lds, hipStream_t s)
{
    return c128 ? launch_split_dmaw_t<d2v>(P, variant, lds, s) : launch_split_dmaw_t<f2v>(P, variant, lds, s);
}
#else
hipError_t launch_sync_split_nofll(const SyncParams &P, bool c128, int variant, size_t lds, hipStream_t s);
hipError_t launch_sync_split_dmaw(const SyncParams &P, bool c128, int variant, size_t lds, hipStream_t s);
hipError_t launch_sync_split_lagr_part1(const SyncParams &P, bool c128, size_t lds, hipStream_t s);
hipError_t launch_sync_split_lagr_part2(const SyncParams &P, bool c128, size_t lds, hipStream_t s);
hipError_t launch_sync_split_lagr_part3(const SyncParams &P, bool c128, size_t lds, hipStream_t s);
hipError_t launch_sync_split_lagr_part4(const SyncParams &P, bool c128, size_t lds, hipStream_t s);
hipError_t launch_sync_split_lagr_part5(const SyncParams &P, bool c128, size_t lds, hipStream_t s);
hipError_t launch_sync_split_lagr_part6(const SyncParams &P, bool c128, size_t lds, hipStream_t s);

// ---------------------------------------------------------------------------
// Ring schedule.  Iteration `it` of the timing loop touches rows
// [sps*(it-1) - h, sps*(it+2) + 2 + h); block b (T iterations) starts with rows
// < sps*(b+1)*T + C landed and issues rows < sps*(b+2)*T + C, which needs
// ring_rows >= sps*(2T+1) + C + h (+ slack for lanes that slipped a symbol).
SyncPlan sync_plan(int elem_bytes, int sps, int half_len, int variant, int n_store, int ring_rows_req)
{
    SyncPlan pl;
    int R = ring_rows_req > 0 ? ring_rows_req : 64;
    int rpp = 1024 / (64 * elem_bytes);
    int step = (variant == DSP_VARIANT_NONE) ? 1 : sps;
    int h = half_len;
    int C = step + 2 + h + 2;
    C += C & 1;
    int slack = (variant == DSP_VARIANT_NONE) ? 0 : step;
    int T = ((R - C - h - slack) / step - 1) / 2;
    if (T > 14) T = 14;                           // (8 until round 3: every block costs a DMA wait, an issue loop and a vote)
    if (n_store > 0) while (T > 1 && T * n_store > 63) T--;
    if ((step * T) % rpp) T -= 1;                 // whole DMA pieces per block
    pl.ring_rows = R;
    pl.C = C;
    pl.T = T;
    pl.use_ring = (T >= 1 && T * n_store <= 63 && step * (2 * T + 1) + C + h <= R) ? 1 : 0;
    if (!pl.use_ring) pl.T = 8;
    pl.nwait = pl.T * n_store;
    pl.lds_bytes = pl.use_ring ? R * 64 * elem_bytes : 0;
    return pl;
}

template <typename IN, int VARIANT, bool FLL>
static hipError_t launch_v(const SyncParams &P, size_t lds, hipStream_t s)
{
    int ntiles = (P.B + 63) / 64;
    if (P.half_len > 0)
        hipLaunchKernelGGL((sync_kernel<IN, VARIANT, FLL, true, true>), dim3(ntiles), dim3(64), lds, s, P);
    else if (P.scale)
        hipLaunchKernelGGL((sync_kernel<IN, VARIANT, FLL, false, true>), dim3(ntiles), dim3(64), lds, s, P);
    else
        hipLaunchKernelGGL((sync_kernel<IN, VARIANT, FLL, false, false>), dim3(ntiles), dim3(64), lds, s, P);
    return hipGetLastError();
}

template <typename IN>
static hipError_t launch_in(const SyncParams &P, int variant, bool fll, size_t lds, hipStream_t s)
{
    if (P.TQ > 0 && P.half_len > 0) { // ... with the Lagrange interpolator: the other translation units
        const bool set2 = P.half_len == 3 || P.half_len == 6 || P.half_len == 8, set3 = P.half_len == 5 || P.half_len == 7;
        if (variant == DSP_VARIANT_GARDNER2)
            return set3 ? launch_sync_split_lagr_part5(P, sizeof(IN) == 16, lds, s)
                 : set2 ? launch_sync_split_lagr_part2(P, sizeof(IN) == 16, lds, s)
                        : launch_sync_split_lagr_part1(P, sizeof(IN) == 16, lds, s);
        return set3 ? launch_sync_split_lagr_part6(P, sizeof(IN) == 16, lds, s)
             : set2 ? launch_sync_split_lagr_part4(P, sizeof(IN) == 16, lds, s)
                    : launch_sync_split_lagr_part3(P, sizeof(IN) == 16, lds, s);
    }
    if (P.TQ > 0 && !fll)  // the timing loop on its own on the two-wave kernel (part 7)
        return launch_sync_split_nofll(P, sizeof(IN) == 16, variant, lds, s);
    if (P.TQ > 0 && P.spare_wave == 3)   // three live waves per tile: timing, FLL, ring feeder (part 8)
        return launch_sync_split_dmaw(P, sizeof(IN) == 16, variant, lds, s);
    if (P.TQ > 0) {        // two waves per tile (timing wave + FLL wave)
        int ntiles = (P.B + 63) / 64;
        const int nwv = P.spare_wave == 2 && P.cu_tickets ? 4 : P.spare_wave ? 3 : 2;
        void (*k)(SyncParams);
#define DSP_SPLIT_PICK(N)                                                                                              \
    k = variant == DSP_VARIANT_GARDNER2                                                                                \
            ? (P.scale ? sync_split_kernel<IN, DSP_VARIANT_GARDNER2, true, N> : sync_split_kernel<IN, DSP_VARIANT_GARDNER2, false, N>) \
            : (P.scale ? sync_split_kernel<IN, DSP_VARIANT_TED, true, N> : sync_split_kernel<IN, DSP_VARIANT_TED, false, N>)
        if (nwv == 4) { DSP_SPLIT_PICK(4); } else if (nwv == 3) { DSP_SPLIT_PICK(3); } else { DSP_SPLIT_PICK(2); }
#undef DSP_SPLIT_PICK
        if (lds > 64 * 1024) {     // ring below 64 KiB, queue above it: has to be asked for
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, dim3(ntiles), dim3(64 * nwv), lds, s, P);
        return hipGetLastError();
    }
    switch (variant) {
        case DSP_VARIANT_GARDNER2:
            return fll ? launch_v<IN, DSP_VARIANT_GARDNER2, true>(P, lds, s)
                       : launch_v<IN, DSP_VARIANT_GARDNER2, false>(P, lds, s);
        case DSP_VARIANT_TED:
            return fll ? launch_v<IN, DSP_VARIANT_TED, true>(P, lds, s)
                       : launch_v<IN, DSP_VARIANT_TED, false>(P, lds, s);
        case DSP_VARIANT_NONE:
            return launch_v<IN, DSP_VARIANT_NONE, true>(P, lds, s);
    }
    return hipErrorInvalidValue;
}

// Two-wave tiles of the linear-interpolator kernel a CU can hold as far as its registers go: read from the
// code object (round 2's kernel held 133 VGPRs, the speculative segments raised it to 192, pinning the loads of
// the constants brought it back to 168: six, four, six tiles per CU -- a number that must not live in the host
// code as a constant; the LDS shares cap the useful number at four, dsp_api.hip sync_tiles_run).
int sync_split_tiles_per_cu(bool c128, int variant, bool scaled)
{
    void (*k)(SyncParams);
    if (c128) k = variant == DSP_VARIANT_GARDNER2 ? (scaled ? sync_split_kernel<d2v, DSP_VARIANT_GARDNER2, true, 2> : sync_split_kernel<d2v, DSP_VARIANT_GARDNER2, false, 2>)
                                                  : (scaled ? sync_split_kernel<d2v, DSP_VARIANT_TED, true, 2> : sync_split_kernel<d2v, DSP_VARIANT_TED, false, 2>);
    else k = variant == DSP_VARIANT_GARDNER2 ? (scaled ? sync_split_kernel<f2v, DSP_VARIANT_GARDNER2, true, 2> : sync_split_kernel<f2v, DSP_VARIANT_GARDNER2, false, 2>)
                                             : (scaled ? sync_split_kernel<f2v, DSP_VARIANT_TED, true, 2> : sync_split_kernel<f2v, DSP_VARIANT_TED, false, 2>);
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)k) != hipSuccess) { (void)hipGetLastError(); return 4; }
    const int regs = (fa.numRegs + 7) & ~7;                      // allocated in blocks of eight
    const int waves = regs > 0 ? 512 / regs : 8;                 // per SIMD, 512 VGPRs each
    const int tiles = (waves > 8 ? 8 : waves) * 4 / 2;           // four SIMDs, two waves per tile
    return tiles < 1 ? 1 : tiles;
}

hipError_t launch_sync(const SyncParams &P, bool c128, int variant, bool fll, size_t lds_bytes,
                       hipStream_t stream)
{
    if (P.B <= 0) return hipSuccess;
    return c128 ? launch_in<d2v>(P, variant, fll, lds_bytes, stream)
                : launch_in<f2v>(P, variant, fll, lds_bytes, stream);
}
#endif  // DSP_SYNC_PART

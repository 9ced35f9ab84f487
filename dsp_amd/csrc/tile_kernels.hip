// tile_kernels.hip -- burst-major [B][ld] <-> tiled [ntiles][rows][64] transposes.
//
// Host callers hand over burst-major arrays (one burst per row, as the
// reference's numpy arrays are laid out); the recurrence kernels want the
// interleaved tile layout (dsp_internal.h).  Both directions go through a
// 64x64 LDS tile so that global reads and writes are both row-contiguous.
#include "dsp_internal.h"

namespace {

template <typename E> struct Pad { enum { STRIDE = 64 + (sizeof(E) >= 8 ? 1 : (sizeof(E) == 4 ? 1 : 4)) }; };

// src[b][n] -> dst[(b/64)*rows + n][b%64]; zero fill beyond lens[b] / B.  With `starts`,
// burst b is the window of one long stream that begins at sample starts[b].
template <typename E>
__global__ __launch_bounds__(256) void tile_from_bursts_k(const E *__restrict__ src, int64_t ld,
                                                         const int *__restrict__ lens, int len_all,
                                                         int B, E *__restrict__ dst, int rows,
                                                         const int64_t *__restrict__ starts)
{
    __shared__ E t[64][Pad<E>::STRIDE];
    const int tile = blockIdx.y, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = 0; i < 16; i++) {
        int bb = w * 16 + i;
        int b = tile * 64 + bb;
        int n = n0 + lane;
        E v = E();
        if (b < B) {
            int len = lens ? lens[b] : len_all;
            if (n < len) v = src[(starts ? starts[b] : (int64_t)b * ld) + n];
        }
        t[bb][lane] = v;
    }
    __syncthreads();
    for (int i = 0; i < 16; i++) {
        int r = w * 16 + i;
        int n = n0 + r;
        if (n < rows) dst[((size_t)tile * rows + n) * 64 + lane] = t[lane][r];
    }
}

// dst[b][n] = src[(b/64)*rows + n][b%64] for n < ncopy
template <typename E>
__global__ __launch_bounds__(256) void bursts_from_tiles_k(const E *__restrict__ src, int rows, int B,
                                                          E *__restrict__ dst, int64_t ld, int ncopy)
{
    __shared__ E t[64][Pad<E>::STRIDE];
    const int tile = blockIdx.y, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = 0; i < 16; i++) {
        int r = w * 16 + i;
        int n = n0 + r;
        E v = E();
        if (n < rows) v = src[((size_t)tile * rows + n) * 64 + lane];
        t[r][lane] = v;
    }
    __syncthreads();
    for (int i = 0; i < 16; i++) {
        int bb = w * 16 + i;
        int b = tile * 64 + bb;
        int n = n0 + lane;
        if (b < B && n < ncopy) dst[(int64_t)b * ld + n] = t[lane][bb];
    }
}

// plain streaming copy, 16 B per thread and step (bench.py measures it beside the kernels: what
// the memory system of THIS box gives a 1:1 read/write stream)
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_k(const f4v *__restrict__ a, f4v *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

}  // namespace

hipError_t launch_stream_copy(const void *src, void *dst, size_t bytes, hipStream_t s)
{
    if (bytes < 16) return hipSuccess;
    // two float4 per thread: the best of the shapes tools/ubench/copy_bw.hip tries on this hardware
    // (5.3 TB/s read + write; 32768 workgroups looping longer reach 4.9, hipMemcpyDtoD 4.7)
    size_t wgs = (bytes / 16 + 511) / 512;
    if (wgs > 131072) wgs = 131072;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(stream_copy_k, dim3((unsigned)wgs), dim3(256), 0, s, (const f4v *)src, (f4v *)dst, bytes / 16);
    return hipGetLastError();
}

hipError_t launch_tile_from_bursts(const void *src, int64_t ld, const int *lens, int len_all, int B,
                                   void *dst, int rows, int elem_bytes, hipStream_t stream,
                                   const int64_t *starts)
{
    if (B <= 0 || rows <= 0) return hipSuccess;
    dim3 grid((rows + 63) / 64, (B + 63) / 64), block(256);
    switch (elem_bytes) {
        case 8:
            hipLaunchKernelGGL(tile_from_bursts_k<float2>, grid, block, 0, stream, (const float2 *)src, ld,
                               lens, len_all, B, (float2 *)dst, rows, starts);
            break;
        case 16:
            hipLaunchKernelGGL(tile_from_bursts_k<double2>, grid, block, 0, stream, (const double2 *)src,
                               ld, lens, len_all, B, (double2 *)dst, rows, starts);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bursts_from_tiles(const void *src, int rows, int B, void *dst, int64_t ld,
                                    int ncopy, int elem_bytes, hipStream_t stream)
{
    if (B <= 0 || ncopy <= 0) return hipSuccess;
    dim3 grid((ncopy + 63) / 64, (B + 63) / 64), block(256);
    switch (elem_bytes) {
        case 1:
            hipLaunchKernelGGL(bursts_from_tiles_k<uint8_t>, grid, block, 0, stream, (const uint8_t *)src,
                               rows, B, (uint8_t *)dst, ld, ncopy);
            break;
        case 4:
            hipLaunchKernelGGL(bursts_from_tiles_k<float>, grid, block, 0, stream, (const float *)src, rows,
                               B, (float *)dst, ld, ncopy);
            break;
        case 8:
            hipLaunchKernelGGL(bursts_from_tiles_k<float2>, grid, block, 0, stream, (const float2 *)src,
                               rows, B, (float2 *)dst, ld, ncopy);
            break;
        case 16:
            hipLaunchKernelGGL(bursts_from_tiles_k<double2>, grid, block, 0, stream, (const double2 *)src,
                               rows, B, (double2 *)dst, ld, ncopy);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// dsp_api.hip -- the C ABI of libdsp_hip.so (include/dsp_hip.h): context,
// device memory, and the host-side orchestration of the kernels.  No C++
// exception crosses the boundary; every entry point returns a status code.
#include "../../include/dsp_hip.h"
#include "dsp_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

struct dsp_ctx {
    int device;
    int n_cu;           // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    size_t lds_per_cu;  // LDS bytes of one CU = the most one workgroup can be given (160 KiB on gfx950)
    hipStream_t stream;
    bool own_stream;
    hipEvent_t ev0, ev1;
    int ring_rows;      // 0 = automatic
    int lds_pad;        // extra dynamic LDS per workgroup (bytes) to steer workgroups per CU
    int split;          // -1 auto / 1: two waves per tile where applicable; 0: one wave per tile
    int batch_split;    // 1 (default): a batch of more tiles than the CUs keep resident goes out as several launches
    int queue_block;    // split kernel: symbols per hand-off block, 0 = automatic
    int lead_copy;      // segment-parallel mode, warm-up rows: 0 = the filter's second store, 1 = seg_lead_copy_k, 2 = filtered per window
    int fuse_derot;     // dsp_rx_bursts: cut + derotation inside the matched filter's staging (1, default) or as a pass of its own (0)
    int prio_role;      // split kernel: timing waves above FLL waves (1), rotation / none (0), -1 = automatic
    int dma_fll;        // split kernel: the FLL wave issues the ring's DMA (1, default) or the timing wave itself (0)
    int guard;          // split kernel: guard rows behind the LDS ring where they fit (1, default) or never (0)
    int big_ring;       // split kernel, one tile per CU: ring of up to 128 KiB (1, default) or 64 KiB as in rounds 1-3 (0)
    int spec;           // split kernel, linear interpolator: speculative branch-free segments (1, default) or the per-symbol votes (0)
    int spare_wave;     // split kernel, idle waves for SIMD balance: -1 auto, 0 never, 1 three waves, 2 four waves + CU tickets
    unsigned *cu_tickets;   // [4096] per-CU workgroup counters for the four-wave launch
    unsigned *place;        // diagnostic (option "place_ptr"): caller's device buffer [tiles][2] for the live waves' HW_ID
    int corr_window;    // preamble search: 1 = register-window kernel where it fits (default), 0 = diff_corr_kernel
    int fir_stream;     // matched filter of <= 41 taps: 2 = persistent fir_strip_kernel where eligible (default),
                        // 1 = fir_stream_kernel, 0 = the tap-blocked kernel
    double *fir_taps;   // [64] the stream kernel's zero-padded taps
    void *static_scratch;   // this device's address of the static `scratch` array
    int prio_period;    // split kernel: priority rotation period in DMA blocks (power of two), 0 = off, -1 = automatic
    // grow-only device scratch kept between calls of the segment-parallel mode (its window tiles are
    // gigabytes: hipMalloc / hipFree per call cost milliseconds); released by dsp_trim / dsp_destroy
    void *scratch[8];
    size_t scratch_bytes[8];
    void *pinned;       // page-locked host block for the small results a call brings home (seg_info, counts)
    size_t pinned_bytes;
    // host link (link_h2d / link_d2h): page-locked ring that large transfers from / to the caller's pageable
    // arrays are staged through by `link_threads` host threads while the previous slot is on the link
    char *link_ring;
    hipEvent_t link_ev[4];
    bool link_busy[4];
    int link_threads;   // 0 = plain hipMemcpyAsync (the runtime's own staging), default 4
#ifdef DSP_TRACE
    unsigned long long *trace;
    int trace_stamps;
#endif
};

// Scratch of dsp_fir_tiles_static_dev, the one entry point that allocates nothing: partial statistics
// and padded taps of one matched-filter call.  Every GPU the library runs on holds its own copy of the
// array (at its own address) until the process ends.  In host code `scratch` is only a handle for
// hipGetSymbolAddress: dsp_create asks for the device address once per device and everything below
// uses that pointer (ctx->static_scratch), never the name.
#define DSP_STATIC_SCRATCH_BYTES (16u << 20)
static __device__ __align__(256) unsigned char scratch[DSP_STATIC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static std::map<int, void *> g_scratch_addr;   // device address of `scratch` per device ordinal

static thread_local char g_err[512] = "";

// A call that fails may return with work in flight on its device blocks (DevBuf); the blocks go back to
// the cache as the call unwinds, on this thread, so the mark is per thread: the first release after a
// failure waits for the block's own device.  Marked by every HIP failure, and by any other failure that
// happens while the thread holds blocks (a plain argument error at an entry point's head costs nothing).
static thread_local bool t_release_needs_sync = false;
static thread_local int t_live_blocks = 0;

static int fail(int code, const char *fmt, ...)
{
    if (code == DSP_E_HIP || t_live_blocks > 0) t_release_needs_sync = true;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(DSP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                \
    } while (0)

// RAII device buffer for the host-pointer entry points
// Device blocks of the host-pointer entry points are kept for the next call instead of going back to the
// driver: a single-burst call makes ~20 of them, and hipMalloc + hipFree (which waits for the device)
// cost it about a millisecond.  A block is handed out again only after the call that used it has
// returned, i.e. after that call synchronised its stream (an entry point that fails may return with work
// in flight: the next release then waits for the device first).  Bounded: 8 GiB and 128 blocks per
// process, blocks up to 2 GiB (of 288 GB: a call on 8192 x 8192-sample bursts makes 1.7 GiB of them, and
// hipMalloc + hipFree of its two 512 MiB blocks cost more than its kernels); dsp_trim gives everything back.  Contents are undefined, as hipMalloc's.
struct DevCache {
    struct Block { void *p; size_t n; int dev; };
    std::mutex mu;
    std::vector<Block> free_;
    size_t bytes = 0;
    hipError_t take(size_t n, void **p, size_t *cap, int *dev)
    {
        n = (n + 255) & ~(size_t)255;
        if (!n) n = 256;
        hipError_t e = hipGetDevice(dev);
        if (e != hipSuccess) return e;
        {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (int i = 0; i < (int)free_.size(); i++)
                if (free_[i].dev == *dev && free_[i].n >= n && free_[i].n <= 2 * n + 4096 &&
                    (best < 0 || free_[i].n < free_[best].n)) best = i;
            if (best >= 0) {
                *p = free_[best].p; *cap = free_[best].n;
                bytes -= free_[best].n;
                free_.erase(free_.begin() + best);
                return hipSuccess;
            }
        }
        *cap = n;
        e = hipMalloc(p, n);
        if (e == hipErrorOutOfMemory) {            // give the kept blocks back and try once more
            (void)hipGetLastError();
            clear(*dev);
            e = hipMalloc(p, n);
        }
        return e;
    }
    void give(void *p, size_t n, int dev)
    {
        if (t_release_needs_sync) {
            t_release_needs_sync = false;
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != dev) (void)hipSetDevice(dev);
            (void)hipDeviceSynchronize();
            if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
        }
        {
            std::lock_guard<std::mutex> g(mu);
            if (n <= ((size_t)2 << 30) && bytes + n <= ((size_t)8 << 30) && free_.size() < 128) {
                free_.push_back(Block{p, n, dev});
                bytes += n;
                return;
            }
        }
        (void)hipFree(p);
    }
    void clear(int dev)
    {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < free_.size();) {
            if (free_[i].dev == dev) {
                (void)hipFree(free_[i].p);
                bytes -= free_[i].n;
                free_.erase(free_.begin() + i);
            } else {
                i++;
            }
        }
    }
};
static DevCache *g_dev_cache = new DevCache;     // never destroyed: no hipFree after the runtime has shut down

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int dev = 0;
    ~DevBuf() { if (p) { g_dev_cache->give(p, cap, dev); t_live_blocks--; } }
    hipError_t alloc(size_t n)
    {
        hipError_t e = g_dev_cache->take(n, &p, &cap, &dev);
        if (e == hipSuccess && p) t_live_blocks++;
        return e;
    }
    template <typename T> T *as() { return (T *)p; }
};

struct ScratchRef {
    void *p = nullptr;
    template <typename T> T *as() { return (T *)p; }
};

struct Bump {                    // sub-allocation of one scratch block, 256-byte aligned
    char *base = nullptr;
    size_t off = 0, cap = 0;
    void *take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        void *p = base + off;
        off += n ? n : 16;
        return p;
    }
};

// a slot of the context's grow-only scratch, at least n bytes (contents undefined)
static hipError_t ctx_scratch(dsp_ctx *c, int slot, size_t n, void **p)
{
    if (c->scratch_bytes[slot] < n) {
        if (c->scratch[slot]) (void)hipFree(c->scratch[slot]);
        c->scratch[slot] = nullptr; c->scratch_bytes[slot] = 0;
        hipError_t e = hipMalloc(&c->scratch[slot], n);
        if (e == hipErrorOutOfMemory) {            // the blocks kept for the host-pointer entry points go back first
            (void)hipGetLastError();
            g_dev_cache->clear(c->device);
            e = hipMalloc(&c->scratch[slot], n);
        }
        if (e != hipSuccess) return e;
        c->scratch_bytes[slot] = n;
    }
    *p = c->scratch[slot];
    return hipSuccess;
}

// page-locked host memory kept in the context: a device-to-host copy into it is one DMA, where a copy into
// pageable memory is staged and blocks the host (0.1 ms for the 512 KB of a 2^30-sample capture's seg_info)
static hipError_t ctx_pinned(dsp_ctx *c, size_t n, void **p)
{
    if (c->pinned_bytes < n) {
        if (c->pinned) (void)hipHostFree(c->pinned);
        c->pinned = nullptr; c->pinned_bytes = 0;
        hipError_t e = hipHostMalloc(&c->pinned, n, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        c->pinned_bytes = n;
    }
    *p = c->pinned;
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// The host link.  The host-pointer entry points take the caller's arrays as they are -- numpy arrays in
// the reference's flows, i.e. pageable memory, which the runtime copies through its own staging buffer
// at 17-18 GB/s on this pool's boxes (tools/ubench/host_link.hip; page-locked memory: 57 GB/s).  A
// transfer of LINK_MIN bytes or more is therefore staged here: `link_threads` host threads copy slot k
// of a page-locked ring while slot k-1 is on the link (52 GB/s either way with four threads).  Arrays
// that are page-locked already (dsp_host_alloc, hipHostMalloc, hipHostRegister) go over the link directly.
//   link_h2d: on return the source has been read; the device copy is ordered on the context's stream.
//   link_d2h: ordered behind the stream's earlier work; a staged transfer is complete on return, a plain
//             one when the stream has been synchronised (every caller does that before it returns).
static const size_t LINK_SLOT = (size_t)16 << 20, LINK_MIN = (size_t)4 << 20;
static const int LINK_SLOTS = 4;

static bool host_is_page_locked(const void *p)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

static void link_par_copy(char *dst, const char *src, size_t n, int T)
{
    if (T <= 1 || n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
    const size_t piece = (n / T + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    size_t own_a = 0, own_b = piece < n ? piece : n;           // the calling thread takes the first piece
    for (int t = 1; t < T; t++) {
        const size_t a = (size_t)t * piece, b = a + piece > n ? n : a + piece;
        if (a >= n) break;
        try { th.emplace_back([=] { memcpy(dst + a, src + a, b - a); }); }
        catch (...) { memcpy(dst + a, src + a, b - a); }        // no thread to be had: copy it here
    }
    memcpy(dst + own_a, src + own_a, own_b - own_a);
    for (auto &x : th) x.join();
}

static hipError_t link_prepare(dsp_ctx *c)
{
    if (c->link_ring) return hipSuccess;
    hipError_t e = hipHostMalloc((void **)&c->link_ring, LINK_SLOT * LINK_SLOTS, hipHostMallocDefault);
    if (e != hipSuccess) { c->link_ring = nullptr; return e; }
    for (int i = 0; i < LINK_SLOTS; i++) {
        if (!c->link_ev[i]) e = hipEventCreateWithFlags(&c->link_ev[i], hipEventDisableTiming);
        if (e != hipSuccess) return e;
        c->link_busy[i] = false;
    }
    return hipSuccess;
}

static hipError_t link_slot_wait(dsp_ctx *c, int i)
{
    if (!c->link_busy[i]) return hipSuccess;
    c->link_busy[i] = false;
    return hipEventSynchronize(c->link_ev[i]);
}

static hipError_t link_h2d(dsp_ctx *c, void *dst, const void *src, size_t n)
{
    if (!n) return hipSuccess;
    if (n < LINK_MIN || c->link_threads <= 0 || host_is_page_locked(src) || link_prepare(c) != hipSuccess) {
        (void)hipGetLastError();
        return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, c->stream);
    }
    int k = 0;
    for (size_t off = 0; off < n; off += LINK_SLOT, k++) {
        const size_t m = off + LINK_SLOT > n ? n - off : LINK_SLOT;
        const int i = k % LINK_SLOTS;
        char *sl = c->link_ring + (size_t)i * LINK_SLOT;
        hipError_t e = link_slot_wait(c, i);
        if (e != hipSuccess) return e;
        link_par_copy(sl, (const char *)src + off, m, c->link_threads);
        e = hipMemcpyAsync((char *)dst + off, sl, m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->link_ev[i], c->stream);
        if (e != hipSuccess) return e;
        c->link_busy[i] = true;
    }
    return hipSuccess;
}

static hipError_t link_d2h(dsp_ctx *c, void *dst, const void *src, size_t n)
{
    if (!n) return hipSuccess;
    if (n < LINK_MIN || c->link_threads <= 0 || host_is_page_locked(dst) || link_prepare(c) != hipSuccess) {
        (void)hipGetLastError();
        return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->stream);
    }
    // slot k comes over the link while slot k-1 is copied out to the caller's array
    int k = 0;
    size_t prev_off = 0, prev_m = 0;
    for (size_t off = 0; off < n; off += LINK_SLOT, k++) {
        const size_t m = off + LINK_SLOT > n ? n - off : LINK_SLOT;
        const int i = k % LINK_SLOTS;
        hipError_t e = link_slot_wait(c, i);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(c->link_ring + (size_t)i * LINK_SLOT, (const char *)src + off, m, hipMemcpyDeviceToHost,
                           c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->link_ev[i], c->stream);
        if (e != hipSuccess) return e;
        c->link_busy[i] = true;
        if (k > 0) {
            const int j = (k - 1) % LINK_SLOTS;
            e = link_slot_wait(c, j);
            if (e != hipSuccess) return e;
            link_par_copy((char *)dst + prev_off, c->link_ring + (size_t)j * LINK_SLOT, prev_m, c->link_threads);
        }
        prev_off = off; prev_m = m;
    }
    const int j = (k - 1) % LINK_SLOTS;
    hipError_t e = link_slot_wait(c, j);
    if (e != hipSuccess) return e;
    link_par_copy((char *)dst + prev_off, c->link_ring + (size_t)j * LINK_SLOT, prev_m, c->link_threads);
    return hipSuccess;
}

static void ctx_trim(dsp_ctx *c)
{
    for (int i = 0; i < 8; i++) {
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
        c->scratch[i] = nullptr; c->scratch_bytes[i] = 0;
    }
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinned_bytes = 0;
    for (int i = 0; i < LINK_SLOTS; i++) (void)link_slot_wait(c, i);
    if (c->link_ring) (void)hipHostFree(c->link_ring);
    c->link_ring = nullptr;
}

extern "C" {

int dsp_version(void) { return 100; }
const char *dsp_last_error(void) { return g_err; }

int dsp_device_count(int *count)
{
    if (!count) return fail(DSP_E_ARG, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return DSP_OK;
}

int dsp_create(int device_id, dsp_ctx **out)
{
    if (!out) return fail(DSP_E_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail(DSP_E_ARG, "device %d not available (%d HIP devices visible)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    dsp_ctx *c = new dsp_ctx();
    c->device = device_id;
    c->own_stream = true;
    c->ring_rows = 0;
    c->lds_pad = -1;
    c->split = -1;
    c->batch_split = 1;
    c->queue_block = 0;
    c->prio_period = -1;
    c->spare_wave = -1;
    c->spec = 1;
    c->big_ring = 1;
    c->guard = 1;
    c->dma_fll = 1;
    c->prio_role = -1;
    c->fuse_derot = 1;
    c->lead_copy = 0;
    c->fir_stream = 2;
    c->corr_window = 1;
    c->fir_taps = nullptr;
#ifdef DSP_TRACE
    c->trace = nullptr;
    c->trace_stamps = 0;
#endif
    c->stream = nullptr; c->ev0 = nullptr; c->ev1 = nullptr; c->cu_tickets = nullptr; c->place = nullptr;
    c->link_ring = nullptr; c->link_threads = 4;
    for (int i = 0; i < LINK_SLOTS; i++) { c->link_ev[i] = nullptr; c->link_busy[i] = false; }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e == hipSuccess) {
        c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024 ? prop.maxSharedMemoryPerMultiProcessor
                                                                          : (size_t)64 * 1024;
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc((void **)&c->cu_tickets, 4096 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(c->cu_tickets, 0, 4096 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&c->fir_taps, 64 * sizeof(double));
    c->static_scratch = nullptr;
    if (e == hipSuccess) {
        // once per device, with that device current (hipSetDevice above)
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        auto it = g_scratch_addr.find(device_id);
        if (it == g_scratch_addr.end()) {
            void *p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e == hipSuccess) it = g_scratch_addr.emplace(device_id, p).first;
        }
        if (e == hipSuccess) c->static_scratch = it->second;
    }
    if (e != hipSuccess) {
        // release whatever was created before the step that failed
        if (c->fir_taps) (void)hipFree(c->fir_taps);
        if (c->cu_tickets) (void)hipFree(c->cu_tickets);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(DSP_E_HIP, "context creation failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return DSP_OK;
}

int dsp_device_info(dsp_ctx *c, int32_t *n_cu, int64_t *lds_per_cu)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (n_cu) *n_cu = c->n_cu;
    if (lds_per_cu) *lds_per_cu = (int64_t)c->lds_per_cu;
    return DSP_OK;
}

void dsp_destroy(dsp_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    ctx_trim(c);
    if (c->cu_tickets) (void)hipFree(c->cu_tickets);
    if (c->fir_taps) (void)hipFree(c->fir_taps);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < LINK_SLOTS; i++)
        if (c->link_ev[i]) (void)hipEventDestroy(c->link_ev[i]);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dsp_trim(dsp_ctx *c)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ctx_trim(c);
    g_dev_cache->clear(c->device);
    return DSP_OK;
}

int dsp_set_stream(dsp_ctx *c, void *s)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return DSP_OK;
}

int dsp_synchronize(dsp_ctx *c)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

int dsp_set_option(dsp_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return fail(DSP_E_ARG, "ctx/name is NULL");
    if (!strcmp(name, "ring_rows")) {
        if (value != 0 && (value < 16 || value > 256 || (value & (value - 1))))
            return fail(DSP_E_ARG, "ring_rows must be 0 or a power of two in [16,256]");
        c->ring_rows = (int)value;
    } else if (!strcmp(name, "lds_pad")) {
        c->lds_pad = (int)value;
    } else if (!strcmp(name, "split")) {
        c->split = (int)value;
    } else if (!strcmp(name, "batch_split")) {          // 0: never; 1: at the kernel's register-limited residency; n > 1: at n tiles per CU
        c->batch_split = (int)value;
#ifdef DSP_TRACE
    } else if (!strcmp(name, "trace_ptr")) {     // diagnostic build: device buffer for wave stamps
        c->trace = (unsigned long long *)(uintptr_t)value;
    } else if (!strcmp(name, "trace_stamps")) {
        c->trace_stamps = (int)value;
#endif
    } else if (!strcmp(name, "place_ptr")) {     // diagnostic: device buffer of 2 x tiles unsigned, 0 = off
        c->place = (unsigned *)(uintptr_t)value;
    } else if (!strcmp(name, "fir_stream")) {
        if (value < 0 || value > 2) return fail(DSP_E_ARG, "fir_stream must be 0, 1 or 2");
        c->fir_stream = (int)value;
    } else if (!strcmp(name, "corr_window")) {
        c->corr_window = value != 0;
    } else if (!strcmp(name, "lead_copy")) {
        if (value < 0 || value > 2) return fail(DSP_E_ARG, "lead_copy must be 0, 1 or 2");
        c->lead_copy = (int)value;
    } else if (!strcmp(name, "link_threads")) {
        if (value < 0 || value > 16) return fail(DSP_E_ARG, "link_threads must be in [0,16]");
        c->link_threads = (int)value;
    } else if (!strcmp(name, "fuse_derot")) {
        c->fuse_derot = value != 0;
    } else if (!strcmp(name, "prio_role")) {
        c->prio_role = (int)value;
    } else if (!strcmp(name, "dma_fll")) {
        c->dma_fll = value != 0;
    } else if (!strcmp(name, "guard")) {
        c->guard = value != 0;
    } else if (!strcmp(name, "spec")) {
        c->spec = value != 0;
    } else if (!strcmp(name, "big_ring")) {
        c->big_ring = value != 0;
    } else if (!strcmp(name, "spare_wave")) {
        if (value < -1 || value > 3) return fail(DSP_E_ARG, "spare_wave must be -1 (auto), 0, 1, 2 or 3");
        c->spare_wave = (int)value;
    } else if (!strcmp(name, "prio_period")) {
        if (value != -1 && (value < 0 || value > 4096 || (value & (value - 1))))
            return fail(DSP_E_ARG, "prio_period must be -1 (auto), 0 or a power of two");
        c->prio_period = (int)value;
    } else if (!strcmp(name, "queue_block")) {
        if (value < 0 || value > 64) return fail(DSP_E_ARG, "queue_block must be in [0,64]");
        c->queue_block = (int)value;
    } else {
        return fail(DSP_E_ARG, "unknown option '%s'", name);
    }
    return DSP_OK;
}

int dsp_malloc(dsp_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return fail(DSP_E_ARG, "ctx/dptr is NULL");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) {                // the blocks kept for the host-pointer entry points go back first
        (void)hipGetLastError();
        g_dev_cache->clear(c->device);
        e = hipMalloc(dptr, bytes ? bytes : 16);
    }
    if (e != hipSuccess) return fail(DSP_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return DSP_OK;
}
int dsp_free(dsp_ctx *c, void *dptr)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (dptr) HIP_TRY(hipFree(dptr));
    return DSP_OK;
}
int dsp_host_alloc(size_t bytes, void **hptr)
{
    if (!hptr) return fail(DSP_E_ARG, "hptr is NULL");
    *hptr = nullptr;
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DSP_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return DSP_OK;
}
int dsp_host_free(void *hptr)
{
    if (hptr) HIP_TRY(hipHostFree(hptr));
    return DSP_OK;
}
int dsp_memset(dsp_ctx *c, void *dptr, int value, size_t bytes)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, c->stream));
    return DSP_OK;
}
int dsp_h2d(dsp_ctx *c, void *dptr, const void *hptr, size_t bytes)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(link_h2d(c, dptr, hptr, bytes));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}
int dsp_d2h(dsp_ctx *c, void *hptr, const void *dptr, size_t bytes)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(link_d2h(c, hptr, dptr, bytes));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}
int dsp_stream_copy_dev(dsp_ctx *c, void *d_dst, const void *d_src, size_t bytes)
{
    if (!c || !d_dst || !d_src) return fail(DSP_E_ARG, "copy: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_stream_copy(d_src, d_dst, bytes, c->stream));
    return DSP_OK;
}

int dsp_timer_start(dsp_ctx *c)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return DSP_OK;
}
int dsp_timer_stop(dsp_ctx *c, float *ms)
{
    if (!c || !ms) return fail(DSP_E_ARG, "ctx/ms is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return DSP_OK;
}

// ---------------------------------------------------------------------------
static int64_t floordiv64(int64_t a, int64_t b)
{
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}

// rrc_filter -- read_file.py:5-36.  Same expression tree as the reference,
// evaluated in float64 with the host libm.
int dsp_rrc_taps(int sps, int span, double alpha, double *h, int64_t n_taps)
{
    if (sps <= 0 || span <= 0 || !(alpha > 0.0)) return fail(DSP_E_ARG, "rrc_taps: bad sps/span/alpha");
    int64_t lo = floordiv64(-(int64_t)span * sps, 2);
    int64_t n = floordiv64((int64_t)span * sps, 2) + 1 - lo;
    if (!h) return (int)n;
    if (n_taps != n) return fail(DSP_E_ARG, "rrc_taps: n_taps must be %lld", (long long)n);
    const double pi = 3.141592653589793;
    double ss = 0.0;
    for (int64_t i = 0; i < n; i++) {
        int64_t ti = lo + i;
        double t = (double)ti;
        double v;
        if (ti == 0) {
            v = 1 + alpha * (4 / pi - 1);
        } else if (fabs(t) == sps / (4 * alpha)) {
            v = (alpha / sqrt(2.0)) *
                ((1 + 2 / pi) * sin(pi / (4 * alpha)) + (1 - 2 / pi) * cos(pi / (4 * alpha)));
        } else {
            double a = pi * t / sps;
            double q = 4 * alpha * t / sps;
            v = (sin(a * (1 - alpha)) + q * cos(a * (1 + alpha))) / (a * (1 - q * q));
        }
        h[i] = v;
    }
    // unit energy: h / sqrt(sum h^2).  Summed pairwise in blocks of 8 like the
    // reference's np.sum so short filters normalise to the same bits.
    {
        std::vector<double> sq((size_t)n);
        for (int64_t i = 0; i < n; i++) sq[(size_t)i] = h[i] * h[i];
        struct PW {
            static double run(const double *a, int64_t m)
            {
                if (m < 8) { double r = 0.0; for (int64_t i = 0; i < m; i++) r += a[i]; return r; }
                if (m <= 128) {
                    double r[8]; int64_t i;
                    for (int j = 0; j < 8; j++) r[j] = a[j];
                    for (i = 8; i < m - (m % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
                    double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                    for (; i < m; i++) s += a[i];
                    return s;
                }
                int64_t m2 = m / 2; m2 -= m2 % 8;
                return run(a, m2) + run(a + m2, m - m2);
            }
        };
        ss = PW::run(sq.data(), n);
    }
    double nrm = sqrt(ss);
    for (int64_t i = 0; i < n; i++) h[i] = h[i] / nrm;
    return DSP_OK;
}

int dsp_fll_gains(double Bn, double y, double *kp, double *ki)
{
    if (!kp || !ki) return fail(DSP_E_ARG, "kp/ki is NULL");
    double d = y + 1 / (4 * y);
    *kp = 4 * Bn / 1000 / d;                  // fll_func.py:5
    double q = Bn / 1000 / d;
    *ki = 4 * pow(q, 2.0);                    // fll_func.py:6
    return DSP_OK;
}

int64_t dsp_tile_rows(int64_t N) { return N <= 0 ? 2 : ((N + 1) / 2) * 2; }

// ---------------------------------------------------------------------------
static int count_nonnull(std::initializer_list<const void *> l)
{
    int n = 0;
    for (const void *p : l) n += p ? 1 : 0;
    return n;
}

// x_tile_rows / x_tile0: SyncParams (segment-parallel mode lays its windows out overlapping; 0 / NULL for
// the C-ABI's separate tiles)
// idle_pass: SyncParams (the caller never reads the output columns of bursts of length 0)
static int sync_tiles_run(dsp_ctx *c, const dsp_sync_args *a, int x_tile_rows, const void *x_tile0, int idle_pass);

int dsp_sync_tiles_dev(dsp_ctx *c, const dsp_sync_args *a) { return sync_tiles_run(c, a, 0, nullptr, 0); }

static int sync_tiles_run(dsp_ctx *c, const dsp_sync_args *a, int x_tile_rows, const void *x_tile0, int idle_pass)
{
    if (!c || !a) return fail(DSP_E_ARG, "ctx/args is NULL");
    if (a->B < 0 || a->B > (1 << 30)) return fail(DSP_E_ARG, "sync: bad B");
    if (a->B == 0) return DSP_OK;
    if (!a->x) return fail(DSP_E_ARG, "sync: x is NULL");
    if (a->dtype != DSP_C64 && a->dtype != DSP_C128) return fail(DSP_E_ARG, "sync: bad dtype");
    if (a->variant < 0 || a->variant > 2) return fail(DSP_E_ARG, "sync: bad variant");
    // The timing loop alone with a Lagrange interpolator (do_fll = 0, interp_half_len 1 ... 8): the two-wave Lagrange
    // kernels exist only with their FLL wave, and the one-wave kernel runs such a call on its per-lane path (5-10 x
    // slower).  The call therefore goes out on the two-wave kernel with an idle FLL: zero gains, no FLL outputs, no
    // state in or out -- the timing wave's outputs are those of any other launch of it.
    const bool lagr_alone = a->do_fll == 0 && a->variant != 2 && a->interp_half_len >= 1 && a->interp_half_len <= 8 &&
                            c && c->spec != 0 && c->split != 0;
    const bool fll = a->do_fll != 0 || a->variant == 2 || lagr_alone;
    const int sps = a->variant == 2 ? 1 : a->sps;
    if (sps < 1 || sps > 1024) return fail(DSP_E_ARG, "sync: sps out of range");
    if (a->N < 0 || a->N >= (1LL << 30)) return fail(DSP_E_ARG, "sync: N out of range");
    if (a->in_rows < a->N || a->in_rows < 2 || (a->in_rows & 1))
        return fail(DSP_E_ARG, "sync: in_rows must be even and >= N (use dsp_tile_rows)");
    if (a->out_rows < 2 || a->out_rows >= (1LL << 30)) return fail(DSP_E_ARG, "sync: bad out_rows");
    if (a->interp_half_len < 0 || a->interp_half_len > 8)
        return fail(DSP_E_ARG, "sync: interp_half_len must be in [0,8]");
    if (fll && !lagr_alone && !(a->y > 0.0)) return fail(DSP_E_ARG, "sync: y must be > 0");

    // More tiles than a CU keeps resident (as far as the kernel's registers go, read from the code object --
    // the Lagrange kernels hold up to 281 -- and at most four, see below): the surplus would run as a second, poorly filled
    // round (98 304 bursts: 4.4 ms against 2 x 1.4).  Such a batch goes out as equal launches of whole tiles
    // instead, each of at most that many tiles per CU; the bursts are independent, so only the pointers move.
    {
        const int64_t ntiles_all = (a->B + 63) / 64;
        // ... and at most four: beyond that a tile's LDS share (160 KiB / tiles) no longer holds the 64-row ring
        // and its queue, and the small-ring shape is the slower one whatever the registers allow (the kernel has
        // needed 168 VGPRs = six tiles per CU since the constants' loads are pinned, sync_kernels.hip
        // load_math_consts; one launch of five tiles per CU: 4.07 ms against 2.40 as 3 + 2; six: 2.84 / 2.78;
        // twelve as 6 + 6: 5.52 / 5.28 as 4 + 4 + 4 -- profiles/r03_s3_ab_experiments.txt)
        int per_cu_max = c->batch_split > 1 ? c->batch_split
                       : (fll && a->variant != 2 && a->interp_half_len == 0)
                             ? sync_split_tiles_per_cu(a->dtype == DSP_C128, a->variant, a->scale != nullptr) : 4;
        if (c->batch_split == 1 && per_cu_max > 4) per_cu_max = 4;
        const int64_t resident = (int64_t)c->n_cu * per_cu_max;
        if (ntiles_all > resident && c->batch_split != 0) {
            const int64_t nlaunch = (ntiles_all + resident - 1) / resident;
            int64_t per = (ntiles_all + nlaunch - 1) / nlaunch;                   // tiles per launch ...
            per = (per + c->n_cu - 1) / c->n_cu * c->n_cu;                       // ... a whole number per CU
            const int64_t es = a->dtype == DSP_C128 ? 8 : 4, ec = 2 * es;       // output scalar / complex bytes
            const int64_t x_rows = x_tile_rows > 0 ? x_tile_rows : a->in_rows;
            for (int64_t t0 = 0; t0 < ntiles_all; t0 += per) {
                const int64_t b0 = t0 * 64;
                dsp_sync_args s = *a;
                s.B = a->B - b0 < per * 64 ? a->B - b0 : per * 64;
                s.x = (const char *)a->x + (ptrdiff_t)t0 * x_rows * 64 * ec;
                if (a->lens) s.lens = a->lens + b0;
                if (a->scale) s.scale = a->scale + b0;
                if (a->i0) s.i0 = a->i0 + b0;
                if (a->mu0_per) s.mu0_per = a->mu0_per + b0;
                if (a->fll_state_in) s.fll_state_in = a->fll_state_in + 2 * b0;
                if (a->fll_state_out) s.fll_state_out = a->fll_state_out + 2 * b0;
                if (a->n_out) s.n_out = a->n_out + b0;
                if (a->status) s.status = a->status + b0;
                const int64_t orow = t0 * a->out_rows * 64;                       // first output element of tile t0
                if (a->rec) s.rec = (char *)a->rec + orow * ec;
                if (a->sig) s.sig = (char *)a->sig + orow * ec;
                if (a->err) s.err = (char *)a->err + orow * es;
                if (a->mu) s.mu = (char *)a->mu + orow * es;
                if (a->pos) s.pos = (char *)a->pos + orow * es;
                if (a->freq) s.freq = (char *)a->freq + orow * es;
                if (a->ef) s.ef = (char *)a->ef + orow * es;
                if (a->dec) s.dec = a->dec + orow;
                const int rc = sync_tiles_run(c, &s, x_tile_rows, t0 == 0 ? x_tile0 : nullptr, idle_pass);
                if (rc != DSP_OK) return rc;
            }
            return DSP_OK;
        }
    }

    SyncParams P;
    memset(&P, 0, sizeof(P));
    P.x = a->x; P.lens = a->lens; P.scale = a->scale;
    P.B = (int)a->B; P.in_rows = (int)a->in_rows; P.len_all = (int)a->N; P.sps = sps;
    P.x_tile_rows = x_tile_rows > 0 ? x_tile_rows : P.in_rows; P.x_tile0 = x_tile0; P.idle_pass = idle_pass;
    P.alpha = a->alpha; P.mu0 = a->mu0; P.mu0_per = a->variant == DSP_GARDNER2 ? a->mu0_per : nullptr; P.half_len = a->variant == 2 ? 0 : a->interp_half_len;
    if (fll && !lagr_alone) dsp_fll_gains(a->Bn, a->y, &P.kp, &P.ki);
    P.i0 = a->i0; P.fll_in = lagr_alone ? nullptr : a->fll_state_in; P.fll_out = lagr_alone ? nullptr : a->fll_state_out;
    P.rec = a->variant == 2 ? nullptr : a->rec;
    P.terr = a->variant == 2 ? nullptr : a->err;
    P.mu = a->variant == 2 ? nullptr : a->mu;
    P.pos = a->variant == 2 ? nullptr : a->pos;
    const bool fll_out = fll && !lagr_alone;
    P.sig = fll_out ? a->sig : nullptr; P.freq = fll_out ? a->freq : nullptr; P.ef = fll_out ? a->ef : nullptr;
    P.dec = fll_out ? a->dec : nullptr;
    P.out_rows = (int)a->out_rows; P.n_out = a->n_out; P.status = a->status;
#ifdef DSP_TRACE
    P.trace = c->trace;
    P.trace_stamps = c->trace_stamps;
#endif

    // ---- kernel shape -------------------------------------------------------
    // One workgroup per tile; its dynamic LDS (input ring [+ hand-off queue]) also sets how
    // many tiles a CU holds (160 KiB per CU), so size it for an even spread over 256 CUs.
    const int elem = a->dtype == DSP_C128 ? 16 : 8;
    const int ntiles = (P.B + 63) / 64;
    // two waves per tile: both loops run, and the interpolator is the reference's linear one or a
    // Lagrange order (half_len 1 ... 8: sync_kernels.hip parts 1-6)
    const bool lagr_built = P.half_len >= 1 && P.half_len <= 8;
    // ... or the timing loop alone with the linear interpolator: the two-wave kernel without its FLL (the second
    // wave feeds the ring), sync_kernels.hip part 7; "spec" = 0 keeps the one-wave kernel for A/B runs
    const bool timing_alone = !fll && a->variant != 2 && P.half_len == 0 && c->spec != 0;
    const bool can_split = (fll && a->variant != 2 && (P.half_len == 0 || lagr_built)) || timing_alone;
    const int n_store_t = count_nonnull({P.rec, P.terr, P.mu, P.pos});
    const int n_store_f = count_nonnull({P.sig, P.freq, P.ef, P.dec});
    int per_cu = (ntiles + c->n_cu - 1) / c->n_cu;
    if (per_cu < 1) per_cu = 1;
    const size_t queue_min = (size_t)2 * 64 * 17 + 16;               // hand-off queue with TQ = 1
    // Residency heuristic: the LDS share that lets every tile of the launch be resident at once.  It
    // only steers the ring / queue sizes: with more tiles than that allows (B > ~160 k bursts) the
    // share stops at the smallest two-wave shape (32-row ring + one-symbol queue) and the surplus
    // tiles simply queue on the CUs.  The hard limit is the workgroup's LDS maximum, checked below.
    size_t budget_all = (c->lds_per_cu / per_cu) & ~(size_t)2047;   // LDS is allocated in coarse blocks
    {
        const size_t least = ((size_t)32 * 64 * elem + queue_min + 32 + 2047) & ~(size_t)2047;
        if (budget_all < least) budget_all = least;
        if (budget_all > c->lds_per_cu) budget_all = c->lds_per_cu;
    }
    // the input ring stays below 64 KiB where tiles share a CU (rounds 1-3 took that for M0's reach; it is
    // not, see ring_cap below); the two-wave kernel's hand-off queue may lie above it
    const size_t budget = budget_all > 64 * 1024 ? 64 * 1024 : budget_all;
    size_t lds = 0;
    bool split = can_split && c->split != 0;
    for (;;) {
        const int n_store = split ? n_store_t : n_store_t + n_store_f;
        // The LDS-DMA's destination base (M0) reaches the whole LDS of a workgroup, not 64 KiB of it
        // (tools/ubench/lds_dma_range.hip: a piece sent to base + 162 816 lands there), so where a two-wave tile has
        // the CU's LDS to itself the ring may take 128 KiB: 128 rows of complex128 -- 11 to 14 symbols per DMA
        // block instead of 6, as the complex64 ring has always had (complex64 at 64 rows: 1.00 ms against 0.82)
        const size_t ring_cap = (split && c->big_ring != 0 &&
                                 budget_all >= (size_t)(128 + 8) * 1024 + (size_t)2 * 7 * 64 * 17 + 64)
                                    ? (size_t)128 * 1024 : (size_t)64 * 1024;
        int R = c->ring_rows;
        if (R == 0) {
            // 128 rows (8 symbols per DMA block instead of 6: 2 % at up to two tiles per CU) wherever the
            // share allows; the loop below comes back to 64 / 32 rows where it does not
            R = 128;
            // prefer a smaller ring when that lets every tile be resident at once
            // (the ring alone has to stay inside the 64 KiB the LDS-DMA reaches; ring + queue inside the
            //  tile's share -- a complex128 ring of 64 rows fills the 64 KiB and its queue lies above it:
            //  a single complex128 burst ran 14 % slower on the 32-row ring this loop used to fall to)
            while (R > 32 && ((size_t)R * 64 * elem > ring_cap ||
                              (size_t)R * 64 * elem + (split ? 2200 : 0) > budget_all) &&
                   sync_plan(elem, sps, P.half_len, a->variant, n_store, R / 2).use_ring)
                R /= 2;
        }
        if ((size_t)R * 64 * elem > ring_cap) R = (int)(ring_cap / (64 * elem));
        SyncPlan pl = sync_plan(elem, sps, P.half_len, a->variant, n_store, R);
        // a ring that leaves no room for the smallest queue runs as one wave per tile instead
        if (split && (size_t)pl.lds_bytes + queue_min > budget_all) { split = false; continue; }
        P.use_ring = pl.use_ring; P.ring_rows = pl.ring_rows; P.T = pl.T; P.C = pl.C; P.nwait = pl.nwait;
        lds = (size_t)pl.lds_bytes;
        break;
    }
    P.TQ = 0;
    P.spec = c->spec;               // (the one-wave kernel's FLL-only block path obeys the same A/B switch)
    if (split) {
        // queue block: as many symbols as the double buffer can hold inside the budget
        const size_t room = (lds + queue_min > budget ? budget_all : budget) - 32;
        // Guard rows behind the ring (sync_kernels.hip Tile::mirror) where the LDS share has room for them and
        // a queue of at least four symbols: 8 rows (4 KiB complex64) buy one LDS address per symbol instead of
        // eight.  At four tiles per CU (40 KiB each: 32 KiB ring + three-symbol queue) they would leave the
        // queue one symbol, so that shape keeps the masked addresses.
        P.guard_rows = 0;
        // (speculative segments: gardner2, and gardner_ted at even samples-per-symbol counts)
        const bool spec_variant = a->variant == DSP_GARDNER2 || (a->variant == DSP_GARDNER_TED && sps >= 2 && (sps & 1) == 0);
        if (c->guard != 0 && c->spec && P.use_ring && spec_variant && P.half_len == 0 && sps + 2 <= 8) {
            const size_t gbytes = (size_t)8 * 64 * elem;
            if (lds + gbytes + (size_t)2 * 4 * 64 * 17 + 16 <= room) { P.guard_rows = 8; lds += gbytes; }
        }
        int TQ = c->queue_block;
        if (TQ == 0) {
            TQ = 1;
            for (int d = 16; d >= 1; d--)
                if (lds + (size_t)2 * d * 64 * 17 + 16 <= room) { TQ = d; break; }
        }
        while (TQ > 1 && lds + (size_t)2 * TQ * 64 * 17 + 16 > room) TQ--;
        // Speculative segments (sync_kernels.hip, gardner2 + linear interpolator) end with the DMA block or
        // the queue block, whichever comes first, and every segment costs a vote and a hand-over: keep the
        // two block lengths multiples of one another and the shorter one as long as the LDS share allows
        // (16384 bursts: T 8 / TQ 7 -> segments of 7, 1, 6, 2 ... symbols, 1.03 ms; T 7 / TQ 7: 0.99 ms).
        if (c->queue_block == 0 && c->spec && P.use_ring && spec_variant && P.half_len == 0) {
            const int rpp = 1024 / (64 * elem);
            int best_n = 0, best_T = P.T, best_TQ = TQ;
            for (int tq = TQ; tq >= 1; tq--)
                for (int t = P.T; t >= 1; t--) {
                    if ((sps * t) % rpp) continue;                         // whole DMA pieces per block
                    if (t % tq && tq % t) continue;
                    const int n = t < tq ? t : tq;
                    if (n > best_n || (n == best_n && t > best_T)) { best_n = n; best_T = t; best_TQ = tq; }
                }
            if (best_n > 0) {
                TQ = best_TQ;
                if (best_T != P.T) { P.T = best_T; P.nwait = best_T * n_store_t; }
            }
        }
        P.TQ = TQ;
        P.dma_fll = (c->dma_fll != 0 && c->spec && P.use_ring && spec_variant && P.half_len == 0 &&
                     P.T % TQ == 0) ? 1 : 0;
        // From three tiles per CU on two or more waves share a SIMD and the arbiter's oldest-first choice lets
        // some run ahead while others starve; rotating the four priority levels over the wave slots every 8
        // DMA blocks evens that out (65 536 bursts 1.93 -> 1.87 ms, 98 304 2.91 -> 2.52 ms; nothing at up
        // to two tiles per CU, where every wave has a SIMD to itself).  Scheduling only: same bits.
        P.prio_period = c->prio_period >= 0 ? c->prio_period : (per_cu >= 3 ? 8 : 0);
        P.prio_role = c->prio_role > 0 ? 1 : 0;
        if (P.prio_role) P.prio_period = 0;
        // two 2-wave workgroups per CU land on three SIMDs (sync_kernels.hip); the idle middle wave
        // spreads them over four.  Measured: 512 tiles 1.57 -> 1.30 ms; no gain at one workgroup per
        // CU, 3 % slower at three or four, and at eight the spare waves cost occupancy
        // At most one tile per CU: a third LIVE wave that feeds the ring in the FLL wave's place (sync_kernels.hip
        // part 8; the FLL wave is the longer chain there, and a symbol of complex128 at 4 sps is 4 KiB of DMA to issue)
        const bool feeder_ok = fll && a->variant != 2 && P.half_len == 0 && P.dma_fll != 0;
        P.spare_wave = c->spare_wave >= 0 ? c->spare_wave : (per_cu == 2 ? 2 : (per_cu == 1 && feeder_ok ? 3 : 0));
        if (P.spare_wave == 3 && !feeder_ok) P.spare_wave = 0;
        P.cu_tickets = c->cu_tickets;
        P.place = c->place;
        P.spec = c->spec;
        lds += (size_t)2 * TQ * 64 * 17 + 16 + 32;       // + the four-wave launch's five selection words
    }
    if (c->lds_pad >= 0) lds += (size_t)c->lds_pad;
    else if (budget > lds) lds = budget;
    if (lds > c->lds_per_cu)
        return fail(DSP_E_ARG, "sync: %zu bytes of LDS needed, a workgroup can have %zu", lds, c->lds_per_cu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_sync(P, a->dtype == DSP_C128, a->variant, fll, lds, c->stream));
    return DSP_OK;
}

int dsp_tile_bursts_dev(dsp_ctx *c, const void *d_src, int dtype, int64_t B, int64_t N,
                        int64_t src_stride, const int32_t *d_lens, void *d_tiles, int64_t rows)
{
    if (!c || !d_src || !d_tiles) return fail(DSP_E_ARG, "tile: NULL argument");
    if (rows < N || B < 0) return fail(DSP_E_ARG, "tile: rows < N");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_tile_from_bursts(d_src, src_stride, d_lens, (int)N, (int)B, d_tiles, (int)rows,
                                    dtype == DSP_C128 ? 16 : 8, c->stream));
    return DSP_OK;
}

int dsp_untile_dev(dsp_ctx *c, const void *d_tiles, int elem_bytes, int64_t rows, int64_t B,
                   void *d_dst, int64_t dst_stride, int64_t ncopy)
{
    if (!c || !d_tiles || !d_dst) return fail(DSP_E_ARG, "untile: NULL argument");
    if (ncopy > rows) ncopy = rows;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_bursts_from_tiles(d_tiles, (int)rows, (int)B, d_dst, dst_stride, (int)ncopy,
                                     elem_bytes, c->stream));
    return DSP_OK;
}

// ---------------------------------------------------------------------------
// Host-pointer orchestration shared by dsp_gardner / dsp_fll / dsp_sync_chain / dsp_rx_chain:
// runs the recurrence kernel on device tiles and brings the requested outputs back as
// burst-major host arrays.
static int sync_to_host(dsp_ctx *c, const void *d_xt, int dtype, int64_t B, int64_t N, int64_t rows,
                        const int32_t *d_lens, const double *d_scale, int sps, double alpha, double mu0,
                        int variant, int half_len, bool fll, double Bn, double y, const double *i0,
                        const double *st_in, double *st_out, void *rec, void *err, void *mu, void *sig,
                        void *freq, void *ef, uint8_t *dec, int32_t *n_out, int64_t out_stride)
{
    const int ce = dtype == DSP_C128 ? 16 : 8, re = ce / 2;
    const int64_t ntiles = (B + 63) / 64;
    const int64_t out_rows = out_stride + 1;      // + the kernels' scratch row
    hipStream_t s = c->stream;
    DevBuf d_i0, d_stin, d_stout, d_nout, d_status, d_stage;
    if (i0) {
        HIP_TRY(d_i0.alloc((size_t)B * 8));
        HIP_TRY(hipMemcpyAsync(d_i0.p, i0, (size_t)B * 8, hipMemcpyHostToDevice, s));
    }
    if (st_in) {
        HIP_TRY(d_stin.alloc((size_t)B * 16));
        HIP_TRY(hipMemcpyAsync(d_stin.p, st_in, (size_t)B * 16, hipMemcpyHostToDevice, s));
    }
    if (st_out) HIP_TRY(d_stout.alloc((size_t)B * 16));
    HIP_TRY(d_nout.alloc((size_t)B * 4));
    HIP_TRY(d_status.alloc((size_t)B * 4));

    struct OutSpec { void *host; int elem; DevBuf tiles; };
    OutSpec outs[7] = {{rec, ce, {}}, {err, re, {}}, {mu, re, {}}, {sig, ce, {}},
                       {freq, re, {}}, {ef, re, {}}, {(void *)dec, 1, {}}};
    size_t max_elem = 1;
    for (auto &o : outs)
        if (o.host) {
            HIP_TRY(o.tiles.alloc((size_t)ntiles * out_rows * 64 * o.elem));
            HIP_TRY(hipMemsetAsync(o.tiles.p, 0, (size_t)ntiles * out_rows * 64 * o.elem, s));
            if ((size_t)o.elem > max_elem) max_elem = o.elem;
        }

    dsp_sync_args a;
    memset(&a, 0, sizeof(a));
    a.x = d_xt; a.dtype = dtype; a.variant = variant; a.B = B; a.N = N; a.in_rows = rows;
    a.lens = d_lens; a.scale = d_scale;
    a.sps = sps; a.interp_half_len = half_len; a.alpha = alpha; a.mu0 = mu0;
    a.do_fll = fll ? 1 : 0; a.Bn = Bn; a.y = y;
    a.i0 = d_i0.as<double>(); a.fll_state_in = d_stin.as<double>(); a.fll_state_out = d_stout.as<double>();
    a.rec = outs[0].tiles.p; a.err = outs[1].tiles.p; a.mu = outs[2].tiles.p; a.sig = outs[3].tiles.p;
    a.freq = outs[4].tiles.p; a.ef = outs[5].tiles.p; a.dec = (uint8_t *)outs[6].tiles.p;
    a.out_rows = out_rows; a.n_out = d_nout.as<int32_t>(); a.status = d_status.as<int32_t>();
    int rc = dsp_sync_tiles_dev(c, &a);
    if (rc != DSP_OK) return rc;

    std::vector<int32_t> status((size_t)B);
    std::vector<int32_t> nout_tmp;
    int32_t *nout_h = n_out;
    if (!nout_h) { nout_tmp.resize((size_t)B); nout_h = nout_tmp.data(); }
    // tiles -> burst-major staging -> host
    auto up256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
    size_t all_out = 0;
    for (auto &o : outs)
        if (o.host) all_out += up256((size_t)B * out_stride * o.elem);
    const size_t off_nout = all_out, off_status = off_nout + up256((size_t)B * 4),
                 off_st = off_status + up256((size_t)B * 4), all_bytes = off_st + up256((size_t)B * 16);
    if (all_bytes <= ((size_t)4 << 20)) {
        // A small call (one burst, a few): every output is brought to burst-major order side by side, comes home
        // into the context's page-locked block with asynchronous copies, and the call waits ONCE -- a copy into
        // the caller's pageable arrays blocks the host each time (one burst of 10 499 symbols, four outputs:
        // 0.36 ms of copies and waits behind the recurrence kernel, now 0.1)
        char *pin = nullptr;
        HIP_TRY(ctx_pinned(c, all_bytes, (void **)&pin));
        HIP_TRY(d_stage.alloc(all_out ? all_out : 256));
        size_t off = 0;
        for (auto &o : outs) {
            if (!o.host) continue;
            HIP_TRY(launch_bursts_from_tiles(o.tiles.p, (int)out_rows, (int)B, (char *)d_stage.p + off, out_stride,
                                             (int)out_stride, o.elem, s));
            off += up256((size_t)B * out_stride * o.elem);
        }
        if (all_out) HIP_TRY(hipMemcpyAsync(pin, d_stage.p, all_out, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(pin + off_nout, d_nout.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(pin + off_status, d_status.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        if (st_out) HIP_TRY(hipMemcpyAsync(pin + off_st, d_stout.p, (size_t)B * 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        off = 0;
        for (auto &o : outs) {
            if (!o.host) continue;
            memcpy(o.host, pin + off, (size_t)B * out_stride * o.elem);
            off += up256((size_t)B * out_stride * o.elem);
        }
        memcpy(nout_h, pin + off_nout, (size_t)B * 4);
        memcpy(status.data(), pin + off_status, (size_t)B * 4);
        if (st_out) memcpy(st_out, pin + off_st, (size_t)B * 16);
    } else {
        HIP_TRY(d_stage.alloc((size_t)B * out_stride * max_elem));
        for (auto &o : outs) {
            if (!o.host) continue;
            HIP_TRY(launch_bursts_from_tiles(o.tiles.p, (int)out_rows, (int)B, d_stage.p, out_stride,
                                             (int)out_stride, o.elem, s));
            HIP_TRY(link_d2h(c, o.host, d_stage.p, (size_t)B * out_stride * o.elem));
            HIP_TRY(hipStreamSynchronize(s));
        }
        HIP_TRY(hipMemcpyAsync(nout_h, d_nout.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(status.data(), d_status.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        if (st_out) HIP_TRY(hipMemcpyAsync(st_out, d_stout.p, (size_t)B * 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    for (int64_t b = 0; b < B; b++)
        if (status[(size_t)b] & 1)
            return fail(DSP_E_CAPACITY, "burst %lld: out_stride %lld too small (loop still running "
                        "after %d symbols)", (long long)b, (long long)out_stride, nout_h[b]);
    return DSP_OK;
}

static int run_host(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                    const int32_t *lens, const double *scale, int sps, double alpha, double mu0,
                    int variant, int half_len, bool fll, double Bn, double y, const double *i0,
                    const double *st_in, double *st_out, void *rec, void *err, void *mu, void *sig,
                    void *freq, void *ef, uint8_t *dec, int32_t *n_out, int64_t out_stride)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (B < 0 || N < 0) return fail(DSP_E_ARG, "negative B or N");
    if (B == 0) return DSP_OK;
    if (!x) return fail(DSP_E_ARG, "x is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "bad dtype");
    if (x_stride < N) return fail(DSP_E_ARG, "x_stride < N");
    if (out_stride < 1) return fail(DSP_E_ARG, "out_stride < 1");
    if (lens)
        for (int64_t b = 0; b < B; b++)
            if (lens[b] < 0 || lens[b] > N) return fail(DSP_E_ARG, "lens[%lld] out of range", (long long)b);
    HIP_TRY(hipSetDevice(c->device));
    const int ce = dtype == DSP_C128 ? 16 : 8;
    const int64_t rows = dsp_tile_rows(N);
    const int64_t ntiles = (B + 63) / 64;
    hipStream_t s = c->stream;
    DevBuf d_x, d_xt, d_lens, d_scale;
    HIP_TRY(d_x.alloc((size_t)B * x_stride * ce));
    HIP_TRY(d_xt.alloc((size_t)ntiles * rows * 64 * ce));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)B * x_stride * ce));
    if (lens) {
        HIP_TRY(d_lens.alloc((size_t)B * 4));
        HIP_TRY(hipMemcpyAsync(d_lens.p, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    }
    if (scale) {
        HIP_TRY(d_scale.alloc((size_t)B * 8));
        HIP_TRY(hipMemcpyAsync(d_scale.p, scale, (size_t)B * 8, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(launch_tile_from_bursts(d_x.p, x_stride, d_lens.as<int>(), (int)N, (int)B, d_xt.p, (int)rows,
                                    ce, s));
    return sync_to_host(c, d_xt.p, dtype, B, N, rows, d_lens.as<int32_t>(), d_scale.as<double>(), sps,
                        alpha, mu0, variant, half_len, fll, Bn, y, i0, st_in, st_out, rec, err, mu, sig,
                        freq, ef, dec, n_out, out_stride);
}

// ---------------------------------------------------------------------------
// FIR / loaders
static int in_elem_bytes(int in_kind) { return in_kind == DSP_C128 ? 16 : (in_kind == DSP_I16 ? 4 : 8); }

int64_t dsp_fir_workspace_bytes(int in_kind, int64_t B, int64_t N, int n_taps)
{
    FirParams P, Q;
    if (n_taps < 1 || !fir_plan(in_kind, n_taps, (int)N, P, false)) return 0;
    fir_plan(in_kind, n_taps, (int)N, Q, true);          // either kernel's chunking fits
    return (int64_t)B * ((P.nchunk > Q.nchunk ? P.nchunk : Q.nchunk) + 8) * 4 * 8;     // + 8: fir_plan_strips
}

// The matched filter on device bursts; `ws` / `ws_bytes` hold the partial statistics, `taps_buf` (>= 64
// doubles) the stream kernels' zero-padded taps.  Launches only: no allocation, no synchronisation.
static int fir_tiles_run(dsp_ctx *c, const dsp_fir_args *a, void *ws, int64_t ws_bytes, double *taps_buf,
                         const char *ws_name)
{
    if (a->B <= 0) return DSP_OK;
    if (!a->src || !a->h || a->n_taps < 1) return fail(DSP_E_ARG, "fir: NULL src/h or no taps");
    if (a->in_kind < 0 || a->in_kind > 2) return fail(DSP_E_ARG, "fir: bad in_kind");
    if (a->N < 0 || a->N >= (1LL << 30)) return fail(DSP_E_ARG, "fir: N out of range");
    if (a->tiles && a->rows < a->N) return fail(DSP_E_ARG, "fir: rows < N");
    FirParams P;
    memset(&P, 0, sizeof(P));
    if (!fir_plan(a->in_kind, a->n_taps, (int)a->N, P, c->fir_stream != 0))
        return fail(DSP_E_ARG, "fir: %d taps are too many for the LDS-tiled kernel", a->n_taps);
    P.h_pad = taps_buf; P.strip_ok = c->fir_stream == 2; P.n_cu = c->n_cu;
    const bool want_std = a->std_out || a->inv_std_out;
    P.src = a->src; P.ld = a->src_stride; P.lens = a->lens; P.len_all = (int)a->N; P.B = (int)a->B;
    P.h = a->h; P.M = a->n_taps; P.start = a->start < 0 ? (a->n_taps - 1) / 2 : a->start;
    P.in_scale = a->in_scale; P.dst = a->tiles; P.rows = (int)a->rows; P.scale = a->scale;
    if (a->in_kind == DSP_IN_C128 && c->fir_stream == 2) (void)fir_try_c128_strips(P);
    if (want_std && (!ws || ws_bytes < (int64_t)a->B * (P.nchunk + 8) * 32))
        return fail(DSP_E_ARG, "fir: %s too small (need %lld bytes, have %lld)", ws_name,
                    (long long)a->B * (P.nchunk + 8) * 32, (long long)ws_bytes);
    P.stats = want_std ? (double *)ws : nullptr;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_fir_tile(P, a->in_kind, c->stream));
    if (want_std)
        HIP_TRY(launch_fir_stats_finish(P, a->std_out, a->inv_std_out, c->stream));
    return DSP_OK;
}

int dsp_fir_tiles_dev(dsp_ctx *c, const dsp_fir_args *a)
{
    if (!c || !a) return fail(DSP_E_ARG, "ctx/args is NULL");
    return fir_tiles_run(c, a, a->workspace, a->workspace_bytes, c->fir_taps, "workspace");
}

// dsp_fir_tiles_dev for callers that cannot allocate (and cannot hand over a workspace): the partial
// statistics and the padded taps live in the library's static device array instead.  The first 512 bytes
// hold the taps, the rest the statistics records; every record a call reads is written by that call
// first, so nothing has to start at zero.  Calls that overlap on one GPU share the array and race.
int dsp_fir_tiles_static_dev(dsp_ctx *c, const dsp_fir_args *a)
{
    if (!c || !a) return fail(DSP_E_ARG, "ctx/args is NULL");
    if (!c->static_scratch) return fail(DSP_E_HIP, "fir: the static scratch array has no device address");
    unsigned char *base = (unsigned char *)c->static_scratch;
    return fir_tiles_run(c, a, base + 512, (int64_t)DSP_STATIC_SCRATCH_BYTES - 512, (double *)base,
                         "static scratch (dsp_static_scratch_bytes())");
}

int64_t dsp_static_scratch_bytes(void) { return (int64_t)DSP_STATIC_SCRATCH_BYTES; }

int dsp_load_iq_i16(dsp_ctx *c, const int16_t *iq, int64_t n_words, int normalize, float *out_c64,
                    int64_t *n_out)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (n_words < 2) return fail(DSP_E_ARG, "load_iq_i16: need at least one I/Q pair");   // :80-81
    if (!iq || !out_c64) return fail(DSP_E_ARG, "load_iq_i16: NULL buffer");
    const int64_t n = n_words / 2;                                                     // :82-83
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_raw, d_out, d_tmp;
    HIP_TRY(d_raw.alloc((size_t)n * 4));
    HIP_TRY(d_out.alloc((size_t)n * 8));
    HIP_TRY(d_tmp.alloc(16));
    HIP_TRY(link_h2d(c, d_raw.p, iq, (size_t)n * 4));
    HIP_TRY(launch_iq_i16(d_raw.p, n, normalize, d_tmp.as<unsigned>(), d_tmp.as<float>() + 1, d_out.p, c->stream));
    HIP_TRY(link_d2h(c, out_c64, d_out.p, (size_t)n * 8));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_out) *n_out = n;
    return DSP_OK;
}

int dsp_load_iq_f32(dsp_ctx *c, const float *iq, int64_t n_words, float *out_c64, int64_t *n_out)
{
    (void)c;
    if (n_words < 0 || (n_words && (!iq || !out_c64))) return fail(DSP_E_ARG, "load_iq_f32: bad arguments");
    const int64_t n = n_words / 2;
    if (n) memcpy(out_c64, iq, (size_t)n * 8);      // interleaved float32 I,Q already is complex64
    if (n_out) *n_out = n;
    return DSP_OK;
}

// FIR of burst-major host bursts -> device tiles (+ stats).  Shared by dsp_fir_same_std and
// dsp_rx_chain.  On return d_tiles holds the (optionally scaled) filter output.
struct FirStage {
    DevBuf d_x, d_lens, d_h, d_tiles, d_ws, d_std, d_inv, d_peak;
    int64_t rows = 0;
};

static int fir_stage(dsp_ctx *c, FirStage &S, const void *x, int in_kind, int64_t B, int64_t N,
                     int64_t x_stride, const int32_t *lens, int normalize_peak, const double *h,
                     int n_taps, int start, bool want_std, bool scaled_output, bool x_on_device = false)
{
    const int ie = in_elem_bytes(in_kind), oe = in_kind == DSP_C128 ? 16 : 8;
    hipStream_t s = c->stream;
    S.rows = dsp_tile_rows(N);
    const int64_t ntiles = (B + 63) / 64;
    void *d_src = const_cast<void *>(x);              // bursts already resident (burst cutter) ...
    if (!x_on_device) {                               // ... or uploaded here
        HIP_TRY(S.d_x.alloc((size_t)B * x_stride * ie));
        HIP_TRY(link_h2d(c, S.d_x.p, x, (size_t)B * x_stride * ie));
        d_src = S.d_x.p;
    }
    if (lens) {
        HIP_TRY(S.d_lens.alloc((size_t)B * 4));
        HIP_TRY(hipMemcpyAsync(S.d_lens.p, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(S.d_h.alloc((size_t)n_taps * 8));
    HIP_TRY(hipMemcpyAsync(S.d_h.p, h, (size_t)n_taps * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(S.d_tiles.alloc((size_t)ntiles * S.rows * 64 * oe));
    HIP_TRY(hipMemsetAsync(S.d_tiles.p, 0, (size_t)ntiles * S.rows * 64 * oe, s));
    HIP_TRY(S.d_std.alloc((size_t)B * 8));
    HIP_TRY(S.d_inv.alloc((size_t)B * 8));
    HIP_TRY(S.d_peak.alloc(16));
    const float *in_scale = nullptr;
    if (in_kind == DSP_I16) {
        // one peak over the whole [B][x_stride] block (a capture cut into bursts shares its peak)
        HIP_TRY(launch_iq_i16(d_src, B * x_stride, normalize_peak, S.d_peak.as<unsigned>(),
                              S.d_peak.as<float>() + 1, nullptr, s));
        in_scale = S.d_peak.as<float>() + 1;
    }
    FirParams probe;
    if (!fir_plan(in_kind, n_taps, (int)N, probe)) {
        // long filter: direct form into a complex128 burst-major buffer, exact two-pass std,
        // scale in float64, then one rounding into the tiles
        DevBuf d_y64, d_y32;
        HIP_TRY(d_y64.alloc((size_t)B * N * 16));
        HIP_TRY(hipMemsetAsync(d_y64.p, 0, (size_t)B * N * 16, s));
        FirParams P;
        memset(&P, 0, sizeof(P));
        P.src = d_src; P.ld = x_stride; P.lens = S.d_lens.as<int>(); P.len_all = (int)N; P.B = (int)B;
        P.h = S.d_h.as<double>(); P.M = n_taps; P.start = start < 0 ? (n_taps - 1) / 2 : start;
        P.in_scale = in_scale;
        HIP_TRY(launch_fir_direct(P, in_kind, d_y64.p, N, (int)N, s));
        if (want_std)
            HIP_TRY(launch_burst_std(d_y64.p, 1, N, S.d_lens.as<int>(), (int)N, (int)B,
                                     S.d_std.as<double>(), S.d_inv.as<double>(), s));
        if (want_std && scaled_output)
            HIP_TRY(launch_burst_scale(d_y64.p, 1, N, S.d_lens.as<int>(), (int)N, (int)B, (int)N,
                                       S.d_inv.as<double>(), s));
        if (oe == 16) {
            HIP_TRY(launch_tile_from_bursts(d_y64.p, N, S.d_lens.as<int>(), (int)N, (int)B, S.d_tiles.p,
                                            (int)S.rows, 16, s));
        } else {
            HIP_TRY(d_y32.alloc((size_t)B * N * 8));
            HIP_TRY(launch_burst_cast(d_y64.p, N, d_y32.p, N, (int)B, (int)N, s));
            HIP_TRY(launch_tile_from_bursts(d_y32.p, N, S.d_lens.as<int>(), (int)N, (int)B, S.d_tiles.p,
                                            (int)S.rows, 8, s));
        }
        HIP_TRY(hipStreamSynchronize(s));      // the temporaries die with this scope
        return DSP_OK;
    }
    dsp_fir_args a;
    memset(&a, 0, sizeof(a));
    a.src = d_src; a.in_kind = in_kind; a.start = start; a.B = B; a.N = N; a.src_stride = x_stride;
    a.lens = S.d_lens.as<int32_t>(); a.h = S.d_h.as<double>(); a.n_taps = n_taps; a.in_scale = in_scale;
    a.rows = S.rows;
    int64_t wsb = dsp_fir_workspace_bytes(in_kind, B, N, n_taps);
    HIP_TRY(S.d_ws.alloc((size_t)(wsb > 0 ? wsb : 16)));
    a.workspace = S.d_ws.p; a.workspace_bytes = wsb;
    if (want_std && scaled_output) {
        // pass 1: statistics only; pass 2: filter again, multiply by 1/std, round once
        a.tiles = nullptr; a.std_out = S.d_std.as<double>(); a.inv_std_out = S.d_inv.as<double>();
        int rc = dsp_fir_tiles_dev(c, &a);
        if (rc != DSP_OK) return rc;
        a.tiles = S.d_tiles.p; a.std_out = nullptr; a.inv_std_out = nullptr; a.scale = S.d_inv.as<double>();
        return dsp_fir_tiles_dev(c, &a);
    }
    a.tiles = S.d_tiles.p;
    if (want_std) { a.std_out = S.d_std.as<double>(); a.inv_std_out = S.d_inv.as<double>(); }
    return dsp_fir_tiles_dev(c, &a);
}

// cut + derotation + matched filter (+ statistics) in one pass over the signal: launches only.  *fused = false
// (nothing launched) when the shape is not the fused kernels' -- more than FIR_MT taps, "fir_stream" 0.
static int fir_derot_run(dsp_ctx *c, const void *d_x, int dtype, const int64_t *d_off, const double *d_f, int f_stride,
                         int64_t B, int64_t N, const double *d_h, int n_taps, void *d_tiles, int64_t rows, double *d_std,
                         double *d_inv, void *ws, int64_t ws_bytes, bool *fused)
{
    *fused = false;
    FirParams P;
    memset(&P, 0, sizeof(P));
    P.src = d_x; P.src_off = d_off; P.derot_f = d_f; P.derot_fstride = f_stride;
    P.ld = N; P.len_all = (int)N; P.B = (int)B; P.M = n_taps; P.start = (n_taps - 1) / 2;
    P.strip_ok = c->fir_stream == 2; P.n_cu = c->n_cu;
    P.dst = d_tiles; P.rows = (int)rows;
    if (c->fir_stream == 0 || !fir_plan_derot(P, dtype)) return DSP_OK;
    const bool want_std = d_std || d_inv;
    if (want_std && (!ws || ws_bytes < B * (P.nchunk + 8) * 32))
        return fail(DSP_E_ARG, "rx_front: workspace too small (need %lld bytes, have %lld)",
                    (long long)B * (P.nchunk + 8) * 32, (long long)ws_bytes);
    P.h = d_h; P.h_pad = c->fir_taps;
    P.stats = want_std ? (double *)ws : nullptr;
    HIP_TRY(launch_fir_derot(P, dtype, c->stream));
    if (want_std) HIP_TRY(launch_fir_stats_finish(P, d_std, d_inv, c->stream));
    *fused = true;
    return DSP_OK;
}

// ... with the buffers of a FirStage (dsp_rx_bursts)
static int fir_stage_derot(dsp_ctx *c, FirStage &S, const void *d_x, int dtype, const int64_t *d_off, const double *d_f,
                           int f_stride, int64_t B, int64_t N, const double *h, int n_taps, bool want_std, bool *fused)
{
    hipStream_t s = c->stream;
    *fused = false;
    if (c->fir_stream == 0 || n_taps > DSP_FIR_MT) return DSP_OK;
    S.rows = dsp_tile_rows(N);
    const int64_t ntiles = (B + 63) / 64;
    const int64_t wsb = dsp_rx_front_workspace_bytes(B, N);
    HIP_TRY(S.d_h.alloc((size_t)n_taps * 8));
    HIP_TRY(hipMemcpyAsync(S.d_h.p, h, (size_t)n_taps * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(S.d_tiles.alloc((size_t)ntiles * S.rows * 64 * 16));
    HIP_TRY(hipMemsetAsync(S.d_tiles.p, 0, (size_t)ntiles * S.rows * 64 * 16, s));
    HIP_TRY(S.d_std.alloc((size_t)B * 8));
    HIP_TRY(S.d_inv.alloc((size_t)B * 8));
    HIP_TRY(S.d_ws.alloc((size_t)wsb));
    return fir_derot_run(c, d_x, dtype, d_off, d_f, f_stride, B, N, S.d_h.as<double>(), n_taps, S.d_tiles.p, S.rows,
                         want_std ? S.d_std.as<double>() : nullptr, want_std ? S.d_inv.as<double>() : nullptr, S.d_ws.p, wsb,
                         fused);
}

int64_t dsp_rx_front_workspace_bytes(int64_t B, int64_t n_pkt)
{
    FirParams P;
    if (B < 1 || n_pkt < 1 || n_pkt > 0x7fffffff || !fir_plan(DSP_IN_C128, DSP_FIR_MT, (int)n_pkt, P, true, true)) return 0;
    return B * (P.nchunk + 8) * 32;
}

// device-resident front end of dsp_rx_bursts, fused: frequency (estimated or given), then cut + derotation +
// matched filter + std in ONE pass over the signal.  Asynchronous.
int dsp_rx_front_dev(dsp_ctx *c, const void *d_x, int dtype, int64_t n, const int64_t *d_offsets, int64_t B,
                     int64_t n_pkt, int sps, const double *d_preamble_c128, int n_pre, int cfo_method, double *d_f,
                     const double *d_h, int n_taps, void *d_tiles_c128, int64_t rows, double *d_std, double *d_inv_std,
                     void *workspace, int64_t workspace_bytes)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "rx_front: dtype must be DSP_C64 or DSP_C128");
    if (!d_x || !d_offsets || !d_f || !d_h) return fail(DSP_E_ARG, "rx_front: NULL argument");
    if (B < 1 || B > 0x7fffffff || n_pkt < 1 || n_pkt > n || n_pkt > 0x7fffffff || sps < 1)
        return fail(DSP_E_ARG, "rx_front: bad sizes");
    if (n_taps < 1 || n_taps > n_pkt) return fail(DSP_E_ARG, "rx_front: 1 <= n_taps <= n_pkt");
    if (n_taps > DSP_FIR_MT || c->fir_stream == 0)
        return fail(DSP_E_ARG, "rx_front: the fused kernels take at most %d taps (dsp_cfo_derotate_dev + dsp_fir_tiles_dev "
                    "for longer filters)", DSP_FIR_MT);
    if (d_tiles_c128 && rows < n_pkt) return fail(DSP_E_ARG, "rx_front: rows < n_pkt");
    if (cfo_method < 0 || cfo_method > 3) return fail(DSP_E_ARG, "rx_front: cfo_method is 0 or 1..3");
    if (cfo_method != 0 && (!d_preamble_c128 || n_pre < 2 || n_pre > 4096 || (int64_t)(n_pre - 1) * sps + 1 > n_pkt))
        return fail(DSP_E_ARG, "rx_front: the preamble needs 2..4096 symbols inside the packet");
    HIP_TRY(hipSetDevice(c->device));
    int f_stride = 1;
    const double *d_fsel = d_f;
    if (cfo_method != 0) {
        HIP_TRY(launch_cfo_preamble(d_x, dtype == DSP_C128, d_offsets, (int)B, sps, d_preamble_c128, n_pre, d_f, c->stream));
        d_fsel = d_f + (cfo_method - 1);
        f_stride = 3;
    }
    bool fused = false;
    int rc = fir_derot_run(c, d_x, dtype, d_offsets, d_fsel, f_stride, B, n_pkt, d_h, n_taps, d_tiles_c128, rows, d_std,
                           d_inv_std, workspace, workspace_bytes, &fused);
    if (rc != DSP_OK) return rc;
    if (!fused) return fail(DSP_E_ARG, "rx_front: shape not supported by the fused kernels");
    return DSP_OK;
}

int dsp_fir_same_std(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                     const int32_t *lens, const double *h, int n_taps, int start, int normalize_std,
                     void *y, int64_t y_stride, double *std_out)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (B < 0 || N < 0) return fail(DSP_E_ARG, "negative B or N");
    if (B == 0 || N == 0) return DSP_OK;
    if (!x || !y || !h || n_taps < 1) return fail(DSP_E_ARG, "fir: NULL argument");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "fir: bad dtype");
    if (x_stride < N || y_stride < N) return fail(DSP_E_ARG, "fir: stride < N");
    if (n_taps > N) return fail(DSP_E_ARG, "fir: n_taps > N (np.convolve would swap its operands)");
    HIP_TRY(hipSetDevice(c->device));
    const int oe = dtype == DSP_C128 ? 16 : 8;
    hipStream_t s = c->stream;
    const bool want_std = normalize_std || std_out;
    DevBuf d_y;
    HIP_TRY(d_y.alloc((size_t)B * y_stride * oe));
    FirStage S;
    int rc = fir_stage(c, S, x, dtype, B, N, x_stride, lens, 0, h, n_taps, start, want_std,
                       normalize_std != 0);
    if (rc != DSP_OK) return rc;
    HIP_TRY(hipMemsetAsync(d_y.p, 0, (size_t)B * y_stride * oe, s));
    HIP_TRY(launch_bursts_from_tiles(S.d_tiles.p, (int)S.rows, (int)B, d_y.p, y_stride, (int)N, oe, s));
    HIP_TRY(link_d2h(c, y, d_y.p, (size_t)B * y_stride * oe));
    if (std_out) HIP_TRY(hipMemcpyAsync(std_out, S.d_std.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DSP_OK;
}

int dsp_gardner(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                const int32_t *lens, int sps, double alpha, double mu0, int variant,
                int interp_half_len, void *rec, void *err, void *mu, int32_t *n_out,
                int64_t out_stride)
{
    if (variant != DSP_GARDNER2 && variant != DSP_GARDNER_TED) return fail(DSP_E_ARG, "gardner: bad variant");
    if (sps < 1) return fail(DSP_E_ARG, "gardner: sps must be >= 1");
    return run_host(c, x, dtype, B, N, x_stride, lens, nullptr, sps, alpha, mu0, variant,
                    interp_half_len, false, 0.0, 1.0, nullptr, nullptr, nullptr, rec, err, mu,
                    nullptr, nullptr, nullptr, nullptr, n_out, out_stride);
}

int dsp_fll(dsp_ctx *c, const void *x, int dtype, int64_t B, const int32_t *M, int64_t M_max,
            int64_t x_stride, double Bn, double y, const double *i0, const double *state_in,
            double *state_out, void *out, void *freq, void *ef, uint8_t *dec, int64_t out_stride)
{
    if (out_stride < M_max) return fail(DSP_E_ARG, "fll: out_stride < M_max");
    return run_host(c, x, dtype, B, M_max, x_stride, M, nullptr, 1, 0.0, 0.0, 2, 0, true, Bn, y, i0,
                    state_in, state_out, nullptr, nullptr, nullptr, out, freq, ef, dec, nullptr,
                    out_stride);
}

int dsp_sync_chain(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                   const int32_t *lens, const double *scale, int sps, double alpha, double mu0,
                   int variant, int interp_half_len, double Bn, double y,
                   void *rec, void *err, void *mu, void *sig, void *freq, void *ef, uint8_t *dec,
                   int32_t *n_out, int64_t out_stride)
{
    if (variant != DSP_GARDNER2 && variant != DSP_GARDNER_TED) return fail(DSP_E_ARG, "chain: bad variant");
    if (sps < 1) return fail(DSP_E_ARG, "chain: sps must be >= 1");
    return run_host(c, x, dtype, B, N, x_stride, lens, scale, sps, alpha, mu0, variant,
                    interp_half_len, true, Bn, y, nullptr, nullptr, nullptr, rec, err, mu, sig, freq,
                    ef, dec, n_out, out_stride);
}

int dsp_rx_chain(dsp_ctx *c, const void *x, int in_kind, int64_t B, int64_t N, int64_t x_stride,
                 const int32_t *lens, int normalize_peak, const double *h, int n_taps,
                 int normalize_std, int sps, double alpha, double mu0, int variant,
                 int interp_half_len, double Bn, double y,
                 void *rec, void *err, void *mu, void *sig, void *freq, void *ef, uint8_t *dec,
                 int32_t *n_out, int64_t out_stride, double *std_out)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (B < 0 || N < 0) return fail(DSP_E_ARG, "negative B or N");
    if (B == 0) return DSP_OK;
    if (!x || !h || n_taps < 1) return fail(DSP_E_ARG, "rx_chain: NULL argument");
    if (in_kind < 0 || in_kind > 2) return fail(DSP_E_ARG, "rx_chain: bad in_kind");
    if (variant != DSP_GARDNER2 && variant != DSP_GARDNER_TED) return fail(DSP_E_ARG, "rx_chain: bad variant");
    if (sps < 1) return fail(DSP_E_ARG, "rx_chain: sps must be >= 1");
    if (x_stride < N) return fail(DSP_E_ARG, "rx_chain: x_stride < N");
    if (n_taps > N) return fail(DSP_E_ARG, "rx_chain: n_taps > N");
    if (lens)
        for (int64_t b = 0; b < B; b++)
            if (lens[b] < 0 || lens[b] > N) return fail(DSP_E_ARG, "lens[%lld] out of range", (long long)b);
    HIP_TRY(hipSetDevice(c->device));
    FirStage S;
    const bool want_std = normalize_std || std_out;
    int rc = fir_stage(c, S, x, in_kind, B, N, x_stride, lens, normalize_peak, h, n_taps, -1, want_std,
                       false);
    if (rc != DSP_OK) return rc;
    if (std_out) HIP_TRY(hipMemcpyAsync(std_out, S.d_std.p, (size_t)B * 8, hipMemcpyDeviceToHost, c->stream));
    const int dtype = in_kind == DSP_C128 ? DSP_C128 : DSP_C64;
    return sync_to_host(c, S.d_tiles.p, dtype, B, N, S.rows, S.d_lens.as<int32_t>(),
                        normalize_std ? S.d_inv.as<double>() : nullptr, sps, alpha, mu0, variant,
                        interp_half_len, true, Bn, y, nullptr, nullptr, nullptr, rec, err, mu, sig, freq,
                        ef, dec, n_out, out_stride);
}

// differential-correlation preamble search -- diff_method.py:92-162 (preamble_kernels.hip)
int dsp_diff_corr(dsp_ctx *c, const void *x, int dtype, int64_t n, const double *preamble_c128, int n_pre, int sps,
                  float *conv_c64, int64_t k_stride)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "diff_corr: dtype must be DSP_C64 or DSP_C128");
    if (!x || !preamble_c128 || !conv_c64) return fail(DSP_E_ARG, "diff_corr: NULL argument");
    if (n < 1 || sps < 1 || sps > 64) return fail(DSP_E_ARG, "diff_corr: bad n / sps");
    if (n_pre < 2 || n_pre > 4096) return fail(DSP_E_ARG, "diff_corr: the preamble needs 2..4096 symbols");
    if (k_stride < 1) return fail(DSP_E_ARG, "diff_corr: k_stride < 1");
    const int L = n_pre - 1;
    std::vector<double> dpre((size_t)L * 2);
    for (int j = 0; j < L; j++) {                    // d_pre = pre[1:] * conj(pre[:-1])   :118
        const double ar = preamble_c128[2 * j], ai = preamble_c128[2 * j + 1];
        const double br = preamble_c128[2 * j + 2], bi = preamble_c128[2 * j + 3];
        dpre[2 * j] = br * ar + bi * ai;
        dpre[2 * j + 1] = bi * ar - br * ai;
    }
    const size_t eb = dtype == DSP_C128 ? 16 : 8;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_x, d_p, d_c;
    HIP_TRY(d_x.alloc((size_t)n * eb)); HIP_TRY(d_p.alloc((size_t)L * 16));
    HIP_TRY(d_c.alloc((size_t)sps * k_stride * 8));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)n * eb));
    HIP_TRY(hipMemcpyAsync(d_p.p, dpre.data(), (size_t)L * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_diff_corr(d_x.p, dtype == DSP_C128, n, sps, d_p.p, L, d_c.p, k_stride, c->corr_window, c->stream));
    HIP_TRY(link_d2h(c, conv_c64, d_c.p, (size_t)sps * k_stride * 8));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

// fll_func with worker.Processing's constellation switch (nco_kernels.hip fll_slicer_kernel)
int dsp_fll_slicer(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                   const int32_t *lens, int kind, const double *points_c128, int n_points, double Bn, double y,
                   void *out, void *freq, void *ef, uint8_t *dec, int64_t out_stride, double *curr_freq)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (B < 0 || N < 0) return fail(DSP_E_ARG, "negative B or N");
    if (B == 0 || N == 0) {
        if (curr_freq) for (int64_t b = 0; b < B; b++) curr_freq[b] = 0.0;
        return DSP_OK;
    }
    if (!x) return fail(DSP_E_ARG, "x is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "bad dtype");
    if (kind < 0 || kind > 2) return fail(DSP_E_ARG, "fll_slicer: kind must be 0 bpsk, 1 qpsk, 2 table");
    if (kind == 2 && (!points_c128 || n_points < 1 || n_points > 64))
        return fail(DSP_E_ARG, "fll_slicer: the table needs 1..64 points");
    if (x_stride < N || out_stride < N) return fail(DSP_E_ARG, "stride < N");
    if (lens)
        for (int64_t b = 0; b < B; b++)
            if (lens[b] < 0 || lens[b] > N) return fail(DSP_E_ARG, "lens[%lld] out of range", (long long)b);
    double kp, ki;
    int rc = dsp_fll_gains(Bn, y, &kp, &ki);
    if (rc != DSP_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const int ce = dtype == DSP_C128 ? 16 : 8, re = ce / 2;
    const int64_t rows = dsp_tile_rows(N), ntiles = (B + 63) / 64;
    hipStream_t s = c->stream;
    DevBuf d_x, d_xt, d_lens, d_ff, d_stage, d_pts;
    HIP_TRY(d_x.alloc((size_t)B * x_stride * ce));
    HIP_TRY(d_xt.alloc((size_t)ntiles * rows * 64 * ce));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)B * x_stride * ce));
    if (lens) {
        HIP_TRY(d_lens.alloc((size_t)B * 4));
        HIP_TRY(hipMemcpyAsync(d_lens.p, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    }
    if (kind == 2) {
        HIP_TRY(d_pts.alloc((size_t)n_points * 16));
        HIP_TRY(hipMemcpyAsync(d_pts.p, points_c128, (size_t)n_points * 16, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(launch_tile_from_bursts(d_x.p, x_stride, d_lens.as<int>(), (int)N, (int)B, d_xt.p, (int)rows, ce, s));
    struct OutSpec { void *host; int elem; DevBuf tiles; };
    OutSpec outs[4] = {{out, ce, {}}, {freq, re, {}}, {ef, re, {}}, {dec, 1, {}}};
    for (auto &o : outs)
        if (o.host) {
            HIP_TRY(o.tiles.alloc((size_t)ntiles * rows * 64 * o.elem));
            HIP_TRY(hipMemsetAsync(o.tiles.p, 0, (size_t)ntiles * rows * 64 * o.elem, s));
        }
    HIP_TRY(d_ff.alloc((size_t)B * 8));
    HIP_TRY(launch_fll_slicer(d_xt.p, dtype == DSP_C128, d_lens.as<int>(), (int)N, (int)B, (int)rows, kind, d_pts.p,
                              kind == 2 ? n_points : 0, kp, ki, outs[0].tiles.p, outs[1].tiles.p, outs[2].tiles.p,
                              outs[3].tiles.as<unsigned char>(), d_ff.as<double>(), s));
    HIP_TRY(d_stage.alloc((size_t)B * out_stride * ce));
    for (auto &o : outs) {
        if (!o.host) continue;
        HIP_TRY(hipMemsetAsync(d_stage.p, 0, (size_t)B * out_stride * o.elem, s));
        HIP_TRY(launch_bursts_from_tiles(o.tiles.p, (int)rows, (int)B, d_stage.p, out_stride, (int)N, o.elem, s));
        HIP_TRY(link_d2h(c, o.host, d_stage.p, (size_t)B * out_stride * o.elem));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (curr_freq) HIP_TRY(hipMemcpyAsync(curr_freq, d_ff.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DSP_OK;
}

// ---- coarse CFO estimators + burst cutter (cfo_kernels.hip) --------------------
static int check_cuts(const char *who, int64_t n, const int64_t *offsets, int64_t B, int64_t span)
{
    for (int64_t b = 0; b < B; b++)
        if (offsets[b] < 0 || offsets[b] + span > n)
            return fail(DSP_E_ARG, "%s: burst %lld (offset %lld, %lld samples) leaves the signal of %lld samples", who,
                        (long long)b, (long long)offsets[b], (long long)span, (long long)n);
    return DSP_OK;
}

int dsp_cfo_preamble(dsp_ctx *c, const void *x, int dtype, int64_t n, const int64_t *offsets, int64_t B, int sps,
                     const double *preamble_c128, int n_pre, double *f_rel)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "cfo_preamble: dtype must be DSP_C64 or DSP_C128");
    if (B < 0) return fail(DSP_E_ARG, "cfo_preamble: negative B");
    if (B == 0) return DSP_OK;
    if (!x || !offsets || !preamble_c128 || !f_rel) return fail(DSP_E_ARG, "cfo_preamble: NULL argument");
    if (sps < 1 || n < 1) return fail(DSP_E_ARG, "cfo_preamble: bad n / sps");
    if (n_pre < 2 || n_pre > 4096) return fail(DSP_E_ARG, "cfo_preamble: the preamble needs 2..4096 symbols");
    if (B > 0x7fffffff) return fail(DSP_E_ARG, "cfo_preamble: too many bursts");
    int rc = check_cuts("cfo_preamble", n, offsets, B, (int64_t)(n_pre - 1) * sps + 1);
    if (rc != DSP_OK) return rc;
    const size_t eb = dtype == DSP_C128 ? 16 : 8;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_x, d_o, d_p, d_f;
    HIP_TRY(d_x.alloc((size_t)n * eb)); HIP_TRY(d_o.alloc((size_t)B * 8));
    HIP_TRY(d_p.alloc((size_t)n_pre * 16)); HIP_TRY(d_f.alloc((size_t)B * 24));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)n * eb));
    HIP_TRY(hipMemcpyAsync(d_o.p, offsets, (size_t)B * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_p.p, preamble_c128, (size_t)n_pre * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_cfo_preamble(d_x.p, dtype == DSP_C128, d_o.as<int64_t>(), (int)B, sps, d_p.p, n_pre,
                                d_f.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(f_rel, d_f.p, (size_t)B * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

int dsp_cfo_fourth_power(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                         const int32_t *lens, int64_t num_samples, double *freq)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "cfo_fourth_power: dtype must be DSP_C64 or DSP_C128");
    if (B < 0 || N < 0 || N > 0x7fffffff || B > 0x7fffffff) return fail(DSP_E_ARG, "cfo_fourth_power: bad B or N");
    if (B == 0) return DSP_OK;
    if (!freq || (!x && N > 0)) return fail(DSP_E_ARG, "cfo_fourth_power: NULL argument");
    if (x_stride < N) return fail(DSP_E_ARG, "cfo_fourth_power: x_stride < N");
    if (lens)
        for (int64_t b = 0; b < B; b++)
            if (lens[b] < 0 || lens[b] > N) return fail(DSP_E_ARG, "lens[%lld] out of range", (long long)b);
    if (N == 0) { for (int64_t b = 0; b < B; b++) freq[b] = 0.0; return DSP_OK; }
    const size_t eb = dtype == DSP_C128 ? 16 : 8;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_x, d_l, d_f;
    HIP_TRY(d_x.alloc((size_t)B * x_stride * eb)); HIP_TRY(d_f.alloc((size_t)B * 8));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)B * x_stride * eb));
    if (lens) {
        HIP_TRY(d_l.alloc((size_t)B * 4));
        HIP_TRY(hipMemcpyAsync(d_l.p, lens, (size_t)B * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(launch_cfo_fourth_power(d_x.p, dtype == DSP_C128, x_stride, d_l.as<int>(), (int)N, (int)B, num_samples,
                                    d_f.as<double>(), c->stream));
    HIP_TRY(hipMemcpyAsync(freq, d_f.p, (size_t)B * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

int dsp_derotate(dsp_ctx *c, const void *x, int dtype, int64_t n, const int64_t *offsets, int64_t B, int64_t n_cut,
                 const double *f_rel, const int64_t *index0, double *out_c128, int64_t out_stride)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "derotate: dtype must be DSP_C64 or DSP_C128");
    if (B < 0 || n_cut < 0) return fail(DSP_E_ARG, "derotate: negative B or n_cut");
    if (B == 0 || n_cut == 0) return DSP_OK;
    if (!x || !f_rel || !out_c128) return fail(DSP_E_ARG, "derotate: NULL argument");
    if (out_stride < n_cut) return fail(DSP_E_ARG, "derotate: out_stride < n_cut");
    if (B > 0x7fffffff) return fail(DSP_E_ARG, "derotate: too many bursts");
    if (offsets) {
        int rc = check_cuts("derotate", n, offsets, B, n_cut);
        if (rc != DSP_OK) return rc;
    } else if (n_cut > n) {
        return fail(DSP_E_ARG, "derotate: n_cut > n");
    }
    const size_t eb = dtype == DSP_C128 ? 16 : 8;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_x, d_o, d_i, d_f, d_y;
    HIP_TRY(d_x.alloc((size_t)n * eb)); HIP_TRY(d_f.alloc((size_t)B * 8));
    HIP_TRY(d_y.alloc((size_t)B * out_stride * 16));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)n * eb));
    HIP_TRY(hipMemcpyAsync(d_f.p, f_rel, (size_t)B * 8, hipMemcpyHostToDevice, c->stream));
    if (offsets) {
        HIP_TRY(d_o.alloc((size_t)B * 8));
        HIP_TRY(hipMemcpyAsync(d_o.p, offsets, (size_t)B * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (index0) {
        HIP_TRY(d_i.alloc((size_t)B * 8));
        HIP_TRY(hipMemcpyAsync(d_i.p, index0, (size_t)B * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (out_stride > n_cut) HIP_TRY(hipMemsetAsync(d_y.p, 0, (size_t)B * out_stride * 16, c->stream));
    HIP_TRY(launch_derotate_cut(d_x.p, dtype == DSP_C128, d_o.as<int64_t>(), d_i.as<int64_t>(), d_f.as<double>(), 1,
                                (int)B, n_cut, d_y.p, out_stride, c->stream));
    HIP_TRY(link_d2h(c, out_c128, d_y.p, (size_t)B * out_stride * 16));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

// device-resident core of dsp_rx_bursts' front end: estimate (cfo_method 1..3 -> d_f [B][3]) or take
// (cfo_method 0, d_f [B]) the frequency, then cut + derotate into complex128 [B][n_pkt].  Asynchronous.
int dsp_cfo_derotate_dev(dsp_ctx *c, const void *d_x, int dtype, int64_t n, const int64_t *d_offsets, int64_t B,
                         int64_t n_pkt, int sps, const double *d_preamble_c128, int n_pre, int cfo_method,
                         double *d_f, double *d_out_c128)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "cfo_derotate: dtype must be DSP_C64 or DSP_C128");
    if (!d_x || !d_offsets || !d_f || !d_out_c128) return fail(DSP_E_ARG, "cfo_derotate: NULL argument");
    if (B < 1 || B > 0x7fffffff || n_pkt < 1 || n_pkt > n || sps < 1) return fail(DSP_E_ARG, "cfo_derotate: bad sizes");
    if (cfo_method < 0 || cfo_method > 3) return fail(DSP_E_ARG, "cfo_derotate: cfo_method is 0 or 1..3");
    if (cfo_method != 0 && (!d_preamble_c128 || n_pre < 2 || n_pre > 4096 || (int64_t)(n_pre - 1) * sps + 1 > n_pkt))
        return fail(DSP_E_ARG, "cfo_derotate: the preamble needs 2..4096 symbols inside the packet");
    HIP_TRY(hipSetDevice(c->device));
    int f_stride = 1;
    const double *d_fsel = d_f;
    if (cfo_method != 0) {
        HIP_TRY(launch_cfo_preamble(d_x, dtype == DSP_C128, d_offsets, (int)B, sps, d_preamble_c128, n_pre, d_f,
                                    c->stream));
        d_fsel = d_f + (cfo_method - 1);
        f_stride = 3;
    }
    // phase index = arange + offset (worker.py:298)
    HIP_TRY(launch_derotate_cut(d_x, dtype == DSP_C128, d_offsets, nullptr, d_fsel, f_stride, (int)B, n_pkt,
                                d_out_c128, n_pkt, c->stream));
    return DSP_OK;
}

// worker.Processing.process per-burst body (worker.py:284-304) for all bursts at once: the signal goes
// to the device once; coarse CFO from the preamble, cut + derotation, matched filter + /std, Gardner,
// FLL all stay there; only the requested outputs come back.
int dsp_rx_bursts(dsp_ctx *c, const void *x, int dtype, int64_t n, const int64_t *offsets, int64_t B, int64_t n_pkt,
                  const double *preamble_c128, int n_pre, int cfo_method, const double *f_rel_in,
                  const double *h, int n_taps, int normalize_std, int sps, double alpha, double mu0, int variant,
                  int interp_half_len, double Bn, double y,
                  void *rec, void *err, void *mu, void *sig, void *freq, void *ef, uint8_t *dec,
                  int32_t *n_out, int64_t out_stride, double *f_rel_out, double *std_out)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "rx_bursts: dtype must be DSP_C64 or DSP_C128");
    if (B < 0 || n_pkt < 0) return fail(DSP_E_ARG, "rx_bursts: negative B or n_pkt");
    if (B == 0) return DSP_OK;
    if (!x || !offsets || !h || n_taps < 1) return fail(DSP_E_ARG, "rx_bursts: NULL argument");
    if (cfo_method < 0 || cfo_method > 3) return fail(DSP_E_ARG, "rx_bursts: cfo_method is 0 (f_rel given) or 1..3");
    if (cfo_method == 0 && !f_rel_in) return fail(DSP_E_ARG, "rx_bursts: cfo_method 0 needs f_rel_in");
    if (cfo_method != 0 && (!preamble_c128 || n_pre < 2 || n_pre > 4096))
        return fail(DSP_E_ARG, "rx_bursts: the preamble needs 2..4096 symbols");
    if (variant != DSP_GARDNER2 && variant != DSP_GARDNER_TED) return fail(DSP_E_ARG, "rx_bursts: bad variant");
    if (sps < 1) return fail(DSP_E_ARG, "rx_bursts: sps must be >= 1");
    if (n_taps > n_pkt) return fail(DSP_E_ARG, "rx_bursts: n_taps > n_pkt");
    if (B > 0x7fffffff || n_pkt > 0x7fffffff) return fail(DSP_E_ARG, "rx_bursts: B or n_pkt too large");
    if (cfo_method != 0 && (int64_t)(n_pre - 1) * sps + 1 > n_pkt)
        return fail(DSP_E_ARG, "rx_bursts: the preamble is longer than the packet");
    int rc = check_cuts("rx_bursts", n, offsets, B, n_pkt);
    if (rc != DSP_OK) return rc;
    const size_t eb = dtype == DSP_C128 ? 16 : 8;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf d_x, d_o, d_p, d_f, d_cut;
    HIP_TRY(d_x.alloc((size_t)n * eb)); HIP_TRY(d_o.alloc((size_t)B * 8)); HIP_TRY(d_f.alloc((size_t)B * 24));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)n * eb));
    HIP_TRY(hipMemcpyAsync(d_o.p, offsets, (size_t)B * 8, hipMemcpyHostToDevice, s));
    if (cfo_method == 0) {
        HIP_TRY(hipMemcpyAsync(d_f.p, f_rel_in, (size_t)B * 8, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(d_p.alloc((size_t)n_pre * 16));
        HIP_TRY(hipMemcpyAsync(d_p.p, preamble_c128, (size_t)n_pre * 16, hipMemcpyHostToDevice, s));
    }
    if (cfo_method != 0)
        HIP_TRY(launch_cfo_preamble(d_x.p, dtype == DSP_C128, d_o.as<int64_t>(), (int)B, sps, d_p.as<double>(), n_pre,
                                    d_f.as<double>(), s));
    const int f_stride = cfo_method == 0 ? 1 : 3;
    const double *d_fsel = d_f.as<double>() + (cfo_method == 0 ? 0 : cfo_method - 1);
    if (f_rel_out) {
        if (f_stride == 1) {
            HIP_TRY(hipMemcpyAsync(f_rel_out, d_f.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
        } else {
            HIP_TRY(hipMemcpy2DAsync(f_rel_out, 8, d_fsel, 24, 8, (size_t)B, hipMemcpyDeviceToHost, s));
        }
    }
    // cut + derotation (worker.py:287,298-299) ride in the matched filter's staging loads where its LDS-tiled
    // kernels apply -- the signal is read once and the derotated bursts never exist in memory -- and are a pass
    // of their own in front of it otherwise (long filters; "fuse_derot" 0).  Same arithmetic, same bits.
    FirStage S;
    const bool want_std = normalize_std || std_out;
    bool fused = false;
    if (c->fuse_derot) {
        rc = fir_stage_derot(c, S, d_x.p, dtype, d_o.as<int64_t>(), d_fsel, f_stride, B, n_pkt, h, n_taps, want_std, &fused);
        if (rc != DSP_OK) return rc;
    }
    if (!fused) {
        HIP_TRY(d_cut.alloc((size_t)B * n_pkt * 16));
        HIP_TRY(launch_derotate_cut(d_x.p, dtype == DSP_C128, d_o.as<int64_t>(), nullptr, d_fsel, f_stride, (int)B, n_pkt,
                                    d_cut.p, n_pkt, s));
        rc = fir_stage(c, S, d_cut.p, DSP_C128, B, n_pkt, n_pkt, nullptr, 0, h, n_taps, -1, want_std, false, true);
        if (rc != DSP_OK) return rc;
    }
    if (std_out) HIP_TRY(hipMemcpyAsync(std_out, S.d_std.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    return sync_to_host(c, S.d_tiles.p, DSP_C128, B, n_pkt, S.rows, nullptr,
                        normalize_std ? S.d_inv.as<double>() : nullptr, sps, alpha, mu0, variant,
                        interp_half_len, true, Bn, y, nullptr, nullptr, nullptr, rec, err, mu, sig, freq,
                        ef, dec, n_out, out_stride);
}

// class FrequencyLockedLoop facade (nco_kernels.hip; parity unpinned: fll.py is absent)
int dsp_fll_class(dsp_ctx *c, const void *x, int dtype, int64_t B, int64_t N, int64_t x_stride,
                  const int32_t *lens, int detector, double Kp, double Ki, double freq_limit,
                  void *out, void *errors, void *freqs, void *phases, int64_t out_stride, double *freq_estimate)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (B < 0 || N < 0) return fail(DSP_E_ARG, "negative B or N");
    if (B == 0 || N == 0) return DSP_OK;
    if (!x) return fail(DSP_E_ARG, "x is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "bad dtype");
    if (detector < 0 || detector > 2) return fail(DSP_E_ARG, "detector must be 0 cross_product, 1 atan2, 2 decision_directed");
    if (x_stride < N || out_stride < N) return fail(DSP_E_ARG, "stride < N");
    if (!(freq_limit > 0.0)) return fail(DSP_E_ARG, "freq_limit must be > 0");
    if (lens)
        for (int64_t b = 0; b < B; b++)
            if (lens[b] < 0 || lens[b] > N) return fail(DSP_E_ARG, "lens[%lld] out of range", (long long)b);
    HIP_TRY(hipSetDevice(c->device));
    const int ce = dtype == DSP_C128 ? 16 : 8, re = ce / 2;
    const int64_t rows = dsp_tile_rows(N), ntiles = (B + 63) / 64;
    hipStream_t s = c->stream;
    DevBuf d_x, d_xt, d_lens, d_ff, d_stage;
    HIP_TRY(d_x.alloc((size_t)B * x_stride * ce));
    HIP_TRY(d_xt.alloc((size_t)ntiles * rows * 64 * ce));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)B * x_stride * ce));
    if (lens) {
        HIP_TRY(d_lens.alloc((size_t)B * 4));
        HIP_TRY(hipMemcpyAsync(d_lens.p, lens, (size_t)B * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(launch_tile_from_bursts(d_x.p, x_stride, d_lens.as<int>(), (int)N, (int)B, d_xt.p, (int)rows, ce, s));
    struct OutSpec { void *host; int elem; DevBuf tiles; };
    OutSpec outs[4] = {{out, ce, {}}, {errors, re, {}}, {freqs, re, {}}, {phases, re, {}}};
    for (auto &o : outs)
        if (o.host) {
            HIP_TRY(o.tiles.alloc((size_t)ntiles * rows * 64 * o.elem));
            HIP_TRY(hipMemsetAsync(o.tiles.p, 0, (size_t)ntiles * rows * 64 * o.elem, s));
        }
    HIP_TRY(d_ff.alloc((size_t)B * 8));
    HIP_TRY(launch_fll_class(d_xt.p, dtype == DSP_C128, d_lens.as<int>(), (int)N, (int)B, (int)rows, detector, Kp, Ki,
                             freq_limit, outs[0].tiles.p, outs[1].tiles.p, outs[2].tiles.p, outs[3].tiles.p,
                             d_ff.as<double>(), s));
    HIP_TRY(d_stage.alloc((size_t)B * out_stride * ce));
    for (auto &o : outs) {
        if (!o.host) continue;
        HIP_TRY(launch_bursts_from_tiles(o.tiles.p, (int)rows, (int)B, d_stage.p, out_stride, (int)N, o.elem, s));
        HIP_TRY(link_d2h(c, o.host, d_stage.p, (size_t)B * out_stride * o.elem));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (freq_estimate) HIP_TRY(hipMemcpyAsync(freq_estimate, d_ff.p, (size_t)B * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return DSP_OK;
}

// lagrange_resample -- hw1/test_lagrange_resampling.py:34-72
int dsp_lagrange_resample(dsp_ctx *c, const void *x, int dtype, int64_t n, const double *positions, int64_t P,
                          int half_len, double *out_c128)
{
    if (!c) return fail(DSP_E_ARG, "ctx is NULL");
    if (dtype != DSP_C64 && dtype != DSP_C128) return fail(DSP_E_ARG, "lagrange: dtype must be DSP_C64 or DSP_C128");
    if (n < 1 || n >= (1LL << 31)) return fail(DSP_E_ARG, "lagrange: signal length out of range");
    if (half_len < 1 || half_len > 8) return fail(DSP_E_ARG, "lagrange: half_len must be in [1,8]");
    if (P < 0 || (P && (!x || !positions || !out_c128))) return fail(DSP_E_ARG, "lagrange: NULL argument");
    if (P == 0) return DSP_OK;
    const size_t eb = dtype == DSP_C128 ? 16 : 8;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf d_x, d_p, d_o;
    HIP_TRY(d_x.alloc((size_t)n * eb)); HIP_TRY(d_p.alloc((size_t)P * 8)); HIP_TRY(d_o.alloc((size_t)P * 16));
    HIP_TRY(link_h2d(c, d_x.p, x, (size_t)n * eb));
    HIP_TRY(link_h2d(c, d_p.p, positions, (size_t)P * 8));
    HIP_TRY(launch_lagrange_resample(d_x.p, dtype == DSP_C128, (int)n, d_p.as<double>(), P, half_len, d_o.p, c->stream));
    HIP_TRY(link_d2h(c, out_c128, d_o.p, (size_t)P * 16));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

// ---------------------------------------------------------------------------
// Segment-parallel mode (segment_kernels.hip)
int64_t dsp_capture_segments(int64_t n_samples, int64_t seg_len)
{
    if (n_samples <= 0 || seg_len <= 0) return 0;
    return (n_samples + seg_len - 1) / seg_len;
}

int dsp_capture_validate(const dsp_capture_args *a)
{
    if (!a) return fail(DSP_E_ARG, "capture: args is NULL");
    if (!a->x || !a->dec) return fail(DSP_E_ARG, "capture: x / dec is NULL");
    if (a->in_kind != DSP_I16 && a->in_kind != DSP_C64) return fail(DSP_E_ARG, "capture: in_kind must be DSP_I16 or DSP_C64");
    if (a->sps < 1 || a->sps > 64) return fail(DSP_E_ARG, "capture: sps out of range");
    if (a->n_samples < 2 * a->sps) return fail(DSP_E_ARG, "capture: too few samples");
    if (a->seg_len < 16 * a->sps || a->seg_len % a->sps) return fail(DSP_E_ARG, "capture: seg_len must be a multiple of sps, >= 16 symbols");
    if (a->warmup < 0) return fail(DSP_E_ARG, "capture: negative warm-up");
    if (a->acquire < 0 || a->acquire > 2) return fail(DSP_E_ARG, "capture: acquire is 0 (mu0), 1 (sample grid) or 2 (fractional)");
    if (a->h && (a->n_taps < 1 || a->n_taps > 4096)) return fail(DSP_E_ARG, "capture: bad n_taps");
    if (!(a->y > 0.0)) return fail(DSP_E_ARG, "capture: y must be > 0");
    if (dsp_capture_segments(a->n_samples, a->seg_len) > (1 << 24)) return fail(DSP_E_ARG, "capture: too many segments");
    // A window's symbol positions are kept as float32 (exact integers up to 2^24, spacing <= 0.5
    // sample below 2^23): the stitch decides on half a symbol, so a window may span at most 2^23
    // samples.  (The windows of a capture are ~3e4 samples; this bounds seg_len + warm-up.)
    const int64_t Nw = a->seg_len + a->warmup * (int64_t)a->sps + 4 * (int64_t)a->sps + 8;
    if (a->warmup >= (1LL << 30) || Nw > (1LL << 23))
        return fail(DSP_E_ARG, "capture: window of %lld samples (seg_len + warm-up) exceeds 2^23", (long long)Nw);
    return DSP_OK;
}

int dsp_sync_capture_segmented_dev(dsp_ctx *c, const dsp_capture_args *a)
{
    if (!c || !a) return fail(DSP_E_ARG, "ctx/args is NULL");
    {
        const int rc = dsp_capture_validate(a);
        if (rc != DSP_OK) return rc;
    }
    const int64_t n = a->n_samples, seg = a->seg_len;
    const int S = (int)dsp_capture_segments(n, seg), sps = a->sps;
    const int64_t lead = a->warmup * sps, tail = 4 * (int64_t)sps + 8;
    const int64_t Nw = seg + lead + tail;                  // longest window
    const int64_t rows = dsp_tile_rows(Nw), out_rows = Nw / sps + 8, ntiles = (S + 63) / 64;
    hipStream_t st = c->stream;
    HIP_TRY(hipSetDevice(c->device));

    // every per-window array comes out of one block of the context's scratch (slot 5), the large
    // tiles have slots of their own: no hipMalloc / hipFree in a call once the context has grown
    ScratchRef d_starts, d_lens, d_nout, d_status, d_j0, d_j1, d_rot, d_pl, d_off;
    ScratchRef d_tiles, d_mu, d_pos, d_dec, d_freq;
    Bump small;
    const int Sa = (int)ntiles * 64;                        // slots: the per-window arrays hold whole tiles
    small.cap = (size_t)64 * 1024 + (size_t)Sa * 96;
    HIP_TRY(ctx_scratch(c, 5, small.cap, (void **)&small.base));
    d_starts.p = small.take((size_t)Sa * 8); d_lens.p = small.take((size_t)Sa * 4);
    // window tiles [ntiles][rows][64], or (by_segment below) lane-major segment tiles [ntiles][seg][64] + the
    // rows behind the last one + tile 0's windows: the buffer serves either
    const int64_t pad_rows = rows - lead - seg > 0 ? rows - lead - seg : 0;
    {
        const size_t win_bytes = (size_t)ntiles * rows * 64 * 8;
        const size_t seg_bytes = ((size_t)ntiles * seg + pad_rows + rows) * 64 * 8;
        HIP_TRY(ctx_scratch(c, 0, win_bytes > seg_bytes ? win_bytes : seg_bytes, &d_tiles.p));
    }
    if (a->mu) HIP_TRY(ctx_scratch(c, 1, (size_t)ntiles * out_rows * 64 * 4, &d_mu.p));
    HIP_TRY(ctx_scratch(c, 2, (size_t)ntiles * out_rows * 64 * 4, &d_pos.p));
    HIP_TRY(ctx_scratch(c, 3, (size_t)ntiles * out_rows * 64, &d_dec.p));
    if (a->freq) HIP_TRY(ctx_scratch(c, 4, (size_t)ntiles * out_rows * 64 * 4, &d_freq.p));
    // j0 | j1 | rot | n_out lie back to back: one copy brings seg_info home
    d_j0.p = small.take((size_t)Sa * 16);
    d_j1.p = d_j0.as<int>() + Sa; d_rot.p = d_j0.as<int>() + 2 * (size_t)Sa; d_nout.p = d_j0.as<int>() + 3 * (size_t)Sa;
    d_status.p = small.take((size_t)Sa * 4);
    d_pl.p = small.take((size_t)Sa * 8); d_off.p = small.take((size_t)(Sa + 1) * 8);
    // ---- windows -> tiles (optionally through the matched filter) ------------------------
    ScratchRef d_peak, d_c64, d_ws, d_inv, d_lo, d_hi, d_part, d_mu0;
    d_peak.p = small.take(16);
    d_lo.p = small.take((size_t)Sa * 4); d_hi.p = small.take((size_t)Sa * 4);
    d_inv.p = small.take((size_t)Sa * 8);
    d_part.p = small.take((size_t)1024 * 32);
    d_mu0.p = small.take((size_t)Sa * 8);
    if (small.off > small.cap) return fail(DSP_E_NOMEM, "capture: scratch block too small");
    // the filter pass writes every row of every lane (zeros past a window's end and in the idle lanes
    // of the last tile: its chunks cover ceil(Nw / CH) * CH >= rows); only the copy path needs clearing
    if (!a->h) HIP_TRY(hipMemsetAsync(d_tiles.p, 0, (size_t)ntiles * rows * 64 * 8, st));
    const float *in_scale = nullptr;
    if (a->in_kind == DSP_I16) {
        if (!a->h) HIP_TRY(ctx_scratch(c, 7, (size_t)n * 8, &d_c64.p));
        HIP_TRY(launch_iq_i16(a->x, n, a->normalize, d_peak.as<unsigned>(), d_peak.as<float>() + 1,
                              a->h ? nullptr : d_c64.p, st));
        in_scale = d_peak.as<float>() + 1;
    }
    // (a) the stream cut into its segments, each filtered once by the persistent strip kernel into lane-major
    //     segment tiles: window (tile t, lane l) = segment l * ntiles + t, its warm-up rows are the rows in
    //     front of tile t (the same lane's previous segment) and its tail the rows behind it ...
    FirParams Q;
    memset(&Q, 0, sizeof(Q));
    bool by_segment = false;
    if (a->h && c->fir_stream == 2 && c->lead_copy == 0 && S >= 64 && lead <= seg &&
        tail <= seg && seg < (1 << 30) && fir_plan(a->in_kind, a->n_taps, (int)seg, Q, true) && Q.stream) {
        Q.stream_mode = 1; Q.strip_ok = 1; Q.n_cu = c->n_cu;
        Q.src = a->x; Q.n_total = n; Q.ld = seg; Q.len_all = (int)seg; Q.B = Sa;   // whole tiles: the segments behind the stream's end are empty
        Q.h = a->h; Q.M = a->n_taps; Q.start = (a->n_taps - 1) / 2; Q.h_pad = c->fir_taps;
        Q.in_scale = in_scale; Q.dst = d_tiles.p; Q.rows = (int)seg;
        by_segment = fir_stream_mode_plan(Q, a->in_kind);
    }
    const int map_tiles = by_segment ? (int)ntiles : 0;
    const int n_slots = by_segment ? Sa : S;                // windows the recurrence kernel sees (the empty ones have length 0)
    // the recurrence kernel's view of the windows: rows of tile t start x_tile_rows * t rows after win_x
    const void *win_x = by_segment ? (const void *)((const char *)d_tiles.p - (ptrdiff_t)lead * 64 * 8) : d_tiles.p;
    void *win_tile0 = by_segment ? (void *)((char *)d_tiles.p + ((size_t)ntiles * seg + pad_rows) * 64 * 8) : nullptr;
    const int win_stride = by_segment ? (int)seg : 0;
    HIP_TRY(launch_seg_windows(S, n, seg, lead, tail, d_starts.as<int64_t>(), d_lens.as<int>(), d_lo.as<int>(),
                               d_hi.as<int>(), st, map_tiles));
    const double *d_scale = nullptr;
    if (a->h) {
        // ONE filter pass over the overlapping windows: unscaled fl32(y) into the tiles, statistics
        // over each window's own samples only (the windows' shares partition the stream), merged
        // into the np.std of the whole filtered stream; 1/std is applied by the recurrence kernel
        if (by_segment) {
            // a last segment cut short by the stream's end is left out of the strips' statistics and
            // counted by a pass over the last window alone (statistics only, its own samples)
            FirParams W;
            memset(&W, 0, sizeof(W));
            const bool cut = n < (int64_t)S * seg;
            if (cut && !fir_plan(a->in_kind, a->n_taps, (int)Nw, W, true))
                return fail(DSP_E_ARG, "capture: %d taps are too many for the LDS-tiled filter", a->n_taps);
            HIP_TRY(ctx_scratch(c, 6, ((size_t)Sa * (Q.nchunk + 8) + (cut ? W.nchunk : 0)) * 32, &d_ws.p));
            Q.stats = d_ws.as<double>();
            HIP_TRY(launch_fir_tile(Q, a->in_kind, st));
            if (cut) {
                W.h_pad = c->fir_taps; W.src = a->x; W.n_total = n; W.len_all = (int)Nw; W.B = 1;
                const int last = seg_slot(S - 1, map_tiles);
                W.starts = d_starts.as<int64_t>() + last; W.lens = d_lens.as<int>() + last;
                W.stat_lo = d_lo.as<int>() + last; W.stat_hi = d_hi.as<int>() + last;
                W.h = a->h; W.M = a->n_taps; W.start = (a->n_taps - 1) / 2; W.in_scale = in_scale;
                W.dst = nullptr; W.rows = (int)rows; W.stats = Q.stats + (size_t)Sa * Q.nrec * 4;
                HIP_TRY(launch_fir_tile(W, a->in_kind, st));
                Q.nrec_extra = W.nrec;
            }
            HIP_TRY(launch_seg_edge_tiles(d_tiles.p, map_tiles, seg, n, d_starts.as<int64_t>(), d_lens.as<int>(), win_tile0,
                                          (int)rows, (int)pad_rows, st));
        } else {
            // (b) ... or the overlapping windows themselves, chunk by chunk
            memset(&Q, 0, sizeof(Q));
            if (!fir_plan(a->in_kind, a->n_taps, (int)Nw, Q, c->fir_stream != 0))
                return fail(DSP_E_ARG, "capture: %d taps are too many for the LDS-tiled filter", a->n_taps);
            Q.h_pad = c->fir_taps;
            HIP_TRY(ctx_scratch(c, 6, (size_t)S * Q.nchunk * 32, &d_ws.p));
            Q.src = a->x; Q.starts = d_starts.as<int64_t>(); Q.n_total = n; Q.lens = d_lens.as<int>();
            Q.stat_lo = d_lo.as<int>(); Q.stat_hi = d_hi.as<int>();
            Q.len_all = (int)Nw; Q.B = S; Q.h = a->h; Q.M = a->n_taps; Q.start = (a->n_taps - 1) / 2;
            Q.in_scale = in_scale; Q.dst = d_tiles.p; Q.rows = (int)rows; Q.stats = d_ws.as<double>();
            // the warm-up rows of a window are rows its predecessor filters anyway: skip them here (every
            // tile but the first) and copy them over afterwards
            const bool share_lead = lead > 0 && lead <= seg && S > 64 && c->lead_copy != 2;
            Q.skip_rows = share_lead ? (int)lead : 0;
            Q.feed_lead = share_lead && c->lead_copy == 0 ? 1 : 0;     // the predecessor's workgroup stores them ...
            HIP_TRY(launch_fir_tile(Q, a->in_kind, st));
            if (share_lead && c->lead_copy == 1)                      // ... or a copy kernel does afterwards
                HIP_TRY(launch_seg_lead_copy(d_tiles.p, (int)rows, d_starts.as<int64_t>(), S, (int)lead, st));
        }
        HIP_TRY(launch_fir_stats_global(Q.stats, (int64_t)(by_segment ? Sa : S) * Q.nrec + Q.nrec_extra, n_slots, nullptr, d_inv.as<double>(),
                                        d_part.as<double>(), 1024, st));
        d_scale = d_inv.as<double>();
    } else {
        const void *src = a->in_kind == DSP_I16 ? d_c64.p : a->x;
        HIP_TRY(launch_tile_from_bursts(src, 0, d_lens.as<int>(), (int)Nw, S, d_tiles.p, (int)rows, 8, st,
                                        d_starts.as<int64_t>()));
    }

    // ---- initial timing phase per window (strobe energy), unless the caller fixes mu0 ------------
    if (a->acquire) {
        HIP_TRY(launch_seg_phase(win_x, (int)rows, d_lens.as<int>(), n_slots, sps,
                                 a->acquire == 2 ? DSP_ACQUIRE_FRAC_SYMBOLS : DSP_ACQUIRE_SYMBOLS,
                                 d_mu0.as<double>(), st, win_stride, win_tile0, a->acquire == 2));
    }

    // ---- the batched recurrence over the windows -------------------------------------------
    dsp_sync_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.mu0_per = a->acquire ? d_mu0.as<double>() : nullptr;
    sa.x = win_x; sa.dtype = DSP_C64; sa.variant = DSP_GARDNER2; sa.B = n_slots; sa.N = Nw; sa.in_rows = rows;
    sa.lens = d_lens.as<int32_t>(); sa.scale = d_scale; sa.sps = sps; sa.alpha = a->alpha; sa.mu0 = a->mu0; sa.do_fll = 1;
    sa.Bn = a->Bn; sa.y = a->y; sa.mu = a->mu ? d_mu.p : nullptr; sa.pos = d_pos.p; sa.dec = d_dec.as<uint8_t>(); sa.freq = a->freq ? d_freq.p : nullptr;
    sa.out_rows = out_rows; sa.n_out = d_nout.as<int32_t>(); sa.status = d_status.as<int32_t>();
    {
        int rc = sync_tiles_run(c, &sa, win_stride, win_tile0, 1);
        if (rc != DSP_OK) return rc;
    }

    // ---- stitch ----------------------------------------------------------------------------
    SegParams G;
    memset(&G, 0, sizeof(G));
    G.S = S; G.sps = sps; G.out_rows = (int)out_rows; G.n_total = n; G.seg_len = seg; G.map_tiles = map_tiles; G.n_slots = n_slots;
    G.starts = d_starts.as<int64_t>(); G.n_out = d_nout.as<int>(); G.mu = a->mu ? d_mu.p : nullptr; G.pos = d_pos.p; G.dec = d_dec.as<uint8_t>();
    G.freq = a->freq ? d_freq.p : nullptr;
    G.j0 = d_j0.as<int>(); G.j1 = d_j1.as<int>(); G.p_last = d_pl.as<double>(); G.rot = d_rot.as<int>();
    G.off = d_off.as<int64_t>();
    int vote = (int)(a->warmup / 2);
    if (vote > 256) vote = 256;
    HIP_TRY(launch_seg_stitch(G, vote, st));
    HIP_TRY(launch_seg_gather(G, a->dec, a->mu, a->freq, a->capacity, st));
    // the count and seg_info come home through the context's page-locked block: [total | j0 | j1 | rot | n_out]
    int64_t *h_total = nullptr;
    HIP_TRY(ctx_pinned(c, 64 + (a->seg_info ? (size_t)Sa * 16 : 0), (void **)&h_total));
    HIP_TRY(hipMemcpyAsync(h_total, d_off.as<int64_t>() + n_slots, 8, hipMemcpyDeviceToHost, st));
    if (a->seg_info) {
        const int32_t *info = (const int32_t *)((const char *)h_total + 64);      // Sa slots each
        HIP_TRY(hipMemcpyAsync((void *)info, d_j0.p, (size_t)Sa * 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t s = 0; s < (size_t)S; s++) {
            const size_t at = (size_t)seg_slot((int)s, map_tiles);           // the device arrays are in slot order
            a->seg_info[4 * s] = info[at]; a->seg_info[4 * s + 1] = info[Sa + at];
            a->seg_info[4 * s + 2] = info[2 * (size_t)Sa + at]; a->seg_info[4 * s + 3] = info[3 * (size_t)Sa + at];
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t total = *h_total;
    if (a->n_symbols) *a->n_symbols = total;
    if (total > a->capacity)
        return fail(DSP_E_CAPACITY, "capture: %lld symbols, capacity %lld", (long long)total, (long long)a->capacity);
    return DSP_OK;
}

int dsp_sync_capture_segmented(dsp_ctx *c, const dsp_capture_args *a)
{
    if (!c || !a) return fail(DSP_E_ARG, "ctx/args is NULL");
    if (!a->x || !a->dec || a->n_samples <= 0 || a->capacity <= 0) return fail(DSP_E_ARG, "capture: bad arguments");
    if (a->in_kind != DSP_I16 && a->in_kind != DSP_C64) return fail(DSP_E_ARG, "capture: in_kind must be DSP_I16 or DSP_C64");
    HIP_TRY(hipSetDevice(c->device));
    const size_t ie = a->in_kind == DSP_I16 ? 4 : 8;
    DevBuf d_x, d_h, d_dec, d_mu, d_freq;
    HIP_TRY(d_x.alloc((size_t)a->n_samples * ie));
    HIP_TRY(link_h2d(c, d_x.p, a->x, (size_t)a->n_samples * ie));
    dsp_capture_args d = *a;
    d.x = d_x.p;
    if (a->h) {
        if (a->n_taps < 1) return fail(DSP_E_ARG, "capture: bad n_taps");
        HIP_TRY(d_h.alloc((size_t)a->n_taps * 8));
        HIP_TRY(hipMemcpyAsync(d_h.p, a->h, (size_t)a->n_taps * 8, hipMemcpyHostToDevice, c->stream));
        d.h = d_h.as<double>();
    }
    HIP_TRY(d_dec.alloc((size_t)a->capacity)); d.dec = d_dec.as<uint8_t>();
    if (a->mu) { HIP_TRY(d_mu.alloc((size_t)a->capacity * 4)); d.mu = d_mu.as<float>(); }
    if (a->freq) { HIP_TRY(d_freq.alloc((size_t)a->capacity * 4)); d.freq = d_freq.as<float>(); }
    int64_t total = 0;
    d.n_symbols = &total;
    int rc = dsp_sync_capture_segmented_dev(c, &d);
    if (a->n_symbols) *a->n_symbols = total;
    if (rc != DSP_OK) return rc;
    HIP_TRY(link_d2h(c, a->dec, d_dec.p, (size_t)total));
    if (a->mu) HIP_TRY(link_d2h(c, a->mu, d_mu.p, (size_t)total * 4));
    if (a->freq) HIP_TRY(link_d2h(c, a->freq, d_freq.p, (size_t)total * 4));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DSP_OK;
}

int dsp_synth_bursts_dev(dsp_ctx *c, uint64_t seed, int64_t b0, int64_t B, int n_sym, int sps,
                         const double *d_taps, int n_taps, double esn0_db, void *d_out_c64)
{
    if (!c || !d_taps || !d_out_c64) return fail(DSP_E_ARG, "synth: NULL argument");
    if (B < 0 || B > 65535 || n_sym < 2 || sps < 1 || n_taps < 1 || n_taps > 512)
        return fail(DSP_E_ARG, "synth: bad B / n_sym / sps / n_taps");
    const float sigma = (float)sqrt(pow(10.0, -esn0_db / 10.0));
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_synth(seed, b0, (int)B, n_sym, sps, d_taps, n_taps, sigma, d_out_c64, c->stream));
    return DSP_OK;
}

int dsp_synth_stream_dev(dsp_ctx *c, uint64_t seed, int64_t n_samples, int sps, const double *d_taps,
                         int n_taps, double esn0_db, double cfo, double tau0, double drift, int out_kind,
                         double amplitude, void *d_out)
{
    if (!c || !d_taps || !d_out) return fail(DSP_E_ARG, "synth: NULL argument");
    if (n_samples < 1 || sps < 1 || n_taps < 1 || n_taps > 512) return fail(DSP_E_ARG, "synth: bad n / sps / n_taps");
    if (out_kind != DSP_I16 && out_kind != DSP_C64) return fail(DSP_E_ARG, "synth: out_kind must be DSP_I16 or DSP_C64");
    const float sigma = (float)sqrt(pow(10.0, -esn0_db / 10.0));
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_synth_stream(seed, n_samples, sps, d_taps, n_taps, sigma, cfo, tau0, drift,
                                out_kind == DSP_I16, (float)amplitude, d_out, c->stream));
    return DSP_OK;
}

}  // extern "C"

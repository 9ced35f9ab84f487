// Host <-> device link rates for the host-pointer entry points: what a caller with plain
// (pageable) arrays can get, and what each way of getting there costs.
//   hipcc -O3 --offload-arch=gfx950 -pthread tools/ubench/host_link.hip -o /tmp/host_link && /tmp/host_link [MiB=512]
// Prints GB/s for: pageable hipMemcpy both ways, page-locked hipMemcpy both ways, hipHostRegister +
// copy + unregister, and a T-thread memcpy into / out of a page-locked ring whose slots go over the
// link with hipMemcpyAsync while the next slot is being filled (T = 1, 2, 4, 8).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// T threads copy [src, src + n) to dst in interleaved 1 MiB pieces
static void par_copy(char *dst, const char *src, size_t n, int T)
{
    if (T <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t piece = (n / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; t++) {
        size_t a = (size_t)t * piece, b = a + piece > n ? n : a + piece;
        if (a >= n) break;
        th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    for (auto &x : th) x.join();
}

int main(int argc, char **argv)
{
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 512) << 20;
    char *pg = (char *)aligned_alloc(4096, n), *pin = nullptr, *dev = nullptr;
    memset(pg, 1, n);
    CK(hipMalloc(&dev, n));
    CK(hipHostMalloc(&pin, n, hipHostMallocDefault));
    memset(pin, 2, n);
    hipStream_t s; CK(hipStreamCreate(&s));
    auto rate = [&](double t) { return n / t / 1e9; };
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipMemcpy(dev, pg, n, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(pg, dev, n, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable  h2d %6.1f GB/s   d2h %6.1f GB/s\n", rate(t1 - t0), rate(t2 - t1));
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipMemcpy(dev, pin, n, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(pin, dev, n, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pinned    h2d %6.1f GB/s   d2h %6.1f GB/s\n", rate(t1 - t0), rate(t2 - t1));
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        hipError_t e = hipHostRegister(pg, n, hipHostRegisterDefault);
        if (e != hipSuccess) { printf("hipHostRegister: %s (skipped)\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
        double t1 = now();
        CK(hipMemcpy(dev, pg, n, hipMemcpyHostToDevice)); double t2 = now();
        CK(hipHostUnregister(pg)); double t3 = now();
        printf("register %6.1f ms  copy %6.1f GB/s  unregister %6.1f ms  -> %6.1f GB/s all in\n", (t1 - t0) * 1e3,
               rate(t2 - t1), (t3 - t2) * 1e3, rate(t3 - t0));
    }
    // staged through a page-locked ring: slot k is filled by T threads while slot k-1 is on the link
    const size_t slot = 16u << 20;
    const int NS = 4;
    hipEvent_t ev[NS];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int T : {1, 2, 4, 8}) {
        double t0 = now();
        int k = 0;
        for (size_t off = 0; off < n; off += slot, k++) {
            const size_t m = off + slot > n ? n - off : slot;
            char *sl = pin + (size_t)(k % NS) * slot;
            if (k >= NS) CK(hipEventSynchronize(ev[k % NS]));
            par_copy(sl, pg + off, m, T);
            CK(hipMemcpyAsync(dev + off, sl, m, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[k % NS], s));
        }
        CK(hipStreamSynchronize(s));
        double t1 = now();
        // the other way: slot k comes over the link while slot k-1 is copied out
        k = 0;
        size_t prev_off = 0, prev_m = 0;
        for (size_t off = 0; off < n; off += slot, k++) {
            const size_t m = off + slot > n ? n - off : slot;
            char *sl = pin + (size_t)(k % NS) * slot;
            CK(hipMemcpyAsync(sl, dev + off, m, hipMemcpyDeviceToHost, s));
            CK(hipEventRecord(ev[k % NS], s));
            if (k > 0) {
                CK(hipEventSynchronize(ev[(k - 1) % NS]));
                par_copy(pg + prev_off, pin + (size_t)((k - 1) % NS) * slot, prev_m, T);
            }
            prev_off = off; prev_m = m;
        }
        CK(hipEventSynchronize(ev[(k - 1) % NS]));
        par_copy(pg + prev_off, pin + (size_t)((k - 1) % NS) * slot, prev_m, T);
        double t2 = now();
        printf("staged T=%d  h2d %6.1f GB/s   d2h %6.1f GB/s\n", T, rate(t1 - t0), rate(t2 - t1));
    }
    // plain host memcpy rate for scale
    {
        char *q = (char *)aligned_alloc(4096, n);
        memset(q, 3, n);
        for (int T : {1, 4, 8}) { double t0 = now(); par_copy(q, pg, n, T); double t1 = now(); printf("host memcpy T=%d %6.1f GB/s\n", T, rate(t1 - t0)); }
        free(q);
    }
    return 0;
}

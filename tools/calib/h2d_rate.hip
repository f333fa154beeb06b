// h2d_rate.hip -- how fast can 1.12 GB of PAGEABLE host memory (a caller's Vec<Triangle>) reach HBM?
//   a) hipMemcpy straight from the pageable buffer
//   b) hipHostRegister (pin in place) + hipMemcpy + hipHostUnregister
//   c) staged: worker threads memcpy chunks into a ring of pinned buffers, hipMemcpyAsync from there
// build: hipcc -O2 --offload-arch=gfx950 tools/calib/h2d_rate.hip -o /tmp/h2d_rate -lpthread ; run: /tmp/h2d_rate [MB]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 1120) << 20;
    char *h = (char *)malloc(bytes);
    {   // first touch on several threads, like a generator would
        std::vector<std::thread> th;
        for (int t = 0; t < 8; t++) th.emplace_back([=] { const size_t b = bytes / 8 * t, e = t == 7 ? bytes : bytes / 8 * (t + 1); memset(h + b, t + 1, e - b); });
        for (auto &x : th) x.join();
    }
    void *d = nullptr;
    double t0 = now();
    CK(hipMalloc(&d, bytes));
    printf("hipMalloc %.1f MB: %.2f ms\n", bytes / 1048576.0, (now() - t0) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
        t0 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double dt = now() - t0;
        printf("a) pageable hipMemcpy: %.1f ms  %.1f GB/s\n", dt * 1e3, bytes / dt / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        t0 = now();
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("b) register %.1f ms + copy %.1f ms (%.1f GB/s) + unregister %.1f ms = %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
    }
    for (int nthr : {1, 2, 4, 8}) for (size_t chunk_mb : {4, 16}) {
        const size_t chunk = chunk_mb << 20;
        const int ring = 4;
        t0 = now();
        char *pin[ring];
        hipEvent_t ev[ring];
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int i = 0; i < ring; i++) { CK(hipHostMalloc((void **)&pin[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
        double t1 = now();
        const size_t n_chunks = (bytes + chunk - 1) / chunk;
        for (size_t c = 0; c < n_chunks; c++) {
            const int slot = (int)(c % ring);
            if (c >= (size_t)ring) CK(hipEventSynchronize(ev[slot]));
            const size_t off = c * chunk, len = off + chunk <= bytes ? chunk : bytes - off;
            if (nthr == 1) memcpy(pin[slot], h + off, len);
            else {
                std::vector<std::thread> th;
                for (int t = 0; t < nthr; t++) th.emplace_back([=] { const size_t b = len / nthr * t, e = t == nthr - 1 ? len : len / nthr * (t + 1); memcpy(pin[slot] + b, h + off + b, e - b); });
                for (auto &x : th) x.join();
            }
            CK(hipMemcpyAsync((char *)d + off, pin[slot], len, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[slot], s));
        }
        CK(hipStreamSynchronize(s));
        double t2 = now();
        for (int i = 0; i < ring; i++) { CK(hipHostFree(pin[i])); CK(hipEventDestroy(ev[i])); }
        CK(hipStreamDestroy(s));
        double t3 = now();
        printf("c) staged %d thr, %zu MB chunks: setup %.1f ms + copy %.1f ms (%.1f GB/s) + teardown %.1f ms = %.1f ms\n", nthr, chunk_mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
    }
    // persistent worker threads variant is what the library would use; the spawn cost above is ~50 us per chunk per thread
    t0 = now();
    CK(hipFree(d));
    printf("hipFree: %.2f ms\n", (now() - t0) * 1e3);
    // big allocations: cost of hipMalloc + hipFree for the builder's workspace
    for (size_t mb : {64, 512, 2048}) {
        void *p = nullptr; t0 = now(); CK(hipMalloc(&p, mb << 20)); double t1 = now(); CK(hipMemset(p, 0, mb << 20)); CK(hipDeviceSynchronize()); double t2 = now(); CK(hipFree(p)); double t3 = now();
        printf("hipMalloc %zu MB: %.2f ms, first memset %.2f ms, hipFree %.2f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    free(h);
    return 0;
}

// level_sync.hip -- what does one level boundary of the level-synchronous BVH builder cost, and what would the alternatives cost?
//   Every level: a counter-row memset + one kernel on each of four blocking streams (a `spin` of WORK_US microseconds stands in for
//   the level's kernels), then the host must learn the level's counters before it can size the level after it.
//   a) today: blocking hipMemcpy D2H of the counters (also the level's barrier)
//   b) a one-wave kernel on the null stream (joins the four streams) copies the counters into PINNED HOST memory and raises a
//      flag; the host polls the flag -- still one host round trip per level, without the copy engine
//   c) like b, but the host runs ONE LEVEL AHEAD: it enqueues level L (grids sized from bounds) as soon as level L-2's snapshot has
//      landed, so the GPU never waits for the host
//   d) no host at all: everything enqueued up front (the floor: the cost of the four-stream join alone)
//   e) no join packets at all: every stream ends its level with a one-thread `mark` kernel; the mark that arrives last (a device
//      atomic) writes the counters and the flag into pinned host memory; the host polls, then launches the next level on idle streams
// build: hipcc -O2 --offload-arch=gfx950 tools/calib/level_sync.hip -o /tmp/level_sync ; run: /tmp/level_sync [levels] [work_us]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Ctrl { uint32_t v[64]; };
struct Snap { uint32_t v[64]; uint32_t flag; uint32_t pad[63]; };

__global__ void spin(Ctrl *c, int slot, long long ticks) {      // 100 MHz constant clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&c->v[slot], 1u);
}
__global__ void snapshot(const Ctrl *c, volatile Snap *out, uint32_t level) {
    if (threadIdx.x < 64) out->v[threadIdx.x] = c->v[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) out->flag = level + 1u;
}

__global__ void mark(const Ctrl *c, uint32_t *arrived, uint32_t expect, volatile Snap *out, uint32_t level) {
    __shared__ uint32_t last;
    if (threadIdx.x == 0) { __threadfence(); last = atomicAdd(arrived, 1u) == expect - 1u; }
    __syncthreads();
    if (!last) return;
    __threadfence();
    out->v[threadIdx.x] = __hip_atomic_load(&c->v[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *arrived = 0u; out->flag = level + 1u; }
}

int main(int argc, char **argv) {
    const int levels = argc > 1 ? atoi(argv[1]) : 29;
    const int work_us = argc > 2 ? atoi(argv[2]) : 100;
    const long long ticks = (long long)work_us * 100;
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreate(&x));
    Ctrl *d = nullptr, *h = nullptr;
    Snap *snaps = nullptr;
    CK(hipMalloc((void **)&d, sizeof(Ctrl)));
    CK(hipHostMalloc((void **)&h, sizeof(Ctrl), hipHostMallocDefault));
    CK(hipHostMalloc((void **)&snaps, sizeof(Snap) * (size_t)(levels + 2), hipHostMallocDefault));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto level_work = [&](int) {
        (void)hipMemsetAsync(&d->v[32], 0, 64, s[3]);
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s[k], d, k, ticks);
    };
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        double t0;
        // a)
        CK(hipMemset(d, 0, sizeof(Ctrl)));
        CK(hipDeviceSynchronize());
        t0 = now();
        CK(hipEventRecord(e0, nullptr));
        for (int l = 0; l < levels; l++) { level_work(l); CK(hipMemcpy(h, d, sizeof(Ctrl), hipMemcpyDeviceToHost)); }
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("a) blocking D2H per level:        %7.3f ms on the GPU, %7.3f ms wall -> %5.1f us per level beyond the %d us of work (check %u)\n", ms, (now() - t0) * 1e3, (ms * 1e3 - (double)levels * work_us) / levels, work_us, h->v[0]);
        // b)
        CK(hipMemset(d, 0, sizeof(Ctrl)));
        memset(snaps, 0, sizeof(Snap) * (size_t)(levels + 2));
        CK(hipDeviceSynchronize());
        t0 = now();
        CK(hipEventRecord(e0, nullptr));
        for (int l = 0; l < levels; l++) {
            level_work(l);
            hipLaunchKernelGGL(snapshot, dim3(1), dim3(64), 0, nullptr, d, snaps + l, (uint32_t)l);
            while (((volatile Snap *)snaps)[l].flag != (uint32_t)l + 1u) {}
        }
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("b) snapshot kernel + host poll:   %7.3f ms on the GPU, %7.3f ms wall -> %5.1f us per level (check %u)\n", ms, (now() - t0) * 1e3, (ms * 1e3 - (double)levels * work_us) / levels, snaps[levels - 1].v[0]);
        // c)
        CK(hipMemset(d, 0, sizeof(Ctrl)));
        memset(snaps, 0, sizeof(Snap) * (size_t)(levels + 2));
        CK(hipDeviceSynchronize());
        t0 = now();
        CK(hipEventRecord(e0, nullptr));
        for (int l = 0; l < levels; l++) {
            if (l >= 2) while (((volatile Snap *)snaps)[l - 2].flag != (uint32_t)(l - 2) + 1u) {}
            level_work(l);
            hipLaunchKernelGGL(snapshot, dim3(1), dim3(64), 0, nullptr, d, snaps + l, (uint32_t)l);
        }
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("c) snapshot, host one level ahead: %7.3f ms on the GPU, %7.3f ms wall -> %5.1f us per level (check %u)\n", ms, (now() - t0) * 1e3, (ms * 1e3 - (double)levels * work_us) / levels, snaps[levels - 1].v[0]);
        // d)
        CK(hipMemset(d, 0, sizeof(Ctrl)));
        CK(hipDeviceSynchronize());
        t0 = now();
        CK(hipEventRecord(e0, nullptr));
        for (int l = 0; l < levels; l++) {
            level_work(l);
            hipLaunchKernelGGL(snapshot, dim3(1), dim3(64), 0, nullptr, d, snaps + l, (uint32_t)l);
        }
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("d) everything enqueued up front:  %7.3f ms on the GPU, %7.3f ms wall -> %5.1f us per level\n", ms, (now() - t0) * 1e3, (ms * 1e3 - (double)levels * work_us) / levels);
    }
    uint32_t *d_arr = nullptr;
    CK(hipMalloc((void **)&d_arr, 4));
    CK(hipMemset(d_arr, 0, 4));
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        CK(hipMemset(d, 0, sizeof(Ctrl)));
        memset(snaps, 0, sizeof(Snap) * (size_t)(levels + 2));
        CK(hipDeviceSynchronize());
        double t0 = now();
        CK(hipEventRecord(e0, s[0]));
        for (int l = 0; l < levels; l++) {
            level_work(l);
            for (int k = 0; k < 4; k++) hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, s[k], d, d_arr, 4u, snaps + l, (uint32_t)l);
            while (((volatile Snap *)snaps)[l].flag != (uint32_t)l + 1u) {}
        }
        CK(hipEventRecord(e1, s[0]));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("e) per-stream mark, last one snapshots, host polls: %7.3f ms on the GPU, %7.3f ms wall -> %5.1f us per level (check %u %u %u %u)\n", ms, (now() - t0) * 1e3,
               (ms * 1e3 - (double)levels * work_us) / levels, snaps[levels - 1].v[0], snaps[levels - 1].v[1], snaps[levels - 1].v[2], snaps[levels - 1].v[3]);
    }
    return 0;
}

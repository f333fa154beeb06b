// gather_calib.hip -- calibration of rocprofv3's memory-side counters (FETCH_SIZE, TCC_EA0_RDREQ*, TCC_BUBBLE, TCC_MISS)
// for the access pattern of the path-tracing kernel: every lane gathers ONE 64-byte record (4 x 16-B loads) from a
// different, random place of a table far larger than L2 + Infinity Cache.  MI355X_MICROARCH.md calibrates FETCH_SIZE only
// for wide coalesced streaming reads (it reports 1/2 of the bytes there) and says other widths are uncalibrated.
//
// Kernels (each launched once, so a rocprofv3 pass lists one row per kernel; names are the row keys):
//   calib_stream        coalesced 16 B/lane read of `bytes` bytes                         -> the guide's reference case
//   calib_gather64      N lanes x one 64-B-aligned 64-B record at a random index          -> the traversal step (inner pair)
//   calib_gather128     N lanes x one 128-B-aligned 128-B record (8 x 16-B loads)         -> does a request move 64 or 128 B?
//   calib_gather48      N lanes x 64 B starting at a random multiple of 48 B              -> the leaf step (48-B tri_pos stride)
//   calib_gather64_hot  the 64-B gather from a 2 MiB table (L2-resident)                  -> requests that never leave L2
// Output: one JSON line per kernel with the KNOWN payload bytes and the HIP-event time; tools/calib/run_calib.sh adds the
// counters of the same launches and writes profiles/r2_fetch_calibration.csv.
//
//   hipcc -O3 --offload-arch=gfx950 -o gather_calib gather_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {           // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

extern "C" __global__ void calib_stream(const uint4 *__restrict__ tab, uint64_t n16, uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = tab[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// REC16 = 16-B loads per record; STRIDE16 = record stride in 16-B units; rec_mask = n_records - 1 (power of two)
template <int REC16, int STRIDE16>
__device__ __forceinline__ void gather_body(const uint4 *__restrict__ tab, uint64_t rec_mask, uint32_t iters, uint32_t *__restrict__ out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint64_t r = mix(gid * iters + it) & rec_mask;
        const uint4 *p = tab + r * STRIDE16;
#pragma unroll
        for (int k = 0; k < REC16; k++) {
            const uint4 v = p[k];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[gid] = acc;
}
extern "C" __global__ void calib_gather64(const uint4 *tab, uint64_t rec_mask, uint32_t iters, uint32_t *out) { gather_body<4, 4>(tab, rec_mask, iters, out); }
extern "C" __global__ void calib_gather128(const uint4 *tab, uint64_t rec_mask, uint32_t iters, uint32_t *out) { gather_body<8, 8>(tab, rec_mask, iters, out); }
extern "C" __global__ void calib_gather48(const uint4 *tab, uint64_t rec_mask, uint32_t iters, uint32_t *out) { gather_body<4, 3>(tab, rec_mask, iters, out); }
extern "C" __global__ void calib_gather64_hot(const uint4 *tab, uint64_t rec_mask, uint32_t iters, uint32_t *out) { gather_body<4, 4>(tab, rec_mask, iters, out); }

int main(int argc, char **argv) {
    const uint64_t table_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 16ull) << 30;   // GiB, power of two
    const uint32_t iters = 8;
    const int threads = 256, blocks = 65536;                                              // 2^24 lanes x 8 records = 2^27 records
    const uint64_t lanes = (uint64_t)threads * blocks, n_rec = lanes * iters;
    uint4 *tab = nullptr;
    uint32_t *out = nullptr;
    CK(hipMalloc(&tab, table_bytes + 256));
    CK(hipMalloc(&out, lanes * 4));
    CK(hipMemset(tab, 0x5a, table_bytes + 256));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char *name, uint64_t records, uint64_t payload) {
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"records\": %llu, \"payload_bytes\": %llu, \"ms\": %.3f, \"payload_GBs\": %.1f, \"Mrecords_s\": %.1f}\n", name,
               (unsigned long long)records, (unsigned long long)payload, ms, payload / (ms * 1e6), records / (ms * 1e3));
        fflush(stdout);
    };
    const uint64_t stream_bytes = 8ull << 30;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_stream, dim3(8192), dim3(256), 0, 0, tab, stream_bytes / 16, out);
    report("calib_stream", stream_bytes / 16, stream_bytes);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_gather64, dim3(blocks), dim3(threads), 0, 0, tab, table_bytes / 64 - 1, iters, out);
    report("calib_gather64", n_rec, n_rec * 64);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_gather128, dim3(blocks), dim3(threads), 0, 0, tab, table_bytes / 128 - 1, iters, out);
    report("calib_gather128", n_rec, n_rec * 128);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_gather48, dim3(blocks), dim3(threads), 0, 0, tab, table_bytes / 64 - 1, iters, out);   // 48 * 2^k <= table
    report("calib_gather48", n_rec, n_rec * 64);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_gather64_hot, dim3(blocks), dim3(threads), 0, 0, tab, (2ull << 20) / 64 - 1, iters, out);
    report("calib_gather64_hot", n_rec, n_rec * 64);
    CK(hipDeviceSynchronize());
    CK(hipFree(tab)); CK(hipFree(out));
    return 0;
}

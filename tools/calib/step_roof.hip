// step_roof.hip -- what can a traversal-SHAPED kernel reach on this chip?  A roof for the trace kernel's own unit of work.
//
// The trace kernel's step is: one 64-B record per lane (4 x 16-B buffer loads) at an address that depends on the previous record,
// ~110 VALU of slab arithmetic on it, next address.  Its achieved rate is ~111 G lane-steps/s at 38 of 64 lanes per iteration
// (DESIGN.md section 4).  The SURVEY's roofline (algorithmic bytes / 8 TB/s) cannot say how close that is to what the memory pipe and
// the SIMDs allow for THIS access pattern, so this program measures it: persistent waves at the product's occupancy (256 threads per
// block, 5 blocks per CU by a 23.8 KB LDS allocation and a 96-VGPR launch bound), every lane a DEPENDENT chain of steps over a table
// of the product's size, with the product's measured cache-hit mix (29 % of the record fetches hit L1, 48 % L2, 23 % go to the memory
// side: profiles/r3_pmc_summary.csv) emulated by drawing the next record from a 16 KB / 2 MB / whole-table region with those
// probabilities.  Variants:
//   gather        the four loads and the address chain only                                 -> memory-pipe bound
//   gather_slab   + the product's exact-quotient slab test of two boxes on the loaded data  -> + VALU
//   *_38          the same with 38 of 64 lanes active (the product's inner-step utilisation)
// Output: one JSON line per variant: G lane-steps/s.  tools/calib/run_step_roof.sh -> profiles/r3_step_roof.jsonl
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o step_roof step_roof.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t x) {             // lowbias32
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float fdiv_ray(float a, float d, float r) {   // pt_device_math.h: exact quotient by the per-ray reciprocal
    const float q0 = a * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, d, a), r, q0);
    return __builtin_fmaf(__builtin_fmaf(-q1, d, a), r, q1);
}
__device__ __forceinline__ float slab(float ox, float oy, float oz, float dx, float dy, float dz, float rx, float ry, float rz,
                                      u32x4 lo, u32x4 hi, float best) {
    const float t0 = fdiv_ray(__uint_as_float(lo.x) - ox, dx, rx), t1 = fdiv_ray(__uint_as_float(lo.y) - oy, dy, ry), t2 = fdiv_ray(__uint_as_float(lo.z) - oz, dz, rz);
    const float t3 = fdiv_ray(__uint_as_float(hi.x) - ox, dx, rx), t4 = fdiv_ray(__uint_as_float(hi.y) - oy, dy, ry), t5 = fdiv_ray(__uint_as_float(hi.z) - oz, dz, rz);
    const float n = fmaxf(fmaxf(fminf(t0, t3), fminf(t1, t4)), fminf(t2, t5)), f = fminf(fminf(fmaxf(t0, t3), fmaxf(t1, t4)), fmaxf(t2, t5));
    return (n <= f && f > 0.0f && n < best) ? n : 1e30f;
}

// hot16k / hot2m: number of 64-B records in the L1- and L2-sized regions (at the table's start); p1, p2: thresholds on a 16-bit
// random value for "L1 region", "L2 region" (else whole table)
template <bool SLAB, int NLOADS>
__global__ __launch_bounds__(256, 5) void step_chain(const void *tab, uint32_t tab_bytes, uint32_t n_rec, uint32_t steps, uint32_t active_lanes,
                                                     uint32_t hot1, uint32_t hot2, uint32_t p1, uint32_t p2, float *out) {
    extern __shared__ uint32_t lds[];                        // only there to set the occupancy
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)tab_bytes, 0x00020000);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    if (threadIdx.x == 0) lds[0] = gid;
    uint32_t rec = mix32(gid) % n_rec, h = gid * 2654435761u;
    float acc = 0.0f, best = 1e30f;
    const float ox = 0.1f, oy = 0.2f, oz = 0.3f, dx = 0.5f + 1e-3f * (float)(lane), dy = -0.7f, dz = 0.4f;
    const float rx = 1.0f / dx, ry = 1.0f / dy, rz = 1.0f / dz;
    if (lane < active_lanes) {
        for (uint32_t s = 0; s < steps; s++) {
            const uint32_t off = rec * 64u;
            const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
            const u32x4 r1 = NLOADS > 1 ? __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 16, 0, 0) : r0;
            const u32x4 r2 = NLOADS > 2 ? __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 32, 0, 0) : r0;
            const u32x4 r3 = NLOADS > 2 ? __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 48, 0, 0) : r1;
            uint32_t dep = r0.x ^ r1.y ^ r2.z ^ r3.w;          // the next address depends on the record (like `a` of the chosen child)
            if (SLAB) {
                const float d1 = slab(ox, oy, oz, dx, dy, dz, rx, ry, rz, r0, r1, best), d2 = slab(ox, oy, oz, dx, dy, dz, rx, ry, rz, r2, r3, best);
                acc += (d1 > d2) ? d2 : d1;
                dep ^= (d1 > d2) ? 1u : 0u;
            }
            h = mix32(h ^ dep ^ s);
            const uint32_t sel = h & 0xffffu, r = h >> 8;
            rec = sel < p1 ? r % hot1 : (sel < p2 ? r % hot2 : r % n_rec);
        }
    }
    out[gid] = acc + (float)rec;
}

// Quad-cooperative fetch: instruction k of four serves quad-lane k's record, the quad's lanes load its four quarters -- 16 distinct
// 64-B segments per load instruction instead of 64 -- and a two-stage butterfly (DPP row exchange with lane^1, lane^2) brings every lane
// its own record's quarters: 48 selects per step instead of 192 L1 lookups.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ void xchg(bool keep_first, uint32_t &x, uint32_t &y, bool lane1) {
    // 2x2 transpose between a lane pair: the lane with keep_first keeps x and receives the partner's x into y; the other keeps y and
    // receives the partner's y into x
    const uint32_t send = keep_first ? y : x;
    const uint32_t recv = lane1 ? dpp_quad<0xB1>(send) : dpp_quad<0x4E>(send);     // quad_perm [1,0,3,2] / [2,3,0,1]
    x = keep_first ? x : recv;
    y = keep_first ? recv : y;
}
template <bool SLAB>
__global__ __launch_bounds__(256, 5) void step_chain_coop(const void *tab, uint32_t tab_bytes, uint32_t n_rec, uint32_t steps,
                                                          uint32_t hot1, uint32_t hot2, uint32_t p1, uint32_t p2, float *out) {
    extern __shared__ uint32_t lds[];
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)tab_bytes, 0x00020000);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, j = lane & 3u;
    if (threadIdx.x == 0) lds[0] = gid;
    uint32_t rec = mix32(gid) % n_rec, h = gid * 2654435761u;
    float acc = 0.0f, best = 1e30f;
    const float ox = 0.1f, oy = 0.2f, oz = 0.3f, dx = 0.5f + 1e-3f * (float)(lane), dy = -0.7f, dz = 0.4f;
    const float rx = 1.0f / dx, ry = 1.0f / dy, rz = 1.0f / dz;
    const bool e1 = (j & 1u) == 0u, e2 = (j & 2u) == 0u;
    for (uint32_t s = 0; s < steps; s++) {
        const uint32_t off = rec * 64u;
        uint32_t W[4][4];                                    // W[k][c]: dword c of what load k returned
        {
            const u32x4 L0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(dpp_quad<0x00>(off) + 16u * j), 0, 0);
            const u32x4 L1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(dpp_quad<0x55>(off) + 16u * j), 0, 0);
            const u32x4 L2 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(dpp_quad<0xAA>(off) + 16u * j), 0, 0);
            const u32x4 L3 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(dpp_quad<0xFF>(off) + 16u * j), 0, 0);
            W[0][0] = L0.x; W[0][1] = L0.y; W[0][2] = L0.z; W[0][3] = L0.w; W[1][0] = L1.x; W[1][1] = L1.y; W[1][2] = L1.z; W[1][3] = L1.w;
            W[2][0] = L2.x; W[2][1] = L2.y; W[2][2] = L2.z; W[2][3] = L2.w; W[3][0] = L3.x; W[3][1] = L3.y; W[3][2] = L3.z; W[3][3] = L3.w;
        }
        // W[k] = quarter j of record k  ->  W[q] = quarter q of MY record
#pragma unroll
        for (int c = 0; c < 4; c++) { xchg(e1, W[0][c], W[1][c], true); xchg(e1, W[2][c], W[3][c], true); }
#pragma unroll
        for (int c = 0; c < 4; c++) { xchg(e2, W[0][c], W[2][c], false); xchg(e2, W[1][c], W[3][c], false); }
        u32x4 R[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { R[q].x = W[q][0]; R[q].y = W[q][1]; R[q].z = W[q][2]; R[q].w = W[q][3]; }
        uint32_t dep = R[0].x ^ R[1].y ^ R[2].z ^ R[3].w;
        if (SLAB) {
            const float d1 = slab(ox, oy, oz, dx, dy, dz, rx, ry, rz, R[0], R[1], best), d2 = slab(ox, oy, oz, dx, dy, dz, rx, ry, rz, R[2], R[3], best);
            acc += (d1 > d2) ? d2 : d1;
            dep ^= (d1 > d2) ? 1u : 0u;
        }
        h = mix32(h ^ dep ^ s);
        const uint32_t sel = h & 0xffffu, r = h >> 8;
        rec = sel < p1 ? r % hot1 : (sel < p2 ? r % hot2 : r % n_rec);
    }
    out[gid] = acc + (float)rec;
}

int main(int argc, char **argv) {
    const uint32_t tab_bytes = 1130u << 20;                   // the product's [pairs | tri_pos] allocation at 10 M triangles
    const uint32_t n_rec = tab_bytes / 64u, steps = argc > 1 ? (uint32_t)atoi(argv[1]) : 4000u;
    void *tab; float *out;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 5, threads = 256;
    CK(hipMalloc(&tab, tab_bytes)); CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CK(hipMemset(tab, 0x3c, tab_bytes));                       // finite floats; the `w` words are equal, so `dep` adds nothing but the dependency
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t hot1 = (16u << 10) / 64u, hot2 = (2u << 20) / 64u;
    const uint32_t p1 = (uint32_t)(0.29 * 65536), p2 = (uint32_t)((0.29 + 0.48) * 65536);
    struct { const char *name; bool slab; uint32_t lanes; uint32_t q1, q2; int nloads; } v[] = {
        {"gather (hit mix of the product, 64 lanes)", false, 64, p1, p2, 4}, {"gather_slab (hit mix, 64 lanes)", true, 64, p1, p2, 4},
        {"gather (hit mix, 38 lanes)", false, 38, p1, p2, 4}, {"gather_slab (hit mix, 38 lanes)", true, 38, p1, p2, 4},
        {"gather (all cold: every fetch a line fill)", false, 64, 0, 0, 4}, {"gather_slab (all cold)", true, 64, 0, 0, 4},
        {"gather (all L2-resident)", false, 64, 0, 65536, 4}, {"gather_slab (all L2-resident)", true, 64, 0, 65536, 4},
        // is the roof per 16-B load (L1 lookups) or per line?  The same chains with 2 and 1 loads per step (32 / 16 B of the record)
        {"gather, 2 x 16 B per step (hit mix)", false, 64, p1, p2, 2}, {"gather, 1 x 16 B per step (hit mix)", false, 64, p1, p2, 1},
        {"gather, 1 x 16 B per step (all L2-resident)", false, 64, 0, 65536, 1}, {"gather, 1 x 16 B per step (all cold)", false, 64, 0, 0, 1}};
    for (auto &c : v) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (c.slab) hipLaunchKernelGGL((step_chain<true, 4>), dim3(blocks), dim3(threads), 23808, 0, tab, tab_bytes, n_rec, steps, c.lanes, hot1, hot2, c.q1, c.q2, out);
            else if (c.nloads == 4) hipLaunchKernelGGL((step_chain<false, 4>), dim3(blocks), dim3(threads), 23808, 0, tab, tab_bytes, n_rec, steps, c.lanes, hot1, hot2, c.q1, c.q2, out);
            else if (c.nloads == 2) hipLaunchKernelGGL((step_chain<false, 2>), dim3(blocks), dim3(threads), 23808, 0, tab, tab_bytes, n_rec, steps, c.lanes, hot1, hot2, c.q1, c.q2, out);
            else hipLaunchKernelGGL((step_chain<false, 1>), dim3(blocks), dim3(threads), 23808, 0, tab, tab_bytes, n_rec, steps, c.lanes, hot1, hot2, c.q1, c.q2, out);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double lane_steps = (double)blocks * (threads / 64) * c.lanes * steps;
        printf("{\"variant\": \"%s\", \"waves\": %d, \"steps_per_lane\": %u, \"ms\": %.3f, \"G_lane_steps_s\": %.1f, \"us_per_wave_step\": %.3f}\n", c.name,
               blocks * threads / 64, steps, ms, lane_steps / (ms * 1e6), ms * 1e3 / steps);
        fflush(stdout);
    }
    for (int slab_on = 0; slab_on < 2; slab_on++)
        for (int mixv = 0; mixv < 2; mixv++) {
            const uint32_t q1 = mixv == 0 ? p1 : 0u, q2 = mixv == 0 ? p2 : 65536u;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (slab_on) hipLaunchKernelGGL(step_chain_coop<true>, dim3(blocks), dim3(threads), 23808, 0, tab, tab_bytes, n_rec, steps, hot1, hot2, q1, q2, out);
                else hipLaunchKernelGGL(step_chain_coop<false>, dim3(blocks), dim3(threads), 23808, 0, tab, tab_bytes, n_rec, steps, hot1, hot2, q1, q2, out);
                CK(hipGetLastError());
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double lane_steps = (double)blocks * (threads / 64) * 64 * steps;
            printf("{\"variant\": \"quad-cooperative fetch + register transpose%s (%s)\", \"waves\": %d, \"steps_per_lane\": %u, \"ms\": %.3f, \"G_lane_steps_s\": %.1f, \"us_per_wave_step\": %.3f}\n",
                   slab_on ? " + slab" : "", mixv == 0 ? "hit mix" : "all L2-resident", blocks * threads / 64, steps, ms, lane_steps / (ms * 1e6), ms * 1e3 / steps);
            fflush(stdout);
        }
    return 0;
}

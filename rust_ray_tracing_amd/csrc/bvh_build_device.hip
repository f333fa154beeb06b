// bvh_build_device.hip -- BVH::build (reference src/bvh.rs:13-161) on the GPU, emitting the IDENTICAL node array and
// triangle order as the host restatement (bvh_build.cpp) and the reference.  SURVEY 8(f) rank 4.
//
// Level-synchronous.  Every level's nodes are sorted into classes as they are created (device-side lists, no empty
// workgroups): > 2 048 triangles -> many workgroups per node, one per 8 192-element chunk (big_* kernels); 33..2 048 -> one WAVE
// per node with its proxies in registers (four instantiations: <= 64, <= 128, <= 512, <= 2 048); 9..32 -> several nodes per wave
// (16- and 32-lane groups); 5..8 -> one thread per node running the reference's loops as written; <= 4 -> one thread finishes the
// node's whole subtree.  A level's kernels run on four streams and end in level_mark (the barrier: no runtime join, see there).
// The steps of a split, in every class:
//   1. centroid range per axis (f32 min/max are exact and order-independent; for the multi-workgroup path it is a by-product of
//      the parent's scatter pass, the root's of make_proxies)
//   2. "first plane the centroid is below" binning with the reference's own plane values and `<` comparisons
//      (bvh.rs:82-84,147), per-bin boxes and counts through LDS integer atomics on order-preserving keys
//   3. thread 0 evaluates the 21 candidate costs with the reference's f32 expression (bvh.rs:150-160, incl. the
//      0*inf = NaN -> f32::MAX rule) and picks the first strictly lower one, axis-major / plane-minor (bvh.rs:86)
//   4. the partition loop (bvh.rs:99-108) is sequential in the reference; its result is a fixed permutation with a closed
//      form (derived and brute-forced in tests/test_bvh.py::test_partition_closed_form):
//         k = #(c < pos);  holes h_0<h_1<.. = positions < k holding a ">=" element;  t_0>t_1>.. = positions >= k holding a "<"
//         "<" at p < k stays;  h_m -> t_(m-1) - 1 (t_-1 = n);  t_m -> h_m;
//         ">=" at p >= k: p > t_last -> p-1;  p == k -> t_last - 1;  else p-1   (t_last = n when there is no hole)
//      computed with block-wide scans (ballots in the wave kernels), written out of place (ping-pong proxy buffers).
//   5. children appended in pairs; a final pass renumbers the breadth-first tree into the reference's depth-first order
//      desc(X) = [A, B] ++ desc(A) ++ desc(B) (bvh.rs:131-135) and gathers the 112-byte triangles.
#include "../../include/mipt.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mipt_internal.h"
#include "mipt_scene.h"

namespace {

constexpr int kT = 256;
constexpr float F32_MAX = FLT_MAX;
constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kSubFlag = 0x80000000u;   // BNode::left = kSubFlag | pool index: the node's whole subtree lives in the pool (build_subtree_tiny)
constexpr uint32_t kBig = 2048;            // nodes with more triangles are split by many workgroups (chunks of kChunk).  = kWaveMax: round 3 had a one-workgroup-per-node
                                           // class in between (2 049..8 192, eight passes behind barriers); once big_scatter lost its scratch traffic the chunked path beat it
constexpr uint32_t kChunk = 8192;
constexpr uint32_t kSub = 4;               // nodes this small: ONE thread finishes the whole subtree (build_subtree_tiny).  (kSub, kTiny) swept in round 4:
                                           // (8,16) 18.2 ms, (4,8) 16.9, (4,12) 16.9, (5,10) 17.0, (2,8) 18.1, (12,16) 21.7, (16,16) 29 -- profiles/r4_bvh_sub_tiny_sweep.txt
constexpr uint32_t kTiny = 8;               // nodes this small (and larger than kSub) are split by ONE thread running the reference's loops as written
constexpr uint32_t kCopies = 8;             // private copies of the LDS bin table in the workgroup kernels
constexpr uint32_t kWaveMax = 2048;          // kTiny+1..kWaveMax triangles: one wave64 per node (build_level_wave)
enum { CLS_WAVE = 0, CLS_TINY = 1, CLS_BIG = 2, CLS_SUB = 3, CLS_WAVE_M = 4, CLS_WAVE_L = 5, CLS_G16 = 6, CLS_G32 = 7, CLS_WAVE_A = 8, kClasses = 9 };   // CLS_G16: kTiny+1..16 and CLS_G32: 17..32 (several nodes per wave), CLS_WAVE: 33..64, _A: 65..128, _M: 129..512, _L: 513..kWaveMax triangles
constexpr uint32_t kWaveS = 64u, kWaveM = 512u;
// per-level work lists: lists[parity][class], level L reads parity L & 1 and appends the children it creates to parity (L + 1) & 1.  Their
// lengths live in THREE rotating counter rows, ctrl->cnt[L % 3][class]: when level L ends, its own row -- the host has its copy, the
// kernels got their counts as arguments -- is zeroed by level_mark, so no reset sits between two levels.  Kernels get "next" = list parity |
// counter row << 8 of the level they feed.
struct alignas(128) Pad32 { uint32_t v; uint32_t pad[31]; };        // one counter per 128-B line: atomics on one line serialise, whichever word they hit
struct Ctrl { Pad32 n_nodes, n_chunks, cnt[3][kClasses + 1], pool_alloc, sub_nodes, arrived; };   // cnt[parity][kClasses]: centroid-range records handed to that level's big nodes (big_finish); pool_*: nodes of the subtrees build_subtree_tiny finishes on its own; arrived: level_mark
struct Lists { uint32_t *l[2][kClasses]; };
// What the host needs of a level to size the next one, in pinned host memory (two slots, alternating): the counters, then the flag.
struct LevelSnap { uint32_t n_nodes, sub_nodes, cnt[kClasses + 1]; uint32_t pad[32 - 3 - kClasses]; volatile uint32_t flag; uint32_t pad2[31]; };

// The level's barrier.  A level's kernels run on up to four streams; joining them through the runtime (an event wait, the null stream,
// a blocking copy) costs 45..90 us per level on this stack -- tools/calib/level_sync.hip -- because every cross-queue dependency is a
// barrier packet on a completion signal.  Instead every stream that got work ends its level with this one-wave kernel: stream order
// puts it behind the stream's kernels, a device atomic finds the one that arrives last, and that one hands the level's counters to
// the host through pinned memory and raises the flag the host polls (~20 us per level, launch latency included).
__global__ __launch_bounds__(64) void level_mark(Ctrl *ctrl, uint32_t expect, uint32_t row_next, uint32_t row_done, LevelSnap *snap, uint32_t flag) {
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(&ctrl->arrived.v, 1u) == expect - 1u ? 1u : 0u; }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const uint32_t t = threadIdx.x;
    if (t <= kClasses) {
        snap->cnt[t] = __hip_atomic_load(&ctrl->cnt[row_next][t].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctrl->cnt[row_done][t].v = 0u;
    } else if (t == kClasses + 1u) snap->n_nodes = __hip_atomic_load(&ctrl->n_nodes.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (t == kClasses + 2u) snap->sub_nodes = __hip_atomic_load(&ctrl->sub_nodes.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (t == kClasses + 3u) ctrl->arrived.v = 0u;
    __threadfence_system();
    __syncthreads();
    if (t == 0) snap->flag = flag;
}

struct alignas(16) Proxy {                                            // 32 B = two 16-B words
    float lo[3]; uint32_t idx; float hi[3]; uint32_t pad;
    __device__ __forceinline__ float c(int a) const { return (lo[a] + hi[a]) / 2.0f; }   // the centroid (scene.rs:125) is re-derived, not stored: a fifth less traffic in every pass
    // the same for an axis only known at run time: selected from the three, because indexing lo[] / hi[] of a register-resident record
    // with a variable sends the record to scratch memory
    __device__ __forceinline__ float cax(int a) const {
        const float c0 = (lo[0] + hi[0]) / 2.0f, c1 = (lo[1] + hi[1]) / 2.0f, c2 = (lo[2] + hi[2]) / 2.0f;
        return a == 0 ? c0 : (a == 1 ? c1 : c2);
    }
};
struct BNode {
    float lo[3], hi[3];
    uint32_t first, n;
    uint32_t left;          // BFS index of the left child (right = left + 1), kNone = leaf
    uint32_t size;          // |desc(X)| in nodes
    uint32_t dfs;           // index in the reference's node array
    uint32_t base;          // where desc(X) starts in the reference's node array
};

// order-preserving float <-> uint key (so LDS integer min/max atomics give float min/max)
__device__ __forceinline__ uint32_t fkey(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float funkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// LDS min / max that first looks: after a bin's first few hundred triangles nearly none of them moves the bin's box, and a plain LDS
// read costs a fraction of an LDS atomic that 8+ lanes of the wave aim at one address.  A stale read only makes the atomic redundant.
__device__ __forceinline__ void lds_min(uint32_t *p, uint32_t v) { if (v < __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMin(p, v); }
__device__ __forceinline__ void lds_max(uint32_t *p, uint32_t v) { if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) atomicMax(p, v); }

__device__ __forceinline__ float box_area(const float *lo, const float *hi) {        // Node::surface_area, bvh.rs:196-203
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return (ex * ez) + (ex * ey) + (ez * ey);
}

// Queue the two children of every active lane for the next level.  One atomic per class per wave: the lanes are ranked
// with ballots (a per-lane atomicAdd on a per-lane class counter is not aggregated by the compiler and serialises in L2:
// measured 30x slower on the deep levels).  Works under divergence -- only the lanes that reach this point take part.
__host__ __device__ __forceinline__ uint32_t node_class(uint32_t n) {
    if (n > kBig) return (uint32_t)CLS_BIG;
    if (n > kWaveM) return (uint32_t)CLS_WAVE_L;
    if (n > 128u) return (uint32_t)CLS_WAVE_M;
    if (n > kWaveS) return (uint32_t)CLS_WAVE_A;
    if (n > 32u) return (uint32_t)CLS_WAVE;
    if (n > 16u) return (uint32_t)CLS_G32;
    if (n > kTiny) return (uint32_t)CLS_G16;
    return n > kSub ? (uint32_t)CLS_TINY : (uint32_t)CLS_SUB;
}
__device__ __forceinline__ uint32_t mask_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ void queue_children(Ctrl *ctrl, const Lists &ls, uint32_t next, uint32_t base_idx, uint32_t na, uint32_t nb) {
    const uint32_t parity = next & 1u, row = next >> 8;
    const uint32_t ca = node_class(na), cb = node_class(nb);
    const unsigned long long active = __ballot(true);
    const int leader = (int)__ffsll((long long)active) - 1;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long ma[kClasses], mb[kClasses];
    uint32_t slot0[kClasses];
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)kClasses; c++) { ma[c] = __ballot(ca == c); mb[c] = __ballot(cb == c); }
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)kClasses; c++) {               // all the wave's list atomics in flight together: one round trip, not one per class
        const uint32_t tot = (uint32_t)__popcll(ma[c]) + (uint32_t)__popcll(mb[c]);
        slot0[c] = 0;
        if (tot != 0u && (int)lane == leader) slot0[c] = atomicAdd(&ctrl->cnt[row][c].v, tot);
    }
#pragma unroll
    for (uint32_t c = 0; c < (uint32_t)kClasses; c++) {
        if ((ma[c] | mb[c]) == 0ull) continue;
        const uint32_t s0 = __shfl(slot0[c], leader);
        uint32_t *list = ls.l[parity][c];
        if (ca == c) list[s0 + mask_rank(ma[c])] = base_idx;
        if (cb == c) list[s0 + (uint32_t)__popcll(ma[c]) + mask_rank(mb[c])] = base_idx + 1u;
    }
}
struct Box3 { float lx, ly, lz, hx, hy, hz; };      // passed by value: keeps the callers' boxes in registers (pointer parameters made
                                                    // the one-thread-per-node kernel's locals spill into 40 KB of LDS per workgroup)
__device__ __forceinline__ uint32_t emit_children(BNode *bn, Ctrl *ctrl, const Lists &ls, uint32_t next_parity, uint32_t node_i,
                                                  Box3 A, Box3 B, uint32_t first, uint32_t k, uint32_t n) {
    const uint32_t base = atomicAdd(&ctrl->n_nodes.v, 2u);
    BNode a, b;
    a.lo[0] = A.lx; a.lo[1] = A.ly; a.lo[2] = A.lz; a.hi[0] = A.hx; a.hi[1] = A.hy; a.hi[2] = A.hz;
    b.lo[0] = B.lx; b.lo[1] = B.ly; b.lo[2] = B.lz; b.hi[0] = B.hx; b.hi[1] = B.hy; b.hi[2] = B.hz;
    a.first = first; a.n = k; a.left = kNone; a.size = 0; a.dfs = 0; a.base = 0;
    b.first = first + k; b.n = n - k; b.left = kNone; b.size = 0; b.dfs = 0; b.base = 0;
    bn[base] = a; bn[base + 1] = b;
    bn[node_i].left = base;
    queue_children(ctrl, ls, next_parity, base, k, n - k);
    return base;
}

// block-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix, *total = block sum
template <int NW = 4>
__device__ uint32_t block_exscan(uint32_t v, uint32_t *s_warp, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
    if (lane == 63u) s_warp[w] = x;
    __syncthreads();
    uint32_t add = 0, tot = 0;
    for (uint32_t i = 0; i < (uint32_t)NW; i++) { const uint32_t sw = s_warp[i]; if (i < w) add += sw; tot += sw; }
    __syncthreads();
    *total = tot;
    return add + x - v;
}

// rootkeys: [0..6) the root box (min keys, max keys), [6..12) the root's centroid range (bvh.rs:67-77 of the first split_node)
__global__ void make_proxies(const MiptTriangle *tris, uint32_t n, Proxy *px, uint32_t *rootkeys) {
    __shared__ uint32_t s_lo[3], s_hi[3], s_clo[3], s_chi[3];
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0xffffffffu; s_hi[threadIdx.x] = 0u; s_clo[threadIdx.x] = 0xffffffffu; s_chi[threadIdx.x] = 0u; }
    __syncthreads();
    float cmn[3] = {F32_MAX, F32_MAX, F32_MAX}, cmx[3] = {-F32_MAX, -F32_MAX, -F32_MAX};      // fminf / fmaxf skip a NaN centroid, as the reference's loop does
    uint32_t klo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, khi[3] = {0u, 0u, 0u};          // the root box as keys, per thread (six LDS atomics per triangle on six addresses were most of this kernel)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Proxy p;
        p.idx = i; p.pad = 0u;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float mn = F32_MAX, mx = -F32_MAX;                                          // scene.rs:115-123, bvh.rs:185-194
#pragma unroll
            for (int v = 0; v < 3; v++) { const float q = (&tris[i].vertices[v].position.x)[a]; mn = fminf(mn, q); mx = fmaxf(mx, q); }
            p.lo[a] = mn; p.hi[a] = mx;                                                 // scene.rs:125: see Proxy::c
            const uint32_t kl = fkey(mn), kh = fkey(mx);
            klo[a] = kl < klo[a] ? kl : klo[a]; khi[a] = kh > khi[a] ? kh : khi[a];
            const float c = (mn + mx) / 2.0f;
            cmn[a] = fminf(cmn[a], c); cmx[a] = fmaxf(cmx[a], c);
        }
        px[i] = p;
    }
    for (int a = 0; a < 3; a++) {
        atomicMin(&s_lo[a], klo[a]); atomicMax(&s_hi[a], khi[a]);
        atomicMin(&s_clo[a], fkey(cmn[a])); atomicMax(&s_chi[a], fkey(cmx[a]));
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&rootkeys[threadIdx.x], s_lo[threadIdx.x]); atomicMax(&rootkeys[3 + threadIdx.x], s_hi[threadIdx.x]);
        atomicMin(&rootkeys[6 + threadIdx.x], s_clo[threadIdx.x]); atomicMax(&rootkeys[9 + threadIdx.x], s_chi[threadIdx.x]);
    }
}

__global__ void init_root(BNode *bn, const uint32_t *rootkeys, uint32_t n) {
    BNode r;
    for (int a = 0; a < 3; a++) { r.lo[a] = funkey(rootkeys[a]); r.hi[a] = funkey(rootkeys[3 + a]); }
    r.first = 0; r.n = n; r.left = kNone; r.size = 0; r.dfs = 0; r.base = 1;
    bn[0] = r;
}

// ---- nodes with kTiny+1..kWaveMax triangles: one wave64 per node, four nodes per workgroup; the same five steps with wave-level
// reductions (f32 min/max and integer sums are exact in any order), a wave-private LDS region and no block barrier ----------
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// Wave-wide reductions of a value every lane holds (all 64 lanes active), result uniform.  Four DPP stages (lane ^ 1, lane ^ 2 inside a
// quad, then the 8-lane and the 16-lane mirror: after each stage the mirrored partner holds the other half's partial result) leave
// every row of 16 lanes with its row result; the four rows are combined through v_readlane.  (The __shfl_xor ladder these replace is
// six dependent ds_bpermute round trips through the LDS crossbar per reduction, and a node makes about twenty of them.)
template <class Op>
__device__ __forceinline__ uint32_t wave_reduce_bits(uint32_t v, Op op) {
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));   // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_fmin(float v) {
    return __uint_as_float(wave_reduce_bits(__float_as_uint(v), [](uint32_t a, uint32_t b) { return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ float wave_fmax(float v) {
    return __uint_as_float(wave_reduce_bits(__float_as_uint(v), [](uint32_t a, uint32_t b) { return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b))); }));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return wave_reduce_bits(v, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint32_t ballot_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// Each lane keeps its (up to kWaveMax / 64) proxies in registers: the node is read from memory once and written once.
// Slot j of lane l is position p = 64 j + l, so "increasing p" is slot-major / lane-minor order.
constexpr int kWaveBigNodes = kWaveMax > 1024u ? 2 : 4;   // nodes (= waves) per workgroup of the largest build_level_wave instantiation (LDS: 8 B per triangle)
struct WaveSplit { bool split; uint32_t k; Box3 A, B; };
// kSlots = proxies per lane: the kernel picks the smallest instantiation that holds the node (most wave-class nodes have <= 64
// triangles; walking 8 mostly empty slots through every stage was most of their cost)
template <int kSlots>
__device__ __forceinline__ WaveSplit wave_node(const BNode &nd, const Proxy *__restrict__ pin, Proxy *__restrict__ pout, uint32_t lane,
                                               uint32_t (*key)[8][6], uint32_t (*cnt)[8], uint32_t *hp, uint32_t *tp) {
    WaveSplit res;
    res.split = false; res.k = 0;
    res.A = Box3{0, 0, 0, 0, 0, 0}; res.B = res.A;
    const uint32_t first = nd.first, n = nd.n;
    const Proxy *in = pin + first;
    Proxy *out = pout + first;

    // 7 registers per proxy: the centroid is (lo + hi) / 2 (make_proxies, scene.rs:125) and is re-derived where it is needed
    float l0[kSlots], l1[kSlots], l2[kSlots], h0[kSlots], h1[kSlots], h2[kSlots];
    uint32_t id[kSlots];
#define C0(j) ((l0[j] + h0[j]) / 2.0f)
#define C1(j) ((l1[j] + h1[j]) / 2.0f)
#define C2(j) ((l2[j] + h2[j]) / 2.0f)
#pragma unroll
    for (int j = 0; j < kSlots; j++) {
        const uint32_t p = 64u * (uint32_t)j + lane;
        if (p < n) {
            const Proxy e = in[p];
            l0[j] = e.lo[0]; l1[j] = e.lo[1]; l2[j] = e.lo[2];
            h0[j] = e.hi[0]; h1[j] = e.hi[1]; h2[j] = e.hi[2]; id[j] = e.idx;
        } else {                                                      // neutral for every min / max below
            l0[j] = l1[j] = l2[j] = F32_MAX; h0[j] = h1[j] = h2[j] = -F32_MAX; id[j] = 0u;
        }
    }
    // ---- 1. centroid ranges + planes (bvh.rs:67-84) ----
    float cmin[3], cmax[3];
    {
        float mn[3] = {F32_MAX, F32_MAX, F32_MAX}, mx[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
#pragma unroll
        for (int j = 0; j < kSlots; j++)
            if (64u * (uint32_t)j + lane < n) {
                const float a0 = C0(j), a1 = C1(j), a2 = C2(j);
                mn[0] = fminf(mn[0], a0); mx[0] = fmaxf(mx[0], a0); mn[1] = fminf(mn[1], a1); mx[1] = fmaxf(mx[1], a1);
                mn[2] = fminf(mn[2], a2); mx[2] = fmaxf(mx[2], a2);
            }
        for (int a = 0; a < 3; a++) { cmin[a] = wave_fmin(mn[a]); cmax[a] = wave_fmax(mx[a]); }
    }
    bool use[3];
    float scale[3];
    for (int a = 0; a < 3; a++) { use[a] = !(cmin[a] == cmax[a]); scale[a] = (cmax[a] - cmin[a]) / 8.0f; }
    for (uint32_t i = lane; i < 144u; i += 64u) (&key[0][0][0])[i] = ((i % 6u) < 3u) ? 0xffffffffu : 0u;
    if (lane < 24u) (&cnt[0][0])[lane] = 0u;
    wave_sync();
    // ---- 2. binning ----
#pragma unroll
    for (int j = 0; j < kSlots; j++) {
        if (64u * (uint32_t)j >= n) break;                           // wave-uniform
        if (64u * (uint32_t)j + lane < n) {
            const float cc[3] = {C0(j), C1(j), C2(j)};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (!use[a]) continue;
                int k = 8;
                for (int q = 7; q >= 1; q--) if (cc[a] < cmin[a] + (float)q * scale[a]) k = q;      // first plane the centroid is below
                uint32_t *kk = key[a][k - 1];
                lds_min(&kk[0], fkey(l0[j])); lds_min(&kk[1], fkey(l1[j])); lds_min(&kk[2], fkey(l2[j]));
                lds_max(&kk[3], fkey(h0[j])); lds_max(&kk[4], fkey(h1[j])); lds_max(&kk[5], fkey(h2[j]));
                atomicAdd(&cnt[a][k - 1], 1u);
            }
        }
    }
    wave_sync();
    // ---- 3. SAH: lane c < 21 evaluates candidate (axis c / 7, plane c % 7 + 1); the first strictly lower cost in
    //         axis-major / plane-minor order wins (bvh.rs:86) = the lowest lane holding the minimum ----
    float my_cost = F32_MAX, my_pos = 0.0f;
    uint32_t my_k = 0;
    if (lane < 21u) {
        const int a = (int)(lane / 7u), i = (int)(lane % 7u) + 1;
        if (use[a]) {
            float llo[3] = {F32_MAX, F32_MAX, F32_MAX}, lhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
            float rlo[3] = {F32_MAX, F32_MAX, F32_MAX}, rhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
            uint32_t lc = 0, rc = 0;
            for (int b = 0; b < 8; b++) {                            // an empty bin's untouched keys decode to NaN, which fminf/fmaxf ignore
                if (b < i) { for (int q = 0; q < 3; q++) { llo[q] = fminf(llo[q], funkey(key[a][b][q])); lhi[q] = fmaxf(lhi[q], funkey(key[a][b][3 + q])); } lc += cnt[a][b]; }
                else { for (int q = 0; q < 3; q++) { rlo[q] = fminf(rlo[q], funkey(key[a][b][q])); rhi[q] = fmaxf(rhi[q], funkey(key[a][b][3 + q])); } rc += cnt[a][b]; }
            }
            const float cost = (float)lc * box_area(llo, lhi) + (float)rc * box_area(rlo, rhi);
            my_cost = (cost > 0.0f) ? cost : F32_MAX;
            my_pos = cmin[a] + (float)i * scale[a];
            my_k = lc;                                               // #(c < pos): bin b < i  <=>  c below plane i (planes increase with i)
        }
    }
    const float best_cost = wave_fmin(my_cost);
    int axis = 0;
    float pos = 0.0f;
    uint32_t k = 0;
    bool k_known = false;
    if (best_cost < F32_MAX) {                                       // else: nothing beat the initial f32::MAX -> axis 0, pos 0.0 (bvh.rs:60-62)
        const unsigned long long m = __ballot(my_cost == best_cost);
        const int win = (int)__ffsll((long long)m) - 1;
        axis = win / 7;
        pos = __shfl(my_pos, win);
        k = __shfl(my_k, win);
        k_known = true;
    }
    const float parent_cost = (float)n * box_area(nd.lo, nd.hi);
    if (best_cost >= parent_cost) {                                  // leaf (bvh.rs:94): carry the range over unchanged
#pragma unroll
        for (int j = 0; j < kSlots; j++) {
            const uint32_t p = 64u * (uint32_t)j + lane;
            if (p < n) { Proxy e; e.pad = 0u; e.lo[0] = l0[j]; e.lo[1] = l1[j]; e.lo[2] = l2[j]; e.hi[0] = h0[j]; e.hi[1] = h1[j]; e.hi[2] = h2[j]; e.idx = id[j]; out[p] = e; }
        }
        return res;
    }
    // ---- 4. the partition permutation (closed form of bvh.rs:99-108, see the header comment) ----
    // Nothing per slot is kept besides the proxy itself: "c < pos" is re-evaluated and the ranks are re-derived from ballots in the
    // second pass (two compares and a ballot cost less than the registers, which decide how many triangles a wave can hold).
#define LESS(j) ((64u * (uint32_t)(j) + lane < n) && ((axis == 0 ? C0(j) : (axis == 1 ? C1(j) : C2(j))) < pos))
    if (!k_known) {                                                  // NaN parent cost with no finite candidate: count directly
        uint32_t cl = 0;
#pragma unroll
        for (int j = 0; j < kSlots; j++) cl += LESS(j) ? 1u : 0u;
        k = wave_sum(cl);
    }
    uint32_t n_holes = 0;
#pragma unroll
    for (int j = 0; j < kSlots; j++) {                               // holes: positions < k holding ">=", increasing p
        const uint32_t p = 64u * (uint32_t)j + lane;
        const bool f = p < k && !LESS(j);
        const unsigned long long m = __ballot(f);
        if (f) hp[n_holes + ballot_rank(m)] = p;
        n_holes += (uint32_t)__popcll(m);
    }
    uint32_t n_tail = 0;
#pragma unroll
    for (int j = kSlots - 1; j >= 0; j--) {                          // tail "<": positions >= k holding "<", decreasing p
        const uint32_t p = 64u * (uint32_t)j + lane;
        const bool f = p >= k && LESS(j);                            // (false beyond n)
        const unsigned long long m = __ballot(f);
        const unsigned long long above = lane == 63u ? 0ull : (m >> (lane + 1u));
        if (f) tp[n_tail + (uint32_t)__popcll(above)] = p;
        n_tail += (uint32_t)__popcll(m);
    }
    wave_sync();
    const uint32_t t_last = n_holes ? tp[n_holes - 1u] : n;
    float clo[2][3] = {{F32_MAX, F32_MAX, F32_MAX}, {F32_MAX, F32_MAX, F32_MAX}}, chi[2][3] = {{-F32_MAX, -F32_MAX, -F32_MAX}, {-F32_MAX, -F32_MAX, -F32_MAX}};
    uint32_t holes_before = 0, tail_through = 0;                     // holes in the slots below j; tail "<" in the slots up to and including j
#pragma unroll
    for (int j = 0; j < kSlots; j++) {
        const uint32_t p = 64u * (uint32_t)j + lane;
        const bool lt = LESS(j);
        const unsigned long long mh = __ballot(p < k && !lt), mt = __ballot(p >= k && lt);
        tail_through += (uint32_t)__popcll(mt);
        if (p < n) {
            uint32_t dest;
            if (p < k) {
                dest = p;
                if (!lt) { const uint32_t mm = holes_before + ballot_rank(mh); dest = (mm ? tp[mm - 1u] : n) - 1u; }
            } else {
                if (lt) {
                    const unsigned long long above = lane == 63u ? 0ull : (mt >> (lane + 1u));
                    dest = hp[(n_tail - tail_through) + (uint32_t)__popcll(above)];
                }
                else if (p > t_last) dest = p - 1u;
                else dest = (p == k) ? (t_last - 1u) : (p - 1u);
            }
            Proxy e; e.pad = 0u; e.lo[0] = l0[j]; e.lo[1] = l1[j]; e.lo[2] = l2[j]; e.hi[0] = h0[j]; e.hi[1] = h1[j]; e.hi[2] = h2[j]; e.idx = id[j];
            out[dest] = e;
            const int side = lt ? 0 : 1;
            if (side) {                                   // (constant indices only: a `[side]` index would send the arrays to scratch)
                clo[1][0] = fminf(clo[1][0], l0[j]); clo[1][1] = fminf(clo[1][1], l1[j]); clo[1][2] = fminf(clo[1][2], l2[j]);
                chi[1][0] = fmaxf(chi[1][0], h0[j]); chi[1][1] = fmaxf(chi[1][1], h1[j]); chi[1][2] = fmaxf(chi[1][2], h2[j]);
            } else {
                clo[0][0] = fminf(clo[0][0], l0[j]); clo[0][1] = fminf(clo[0][1], l1[j]); clo[0][2] = fminf(clo[0][2], l2[j]);
                chi[0][0] = fmaxf(chi[0][0], h0[j]); chi[0][1] = fmaxf(chi[0][1], h1[j]); chi[0][2] = fmaxf(chi[0][2], h2[j]);
            }
        }
        holes_before += (uint32_t)__popcll(mh);
    }
#undef LESS
    for (int sd = 0; sd < 2; sd++)
        for (int q = 0; q < 3; q++) { clo[sd][q] = wave_fmin(clo[sd][q]); chi[sd][q] = wave_fmax(chi[sd][q]); }
    if (k == 0u || k == n) return res;                               // bvh.rs:110-113 (cannot happen with a finite best cost)
    res.split = true; res.k = k;
    res.A = Box3{clo[0][0], clo[0][1], clo[0][2], chi[0][0], chi[0][1], chi[0][2]};
    res.B = Box3{clo[1][0], clo[1][1], clo[1][2], chi[1][0], chi[1][1], chi[1][2]};
    return res;
}
#undef C0
#undef C1
#undef C2
// ---- 5. children: the workgroup's waves hand their splits to wave 0, which allocates the child nodes and queues them with
// ONE set of atomics per workgroup (a per-node atomic on the shared counters serialises in L2: ~6 ns each, which was the whole
// cost of the deep levels) ----
// Four instantiations, each with its own class list, take the nodes with LO < n <= HI triangles: 33..64 (one proxy per lane: a third of
// the registers, so 8 waves per SIMD -- a node is a chain of dependent memory round trips, and most wave-class nodes are this small),
// 65..128 (2 per lane, 6 waves), 129..512 (8 per lane, 3 waves: at 4 the kernel spilled 68 B per lane for the same time) and
// 513..kWaveMax (kWaveMax / 64 proxies per lane, 2 waves per SIMD, fewer nodes per workgroup for the LDS position lists).  Reading a
// node once into registers and writing it once is what made this kernel ~10x faster per triangle than the one-workgroup-per-node
// kernel of round 3 with its eight passes over global memory -- hence the wide range.
template <uint32_t LO, uint32_t HI, int NODES, int MINW>
__global__ __launch_bounds__(64 * NODES, MINW) void build_level_wave(BNode *bn, const uint32_t *__restrict__ list, uint32_t count,
                                                                      const Proxy *__restrict__ pin, Proxy *__restrict__ pout, Ctrl *ctrl,
                                                                      Lists ls, uint32_t next_parity) {
    constexpr int kS = (int)(HI / 64u);
    __shared__ uint32_t s_keyw[NODES][3][8][6];
    __shared__ uint32_t s_cntw[NODES][3][8];
    __shared__ uint32_t s_hp[NODES][HI], s_tp[NODES][HI];
    __shared__ uint32_t s_node[NODES], s_first[NODES], s_n[NODES], s_k[NODES];
    __shared__ Box3 s_box[NODES][2];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t item = blockIdx.x * (uint32_t)NODES + wv;
    WaveSplit r;
    r.split = false; r.k = 0;
    uint32_t node_i = 0, first = 0, n = 0;
    if (item < count) {                                              // wave-uniform
        node_i = list[item];
        n = bn[node_i].n;
        if (n > LO && n <= HI) {                                     // wave-uniform
            const BNode nd = bn[node_i];
            first = nd.first;
            if (kS >= 8 && n <= 4u * 64u) {                          // the smallest instantiation that holds the node
                if (LO < 128u && n <= 128u) r = wave_node<2>(nd, pin, pout, lane, s_keyw[wv], s_cntw[wv], s_hp[wv], s_tp[wv]);
                else r = wave_node<4>(nd, pin, pout, lane, s_keyw[wv], s_cntw[wv], s_hp[wv], s_tp[wv]);
            } else if (kS >= 32 && n <= 16u * 64u) {
                r = wave_node<16>(nd, pin, pout, lane, s_keyw[wv], s_cntw[wv], s_hp[wv], s_tp[wv]);
            } else {
                r = wave_node<kS>(nd, pin, pout, lane, s_keyw[wv], s_cntw[wv], s_hp[wv], s_tp[wv]);
            }
        }
    }
    if (lane == 0u) {
        s_node[wv] = node_i; s_first[wv] = first; s_n[wv] = n; s_k[wv] = r.split ? r.k : 0u;     // k == 0 marks "no split" (or: not this kernel's node)
        s_box[wv][0] = r.A; s_box[wv][1] = r.B;
    }
    __syncthreads();
    if (wv == 0u) {                                                  // lanes 0..NODES-1 each emit one node's children
        const bool mine = lane < (uint32_t)NODES && s_k[lane] != 0u;
        if (mine) emit_children(bn, ctrl, ls, next_parity, s_node[lane], s_box[lane][0], s_box[lane][1], s_first[lane], s_k[lane], s_n[lane]);
    }
}

// ---- small wave-class nodes, SEVERAL PER WAVE: a node with <= G triangles (G = 16, 32) occupies an aligned group of G lanes, one
// proxy per lane -- the same five steps as wave_node<1> with the reductions, ballots and rank computations confined to the group.
// Most nodes of the deep levels hold 9..32 triangles; a whole wave each left three quarters of the lanes idle through a chain of
// ~20 dependent reductions.  Everything here is executed by all 64 lanes (the DPP reductions need them); a group without a node, or
// whose node stays a leaf, carries neutral values through the later steps.
template <int G, class Op> __device__ __forceinline__ uint32_t group_reduce_bits(uint32_t v, Op op) {
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));               // quad_perm [1,0,3,2]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));               // quad_perm [2,3,0,1]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));              // row_half_mirror: 8 lanes
    if (G >= 16) v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false)); // row_mirror: a row of 16
    if (G >= 32) v = op(v, (uint32_t)__shfl_xor((int)v, 16));                                          // the neighbouring row
    return v;
}
template <int G> __device__ __forceinline__ float group_fmin(float v) {
    return __uint_as_float(group_reduce_bits<G>(__float_as_uint(v), [](uint32_t a, uint32_t b) { return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b))); }));
}
template <int G> __device__ __forceinline__ float group_fmax(float v) {
    return __uint_as_float(group_reduce_bits<G>(__float_as_uint(v), [](uint32_t a, uint32_t b) { return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b))); }));
}
template <int G> __device__ __forceinline__ uint32_t group_umin(uint32_t v) { return group_reduce_bits<G>(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
template <int G> __device__ __forceinline__ uint32_t group_sum(uint32_t v) { return group_reduce_bits<G>(v, [](uint32_t a, uint32_t b) { return a + b; }); }

template <int G, uint32_t LO, uint32_t HI>
__global__ __launch_bounds__(512, (G == 16 ? 6 : 8)) void build_level_group(BNode *bn, const uint32_t *__restrict__ list, uint32_t count, const Proxy *__restrict__ pin,
                                                            Proxy *__restrict__ pout, Ctrl *ctrl, Lists ls, uint32_t next_parity) {
    static_assert(G == 16 || G == 32, "group of 16 or 32 lanes");
    constexpr int kPerWave = 64 / G, kWaves = 8, kNodes = kPerWave * kWaves;     // nodes per workgroup: 32 (G = 16) or 16 (G = 32)
    __shared__ uint32_t s_key[kNodes][3][8][6];
    __shared__ uint32_t s_cnt[kNodes][3][8];
    __shared__ uint32_t s_hp[kNodes][G], s_tp[kNodes][G];
    __shared__ uint32_t s_node[kNodes], s_first[kNodes], s_n[kNodes], s_k[kNodes];
    __shared__ Box3 s_box[kNodes][2];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t lig = lane & (uint32_t)(G - 1), grp = lane / (uint32_t)G;            // lane in group, group in wave
    const uint32_t slot = wv * (uint32_t)kPerWave + grp;                                 // the group's node slot in the workgroup
    const uint32_t item = blockIdx.x * (uint32_t)kNodes + slot;
    const uint32_t gshift = grp * (uint32_t)G;
    const unsigned long long gmask = (G == 32 ? 0xffffffffull : 0xffffull);
    uint32_t (*key)[8][6] = s_key[slot];
    uint32_t (*cnt)[8] = s_cnt[slot];
    uint32_t *hp = s_hp[slot], *tp = s_tp[slot];

    uint32_t node_i = 0, first = 0, n = 0;
    float blo0 = 0, blo1 = 0, blo2 = 0, bhi0 = 0, bhi1 = 0, bhi2 = 0;                   // the node's own box
    bool has = false;
    if (item < count) {
        node_i = list[item];
        const BNode nd = bn[node_i];
        if (nd.n > LO && nd.n <= HI) {
            has = true; first = nd.first; n = nd.n;
            blo0 = nd.lo[0]; blo1 = nd.lo[1]; blo2 = nd.lo[2]; bhi0 = nd.hi[0]; bhi1 = nd.hi[1]; bhi2 = nd.hi[2];
        }
    }
    const Proxy *in = pin + first;
    Proxy *out = pout + first;
    const bool valid = has && lig < n;
    float l0 = F32_MAX, l1 = F32_MAX, l2 = F32_MAX, h0 = -F32_MAX, h1 = -F32_MAX, h2 = -F32_MAX;
    uint32_t id = 0;
    if (valid) { const Proxy e = in[lig]; l0 = e.lo[0]; l1 = e.lo[1]; l2 = e.lo[2]; h0 = e.hi[0]; h1 = e.hi[1]; h2 = e.hi[2]; id = e.idx; }
    const float c0 = (l0 + h0) / 2.0f, c1 = (l1 + h1) / 2.0f, c2 = (l2 + h2) / 2.0f;    // scene.rs:125 (lanes without a proxy: never used)
    // ---- 1. centroid ranges + planes (bvh.rs:67-84) ----
    float cmin[3], cmax[3];
    cmin[0] = group_fmin<G>(valid ? c0 : F32_MAX); cmax[0] = group_fmax<G>(valid ? c0 : -F32_MAX);
    cmin[1] = group_fmin<G>(valid ? c1 : F32_MAX); cmax[1] = group_fmax<G>(valid ? c1 : -F32_MAX);
    cmin[2] = group_fmin<G>(valid ? c2 : F32_MAX); cmax[2] = group_fmax<G>(valid ? c2 : -F32_MAX);
    bool use[3];
    float scale[3];
    for (int a = 0; a < 3; a++) { use[a] = !(cmin[a] == cmax[a]); scale[a] = (cmax[a] - cmin[a]) / 8.0f; }
    for (uint32_t i = lig; i < 144u; i += (uint32_t)G) (&key[0][0][0])[i] = ((i % 6u) < 3u) ? 0xffffffffu : 0u;
    for (uint32_t i = lig; i < 24u; i += (uint32_t)G) (&cnt[0][0])[i] = 0u;
    wave_sync();
    // ---- 2. binning ----
    if (valid) {
        const float cc[3] = {c0, c1, c2};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (!use[a]) continue;
            int k = 8;
            for (int q = 7; q >= 1; q--) if (cc[a] < cmin[a] + (float)q * scale[a]) k = q;          // first plane the centroid is below
            uint32_t *kk = key[a][k - 1];
            lds_min(&kk[0], fkey(l0)); lds_min(&kk[1], fkey(l1)); lds_min(&kk[2], fkey(l2));
            lds_max(&kk[3], fkey(h0)); lds_max(&kk[4], fkey(h1)); lds_max(&kk[5], fkey(h2));
            atomicAdd(&cnt[a][k - 1], 1u);
        }
    }
    wave_sync();
    // ---- 3. SAH: candidate c = 7 * axis + plane - 1 (axis-major / plane-minor); lane `lig` evaluates c = lig, lig + G, ...; the first
    //         strictly lower cost wins (bvh.rs:86) = the lowest candidate index holding the minimum ----
    float my_cost = F32_MAX, my_pos = 0.0f;
    uint32_t my_k = 0, my_c = 0xffu;
    for (uint32_t c = lig; c < 21u; c += (uint32_t)G) {
        const int a = (int)(c / 7u), i = (int)(c % 7u) + 1;
        if (!has || !use[a]) continue;
        float llo[3] = {F32_MAX, F32_MAX, F32_MAX}, lhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
        float rlo[3] = {F32_MAX, F32_MAX, F32_MAX}, rhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
        uint32_t lc = 0, rc = 0;
        for (int b = 0; b < 8; b++) {                            // an empty bin's untouched keys decode to NaN, which fminf/fmaxf ignore
            if (b < i) { for (int q = 0; q < 3; q++) { llo[q] = fminf(llo[q], funkey(key[a][b][q])); lhi[q] = fmaxf(lhi[q], funkey(key[a][b][3 + q])); } lc += cnt[a][b]; }
            else { for (int q = 0; q < 3; q++) { rlo[q] = fminf(rlo[q], funkey(key[a][b][q])); rhi[q] = fmaxf(rhi[q], funkey(key[a][b][3 + q])); } rc += cnt[a][b]; }
        }
        const float cost = (float)lc * box_area(llo, lhi) + (float)rc * box_area(rlo, rhi);
        const float sc = (cost > 0.0f) ? cost : F32_MAX;
        if (sc < my_cost) { my_cost = sc; my_pos = cmin[a] + (float)i * scale[a]; my_k = lc; my_c = c; }   // (a lane's candidates come in increasing order)
    }
    const float best_cost = group_fmin<G>(my_cost);
    int axis = 0;
    float pos = 0.0f;
    uint32_t k = 0;
    bool k_known = false;
    {
        const uint32_t win_c = group_umin<G>((my_cost == best_cost && my_c != 0xffu) ? my_c : 0xffu);
        if (best_cost < F32_MAX && win_c != 0xffu) {              // else: nothing beat the initial f32::MAX -> axis 0, pos 0.0 (bvh.rs:60-62)
            const int src = (int)(gshift + (win_c % (uint32_t)G));
            axis = (int)(win_c / 7u);
            pos = __shfl(my_pos, src);
            k = __shfl(my_k, src);
            k_known = true;
        }                                                         // (group-uniform branch: the source lanes of the shuffles sit in the same group)
    }
    const float nb_lo[3] = {blo0, blo1, blo2}, nb_hi[3] = {bhi0, bhi1, bhi2};
    const float parent_cost = (float)n * box_area(nb_lo, nb_hi);
    const bool split_try = has && !(best_cost >= parent_cost);   // bvh.rs:94
    // ---- 4. the partition permutation (closed form of bvh.rs:99-108) ----
    const float cax = axis == 0 ? c0 : (axis == 1 ? c1 : c2);
    const bool less = valid && (cax < pos);
    {
        const uint32_t kc = group_sum<G>(less ? 1u : 0u);          // NaN parent cost with no finite candidate: count directly
        if (!k_known) k = kc;
    }
    const bool is_hole = split_try && valid && lig < k && !less;
    const bool is_tail = split_try && valid && lig >= k && less;
    const uint32_t mh = (uint32_t)((__ballot(is_hole) >> gshift) & gmask), mt = (uint32_t)((__ballot(is_tail) >> gshift) & gmask);
    const uint32_t below = (lig == 0u) ? 0u : (0xffffffffu >> (32u - lig));            // bits of the lanes below this one in its group
    const uint32_t hole_rank = (uint32_t)__popc(mh & below);                             // holes in increasing position
    const uint32_t tail_rank = (uint32_t)__popc(mt & ~below & ~(1u << lig));             // tails in decreasing position
    const uint32_t n_holes = (uint32_t)__popc(mh);
    if (is_hole) hp[hole_rank] = lig;
    if (is_tail) tp[tail_rank] = lig;
    wave_sync();
    const uint32_t t_last = n_holes ? tp[n_holes - 1u] : n;
    if (valid) {
        uint32_t dest = lig;                                       // a leaf carries its range over unchanged
        if (split_try) {
            if (lig < k) { if (!less) dest = (hole_rank ? tp[hole_rank - 1u] : n) - 1u; }
            else if (less) dest = hp[tail_rank];
            else if (lig > t_last) dest = lig - 1u;
            else dest = (lig == k) ? (t_last - 1u) : (lig - 1u);
        }
        Proxy e; e.pad = 0u; e.lo[0] = l0; e.lo[1] = l1; e.lo[2] = l2; e.hi[0] = h0; e.hi[1] = h1; e.hi[2] = h2; e.idx = id;
        out[dest] = e;
    }
    // ---- child boxes (bvh.rs:115-130) ----
    const bool inA = valid && less, inB = valid && !less;
    Box3 A, B;
    A.lx = group_fmin<G>(inA ? l0 : F32_MAX); A.ly = group_fmin<G>(inA ? l1 : F32_MAX); A.lz = group_fmin<G>(inA ? l2 : F32_MAX);
    A.hx = group_fmax<G>(inA ? h0 : -F32_MAX); A.hy = group_fmax<G>(inA ? h1 : -F32_MAX); A.hz = group_fmax<G>(inA ? h2 : -F32_MAX);
    B.lx = group_fmin<G>(inB ? l0 : F32_MAX); B.ly = group_fmin<G>(inB ? l1 : F32_MAX); B.lz = group_fmin<G>(inB ? l2 : F32_MAX);
    B.hx = group_fmax<G>(inB ? h0 : -F32_MAX); B.hy = group_fmax<G>(inB ? h1 : -F32_MAX); B.hz = group_fmax<G>(inB ? h2 : -F32_MAX);
    const bool split = split_try && k != 0u && k != n;            // bvh.rs:110-113 (cannot fail with a finite best cost)
    if (lig == 0u) {
        s_node[slot] = node_i; s_first[slot] = first; s_n[slot] = n; s_k[slot] = split ? k : 0u;     // k == 0 marks "no split" / no node
        s_box[slot][0] = A; s_box[slot][1] = B;
    }
    __syncthreads();
    if (wv == 0u) {                                                // lanes 0..kNodes-1 each emit one node's children
        const bool mine = lane < (uint32_t)kNodes && s_k[lane] != 0u;
        if (mine) emit_children(bn, ctrl, ls, next_parity, s_node[lane], s_box[lane][0], s_box[lane][1], s_first[lane], s_k[lane], s_n[lane]);
    }
}

// ---- nodes with <= kTiny triangles: one thread per node, the reference's own loops (bvh.rs:56-161) on the proxies ----------
// The node's proxies are staged in LDS ([slot][word][thread]: conflict-free) -- the 21 candidate sweeps and the in-place partition
// were a chain of ~300 dependent global-memory reads per thread before, which is what a level cost whenever it had such nodes.
__global__ __launch_bounds__(64) void build_level_tiny(BNode *bn, const uint32_t *__restrict__ list, uint32_t count, const Proxy *__restrict__ pin, Proxy *__restrict__ pout,
                                 Ctrl *ctrl, Lists ls, uint32_t next_parity) {
    __shared__ float s_p[kTiny][7][64];                   // lo.xyz, hi.xyz, idx (as bits)
    const uint32_t t = threadIdx.x, item = blockIdx.x * 64u + t;
    if (item >= count) return;
    const uint32_t node_i = list[item];
    const BNode nd = bn[node_i];
    const uint32_t n = nd.n;
    const Proxy *in = pin + nd.first;
    Proxy *out = pout + nd.first;
    for (uint32_t i = 0; i < n; i++) {
        const Proxy p = in[i];
        for (int q = 0; q < 3; q++) { s_p[i][q][t] = p.lo[q]; s_p[i][3 + q][t] = p.hi[q]; }
        s_p[i][6][t] = __uint_as_float(p.idx);
    }
    auto cen = [&](uint32_t i, int a) { return (s_p[i][a][t] + s_p[i][3 + a][t]) / 2.0f; };
    auto store = [&]() {                                  // the range in its current order
        for (uint32_t i = 0; i < n; i++) {
            Proxy p;
            for (int q = 0; q < 3; q++) { p.lo[q] = s_p[i][q][t]; p.hi[q] = s_p[i][3 + q][t]; }
            p.idx = __float_as_uint(s_p[i][6][t]); p.pad = 0u;
            out[i] = p;
        }
    };
    const float parent_cost = (float)n * box_area(nd.lo, nd.hi);
    int best_axis = 0;
    float best_pos = 0.0f, best_cost = F32_MAX;
    for (int a = 0; a < 3; a++) {
        float cmin = F32_MAX, cmax = -F32_MAX;
        for (uint32_t i = 0; i < n; i++) { const float c = cen(i, a); cmin = fminf(cmin, c); cmax = fmaxf(cmax, c); }
        if (cmin == cmax) continue;
        const float scale = (cmax - cmin) / 8.0f;
        for (int i = 1; i < 8; i++) {
            const float pos = cmin + (float)i * scale;
            float llo[3] = {F32_MAX, F32_MAX, F32_MAX}, lhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
            float rlo[3] = {F32_MAX, F32_MAX, F32_MAX}, rhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
            uint32_t lc = 0, rc = 0;
            for (uint32_t j = 0; j < n; j++) {                                            // evaluate_sah, bvh.rs:138-161
                const bool left = cen(j, a) < pos;
                for (int q = 0; q < 3; q++) {
                    const float pl = s_p[j][q][t], ph = s_p[j][3 + q][t];
                    if (left) { llo[q] = fminf(llo[q], pl); lhi[q] = fmaxf(lhi[q], ph); }
                    else { rlo[q] = fminf(rlo[q], pl); rhi[q] = fmaxf(rhi[q], ph); }
                }
                if (left) lc++; else rc++;
            }
            const float cost = (float)lc * box_area(llo, lhi) + (float)rc * box_area(rlo, rhi);
            const float split_cost = (cost > 0.0f) ? cost : F32_MAX;
            if (split_cost < best_cost) { best_axis = a; best_pos = pos; best_cost = split_cost; }
        }
    }
    if (best_cost >= parent_cost) { store(); return; }                                    // leaf (bvh.rs:94): the range unchanged
    uint32_t i = 0, j = n - 1u;                                                           // bvh.rs:99-108, in place
    while (i <= j) {
        if (cen(i, best_axis) < best_pos) i++;
        else {
            for (int w = 0; w < 7; w++) { const float x = s_p[i][w][t]; s_p[i][w][t] = s_p[j][w][t]; s_p[j][w][t] = x; }
            if (j == 0u) break;
            j--;
        }
    }
    const uint32_t k = i;
    store();
    if (k == 0u || k == n) return;                                                        // bvh.rs:110-113
    float alo[3], ahi[3], blo[3], bhi[3];
    for (int q = 0; q < 3; q++) { alo[q] = F32_MAX; ahi[q] = -F32_MAX; blo[q] = F32_MAX; bhi[q] = -F32_MAX; }
    for (uint32_t u = 0; u < k; u++) for (int q = 0; q < 3; q++) { alo[q] = fminf(alo[q], s_p[u][q][t]); ahi[q] = fmaxf(ahi[q], s_p[u][3 + q][t]); }
    for (uint32_t u = k; u < n; u++) for (int q = 0; q < 3; q++) { blo[q] = fminf(blo[q], s_p[u][q][t]); bhi[q] = fmaxf(bhi[q], s_p[u][3 + q][t]); }
    emit_children(bn, ctrl, ls, next_parity, node_i, Box3{alo[0], alo[1], alo[2], ahi[0], ahi[1], ahi[2]},
                  Box3{blo[0], blo[1], blo[2], bhi[0], bhi[1], bhi[2]}, nd.first, k, n);
}

// ---- nodes with <= kSub triangles: one thread builds the node's WHOLE subtree (round 3) --------------------------------
// The level-synchronous scheme spent most of its time here: one launch, one host read-back and one pass over global memory per
// tree level for nodes that hold a handful of triangles (31 levels at 10 M triangles, the last ~12 of them nothing but such nodes).
// Now the thread that reaches a node with <= kSub triangles loads its proxies into LDS ([slot][word][thread]: conflict-free) and
// runs the reference's recursion to the end -- split_node (bvh.rs:56-136) with evaluate_sah (:138-161) as written, the partition
// loop (:99-108) in place, children pushed in pairs and then left before right (:131-135) -- with an explicit stack.  The
// subtree's nodes go to a pool in exactly the order the reference appends them, so their final indices are base(X) + local index
// (desc(X) = [A, B] ++ desc(A) ++ desc(B)); sizes_level / bases_level treat X as a node with `size` descendants and place_node
// copies the pool block with its child indices re-based.  A child's box is recomputed from its range when the child is popped:
// the same min/max over the same proxies in the same order as the parent's own child-box loops (bvh.rs:115-130), so the same bits.
struct PoolNode { float lo[3]; uint32_t a; float hi[3]; uint32_t n; };            // a: local index of the left child (inner) / first triangle (leaf)
constexpr uint32_t kSubRoot = 0xffffu;

__global__ __launch_bounds__(64) void build_subtree_tiny(BNode *bn, const uint32_t *__restrict__ list, uint32_t count, const Proxy *pin,
                                                         Proxy *pa, Proxy *pb, PoolNode *__restrict__ pool, Ctrl *ctrl) {   // pin is pa or pb
    __shared__ float s_p[kSub][7][64];                    // lo.xyz, hi.xyz, idx (as bits); the centroid is (lo + hi) / 2 recomputed (scene.rs:125)
    __shared__ uint32_t s_st[kSub + 1][64];
    const uint32_t t = threadIdx.x, item = blockIdx.x * 64u + t;
    const bool act = item < count;
    uint32_t node_i = 0, n0 = 0, first0 = 0;
    if (act) {
        node_i = list[item];
        n0 = bn[node_i].n; first0 = bn[node_i].first;
        for (uint32_t i = 0; i < n0; i++) {
            const Proxy p = pin[first0 + i];
            for (int q = 0; q < 3; q++) { s_p[i][q][t] = p.lo[q]; s_p[i][3 + q][t] = p.hi[q]; }
            s_p[i][6][t] = __uint_as_float(p.idx);
        }
    }
    // pool space for the most nodes n0 triangles can make (2 n0 - 2), one atomic per wave
    uint32_t need = act ? 2u * n0 - 2u : 0u, pre = need;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(pre, o); if (t >= (uint32_t)o) pre += y; }
    uint32_t base = 0;
    if (t == 63u) base = atomicAdd(&ctrl->pool_alloc.v, pre);
    base = __shfl(base, 63) + pre - need;
    if (!act) return;

    auto cen = [&](uint32_t i, int a) { return (s_p[i][a][t] + s_p[i][3 + a][t]) / 2.0f; };
    uint32_t cnt = 0, sp = 0;
    s_st[sp++][t] = 0u | (n0 << 8) | (kSubRoot << 16);
    while (sp > 0u) {
        const uint32_t e = s_st[--sp][t];
        const uint32_t f = e & 255u, n = (e >> 8) & 255u, li = e >> 16;
        float lo[3] = {F32_MAX, F32_MAX, F32_MAX}, hi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
        for (uint32_t i = f; i < f + n; i++)
            for (int q = 0; q < 3; q++) { lo[q] = fminf(lo[q], s_p[i][q][t]); hi[q] = fmaxf(hi[q], s_p[i][3 + q][t]); }
        if (li == kSubRoot) {                                // the subtree's root already has its box (from its parent: the same loop)
            const BNode nd = bn[node_i];
            for (int q = 0; q < 3; q++) { lo[q] = nd.lo[q]; hi[q] = nd.hi[q]; }
        }
        const float parent_cost = (float)n * box_area(lo, hi);
        int best_axis = 0;
        float best_pos = 0.0f, best_cost = F32_MAX;
        for (int a = 0; a < 3; a++) {
            float cmin = F32_MAX, cmax = -F32_MAX;
            for (uint32_t i = f; i < f + n; i++) { const float c = cen(i, a); cmin = fminf(cmin, c); cmax = fmaxf(cmax, c); }
            if (cmin == cmax) continue;
            const float scale = (cmax - cmin) / 8.0f;
            for (int i = 1; i < 8; i++) {
                const float pos = cmin + (float)i * scale;
                float llo[3] = {F32_MAX, F32_MAX, F32_MAX}, lhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
                float rlo[3] = {F32_MAX, F32_MAX, F32_MAX}, rhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
                uint32_t lc = 0, rc = 0;
                for (uint32_t j = f; j < f + n; j++) {                                        // evaluate_sah, bvh.rs:138-161
                    const bool left = cen(j, a) < pos;
                    for (int q = 0; q < 3; q++) {
                        const float pl = s_p[j][q][t], ph = s_p[j][3 + q][t];
                        if (left) { llo[q] = fminf(llo[q], pl); lhi[q] = fmaxf(lhi[q], ph); }
                        else { rlo[q] = fminf(rlo[q], pl); rhi[q] = fmaxf(rhi[q], ph); }
                    }
                    if (left) lc++; else rc++;
                }
                const float cost = (float)lc * box_area(llo, lhi) + (float)rc * box_area(rlo, rhi);
                const float split_cost = (cost > 0.0f) ? cost : F32_MAX;
                if (split_cost < best_cost) { best_axis = a; best_pos = pos; best_cost = split_cost; }
            }
        }
        bool split = !(best_cost >= parent_cost);                                              // bvh.rs:94
        uint32_t k = 0;
        if (split) {
            uint32_t i = f, j = f + n - 1u;                                                    // bvh.rs:99-108, in place
            while (i <= j) {
                if (cen(i, best_axis) < best_pos) i++;
                else {
                    for (int w = 0; w < 7; w++) { const float x = s_p[i][w][t]; s_p[i][w][t] = s_p[j][w][t]; s_p[j][w][t] = x; }
                    if (j == 0u) break;
                    j--;
                }
            }
            k = i - f;
            if (k == 0u || k == n) split = false;                                              // bvh.rs:110-113
        }
        PoolNode out;
        for (int q = 0; q < 3; q++) { out.lo[q] = lo[q]; out.hi[q] = hi[q]; }
        if (split) {
            out.a = cnt; out.n = 0u;
            s_st[sp++][t] = (f + k) | ((n - k) << 8) | ((cnt + 1u) << 16);                       // right child: after the whole left subtree
            s_st[sp++][t] = f | (k << 8) | (cnt << 16);
            cnt += 2u;
        } else {
            out.a = first0 + f; out.n = n;
        }
        if (li != kSubRoot) pool[base + li] = out;
        else if (split) { bn[node_i].left = kSubFlag | base; bn[node_i].size = cnt; }          // size is final after the loop (below)
    }
    if (cnt) bn[node_i].size = cnt;
    for (uint32_t i = 0; i < n0; i++) {                                                        // the range in its final order, in both ping-pong buffers
        Proxy p;
        for (int q = 0; q < 3; q++) { p.lo[q] = s_p[i][q][t]; p.hi[q] = s_p[i][3 + q][t]; }
        p.idx = __float_as_uint(s_p[i][6][t]); p.pad = 0u;
        pa[first0 + i] = p; pb[first0 + i] = p;
    }
    {                                                                                          // subtree nodes made, one atomic per wave
        uint32_t tot = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        const unsigned long long m = __ballot(true);
        if (t == (uint32_t)(__ffsll((long long)m) - 1) && tot) atomicAdd(&ctrl->sub_nodes.v, tot);
    }
}

// ---- nodes with > kBig triangles: the same five steps, spread over one workgroup per kChunk elements ---------------------
struct BigState {
    uint32_t node;                 // BFS index
    uint32_t first, n;
    uint32_t cmin[3], cmax[3];     // keys
    float pos[3][8];
    int use[3];
    uint32_t key[3][8][6];
    uint32_t cnt[3][8];
    int split, axis;
    float splitpos;
    uint32_t k, n_holes;
    int k_known;                   // k taken from the bin counts of the winning plane (big_choose); else big_choose counts it
    int plane;                     // the winning plane's index 1..7 (bins below it hold the "<" elements)
    uint32_t ckey[2][6];           // child boxes (A, B) as keys
    uint32_t ccen[2][6];           // child centroid ranges (A, B) as keys: the next level's step 1, gathered by big_scatter
    float lo[3], hi[3];            // node bounds
};
struct ChunkInfo { uint32_t big, off, len, hole_cnt, tail_cnt, hole_base, tail_base, pad; };
struct ChunkBins { uint32_t cnt[3][8]; };   // per chunk: elements per (axis, bin) -- big_bin's by-product, from which big_count2 derives holes / tails without a pass

// One workgroup: a BigState per big node of the level and the chunk table (chunks of a node contiguous, increasing offset),
// all on the device -- the host only launches (its grids are sized by the bound n_tris / kChunk + nb and the surplus
// workgroups leave at once).
// The node's centroid range (bvh.rs:67-77) is already known: the root's from make_proxies, every other big node's from its parent's
// big_scatter (big_finish left the record index in the node's `dfs` field, which the final renumbering overwrites).  So the planes
// (bvh.rs:82-84) are computed here and the level needs no pass of its own for them.
constexpr int kSetupT = 1024;
__global__ __launch_bounds__(kSetupT) void big_setup(BigState *bs, const BNode *bn, const uint32_t *ids, uint32_t nb, ChunkInfo *ch,
                                                     uint32_t *chunk_begin, Ctrl *ctrl, const uint32_t *rootkeys, const uint32_t *crange) {
    __shared__ uint32_t s_warp[kSetupT / 64];
    uint32_t running = 0;
    for (uint32_t base = 0; base < nb; base += kSetupT) {
        const uint32_t j = base + threadIdx.x;
        uint32_t my_chunks = 0;
        if (j < nb) {
            const BNode nd = bn[ids[j]];
            BigState &s = bs[j];                                                         // written in place: a local copy of the 700-byte record lives in scratch
            s.node = ids[j]; s.first = nd.first; s.n = nd.n;
            s.split = 0; s.axis = 0; s.splitpos = 0.0f; s.k = 0u; s.n_holes = 0u; s.k_known = 0; s.plane = 0;
            uint32_t cr[6];
            if (ids[j] == 0u) { for (int q = 0; q < 6; q++) cr[q] = rootkeys[6 + q]; }
            else { for (int q = 0; q < 6; q++) cr[q] = crange[(size_t)(nd.dfs - 1u) * 6u + (uint32_t)q]; }
            for (int a = 0; a < 3; a++) {
                s.cmin[a] = cr[a]; s.cmax[a] = cr[3 + a]; s.lo[a] = nd.lo[a]; s.hi[a] = nd.hi[a];
                const float cmin = funkey(cr[a]), cmax = funkey(cr[3 + a]);
                s.use[a] = !(cmin == cmax);                                              // bvh.rs:78
                const float scale = (cmax - cmin) / 8.0f;                                // bvh.rs:82
                s.pos[a][0] = 0.0f;
                for (int i = 1; i < 8; i++) s.pos[a][i] = cmin + (float)i * scale;       // bvh.rs:84
            }
            for (int a = 0; a < 3; a++) for (int k = 0; k < 8; k++) { s.cnt[a][k] = 0u; for (int q = 0; q < 6; q++) s.key[a][k][q] = q < 3 ? 0xffffffffu : 0u; }
            for (int sd = 0; sd < 2; sd++) for (int q = 0; q < 6; q++) { s.ckey[sd][q] = q < 3 ? 0xffffffffu : 0u; s.ccen[sd][q] = q < 3 ? 0xffffffffu : 0u; }
            my_chunks = (nd.n + kChunk - 1u) / kChunk;
        }
        uint32_t tot;
        const uint32_t off0 = running + block_exscan<kSetupT / 64>(my_chunks, s_warp, &tot);
        if (j < nb) chunk_begin[j] = off0;
        running += tot;
    }
    if (threadIdx.x == 0) { chunk_begin[nb] = running; ctrl->n_chunks.v = running; }
    __threadfence_block();
    __syncthreads();
    // the chunk table, all threads: chunk g belongs to the last node whose first chunk is <= g (the root alone has 1 221 chunks)
    for (uint32_t g = threadIdx.x; g < running; g += kSetupT) {
        uint32_t lo = 0, hi = nb;                                                        // chunk_begin[lo] <= g < chunk_begin[hi]
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) / 2u; if (chunk_begin[mid] <= g) lo = mid; else hi = mid; }
        const uint32_t n = bs[lo].n, c = g - chunk_begin[lo];
        ChunkInfo ci;
        ci.big = lo; ci.off = c * kChunk; ci.len = n - ci.off < kChunk ? n - ci.off : kChunk;
        ci.hole_cnt = ci.tail_cnt = ci.hole_base = ci.tail_base = ci.pad = 0u;
        ch[g] = ci;
    }
}
__global__ __launch_bounds__(kT) void big_bin(BigState *bs, const ChunkInfo *ch, ChunkBins *cbins, const Proxy *__restrict__ pin, const Ctrl *ctrl) {
    __shared__ uint32_t s_keyc[kCopies][3][8][6];
    __shared__ uint32_t s_cntc[kCopies][3][8];
    __shared__ float s_pos[3][8];
    __shared__ int s_use[3];
    if (blockIdx.x >= ctrl->n_chunks.v) return;
    const ChunkInfo c = ch[blockIdx.x];
    BigState *b = bs + c.big;
    const Proxy *in = pin + b->first + c.off;
    for (uint32_t i = threadIdx.x; i < kCopies * 144u; i += kT) (&s_keyc[0][0][0][0])[i] = ((i % 6) < 3) ? 0xffffffffu : 0u;
    for (uint32_t i = threadIdx.x; i < kCopies * 24u; i += kT) (&s_cntc[0][0][0])[i] = 0u;
    if (threadIdx.x < 24) (&s_pos[0][0])[threadIdx.x] = (&b->pos[0][0])[threadIdx.x];
    if (threadIdx.x < 3) s_use[threadIdx.x] = b->use[threadIdx.x];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < c.len; i += kT) {
        const Proxy p = in[i];
        for (int a = 0; a < 3; a++) {
            if (!s_use[a]) continue;
            int k = 8;
            for (int j = 1; j < 8; j++) if (p.c(a) < s_pos[a][j]) { k = j; break; }
            uint32_t *key = s_keyc[threadIdx.x & (kCopies - 1u)][a][k - 1];
            for (int q = 0; q < 3; q++) { lds_min(&key[q], fkey(p.lo[q])); lds_max(&key[3 + q], fkey(p.hi[q])); }
            atomicAdd(&s_cntc[threadIdx.x & (kCopies - 1u)][a][k - 1], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 144u) {
        const uint32_t i = threadIdx.x;
        uint32_t v = (&s_keyc[0][0][0][0])[i];
        for (uint32_t cc = 1; cc < kCopies; cc++) { const uint32_t w = (&s_keyc[cc][0][0][0])[i]; v = ((i % 6u) < 3u) ? (w < v ? w : v) : (w > v ? w : v); }
        if ((i % 6) < 3) atomicMin(&(&b->key[0][0][0])[i], v); else atomicMax(&(&b->key[0][0][0])[i], v);
    } else if (threadIdx.x < 168u) {
        const uint32_t i = threadIdx.x - 144u;
        uint32_t v = 0;
        for (uint32_t cc = 0; cc < kCopies; cc++) v += (&s_cntc[cc][0][0])[i];
        if (v) atomicAdd(&(&b->cnt[0][0])[i], v);
        (&cbins[blockIdx.x].cnt[0][0])[i] = v;
    }
}
// One wave per node; lane 0 picks the split.  If no plane was usable and the node splits all the same (a parent cost that is NaN or
// infinite: `best >= parent` is false), the split position is 0.0 on axis 0 and k has to be counted: the wave does that here over the
// whole node -- never seen on finite geometry, and a kernel of its own for it (one workgroup per chunk, leaving at once) sat in the
// chain of every level, up to 190 us of it waiting for a free CU under the wave kernels.
__global__ __launch_bounds__(64) void big_choose(BigState *bs, uint32_t nb, const Proxy *__restrict__ pin) {   // same expression order as wave_node's step 3
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= nb) return;
    BigState *b = bs + j;
    int need_count = 0, axis = 0;
    float pos = 0.0f;
    if (lane == 0) {
        const float parent_cost = (float)b->n * box_area(b->lo, b->hi);
        float best_cost = F32_MAX, best_pos = 0.0f;
        int best_axis = 0, best_plane = 0;
        uint32_t best_k = kNone;
        for (int a = 0; a < 3; a++) {
            if (!b->use[a]) continue;
            float rlo[8][3], rhi[8][3];
            uint32_t rcnt[8];
            float alo[3] = {F32_MAX, F32_MAX, F32_MAX}, ahi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
            uint32_t c = 0;
            for (int k = 7; k >= 0; k--) {
                for (int q = 0; q < 3; q++) { alo[q] = fminf(alo[q], funkey(b->key[a][k][q])); ahi[q] = fmaxf(ahi[q], funkey(b->key[a][k][3 + q])); }
                c += b->cnt[a][k];
                for (int q = 0; q < 3; q++) { rlo[k][q] = alo[q]; rhi[k][q] = ahi[q]; }
                rcnt[k] = c;
            }
            float llo[3] = {F32_MAX, F32_MAX, F32_MAX}, lhi[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
            uint32_t lc = 0;
            for (int i = 1; i < 8; i++) {
                for (int q = 0; q < 3; q++) { llo[q] = fminf(llo[q], funkey(b->key[a][i - 1][q])); lhi[q] = fmaxf(lhi[q], funkey(b->key[a][i - 1][3 + q])); }
                lc += b->cnt[a][i - 1];
                const float cost = (float)lc * box_area(llo, lhi) + (float)rcnt[i] * box_area(rlo[i], rhi[i]);
                const float split_cost = (cost > 0.0f) ? cost : F32_MAX;
                if (split_cost < best_cost) { best_axis = a; best_pos = b->pos[a][i]; best_cost = split_cost; best_k = lc; best_plane = i; }
            }
        }
        b->split = !(best_cost >= parent_cost);
        b->axis = best_axis; b->splitpos = best_pos; b->plane = best_plane;
        b->k_known = best_k != kNone;
        if (b->k_known) b->k = best_k;
        need_count = (b->split && !b->k_known) ? 1 : 0;
        axis = best_axis; pos = best_pos;
    }
    need_count = __shfl(need_count, 0);
    if (!need_count) return;
    axis = __shfl(axis, 0); pos = __shfl(pos, 0);
    const Proxy *in = pin + b->first;
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < b->n; i += 64u) cnt += (in[i].cax(axis) < pos) ? 1u : 0u;
    cnt = wave_sum(cnt);
    if (lane == 0) b->k = cnt;
}
// holes (positions < k holding ">=") and tail "<" (positions >= k holding "<") per chunk.  With k taken from the winning plane's bin
// counts the chunk's own counts say how many of its elements are "<" (bins below the plane: the planes increase with their index), so
// a chunk that lies wholly below or above k needs no look at its elements; only the one chunk per node that straddles k is read.
__global__ __launch_bounds__(kT) void big_count2(BigState *bs, ChunkInfo *ch, const ChunkBins *cbins, const Proxy *__restrict__ pin, const Ctrl *ctrl) {
    __shared__ uint32_t s_warp[4];
    if (blockIdx.x >= ctrl->n_chunks.v) return;
    ChunkInfo c = ch[blockIdx.x];
    BigState *b = bs + c.big;
    if (!b->split) return;
    const int axis = b->axis; const float pos = b->splitpos; const uint32_t k = b->k;
    if (b->k_known && (c.off + c.len <= k || c.off >= k)) {
        if (threadIdx.x == 0) {
            uint32_t less = 0;
            for (int j = 0; j < b->plane; j++) less += cbins[blockIdx.x].cnt[axis][j];
            const bool below = c.off + c.len <= k;
            ch[blockIdx.x].hole_cnt = below ? c.len - less : 0u;
            ch[blockIdx.x].tail_cnt = below ? 0u : less;
        }
        return;
    }
    const Proxy *in = pin + b->first + c.off;
    uint32_t holes = 0, tails = 0;
    for (uint32_t i = threadIdx.x; i < c.len; i += kT) {
        const uint32_t p = c.off + i;
        const bool l = in[i].cax(axis) < pos;
        holes += (p < k && !l) ? 1u : 0u;
        tails += (p >= k && l) ? 1u : 0u;
    }
    uint32_t th, tt;
    (void)block_exscan(holes, s_warp, &th);
    (void)block_exscan(tails, s_warp, &tt);
    if (threadIdx.x == 0) { ch[blockIdx.x].hole_cnt = th; ch[blockIdx.x].tail_cnt = tt; }
}
// one wave per big node: chunk bases (chunks of a node are contiguous in the chunk table, increasing offset); 64 chunks per
// step, a wave-level exclusive scan plus a running carry -- the root has 1 221 chunks, which one thread scanned in 0.28 ms
__global__ __launch_bounds__(64) void big_scan(BigState *bs, ChunkInfo *ch, const uint32_t *chunk_begin, uint32_t nb) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    if (j >= nb || !bs[j].split) return;
    const uint32_t cb = chunk_begin[j], ce = chunk_begin[j + 1];
    uint32_t acc = 0;
    for (uint32_t base = cb; base < ce; base += 64u) {                   // holes: increasing chunk index
        const uint32_t c = base + lane;
        const uint32_t v = c < ce ? ch[c].hole_cnt : 0u;
        uint32_t x = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
        if (c < ce) ch[c].hole_base = acc + x - v;
        acc += __shfl(x, 63);
    }
    if (lane == 0) bs[j].n_holes = acc;
    acc = 0;
    for (uint32_t done = 0; cb + done < ce; done += 64u) {               // tails: decreasing chunk index
        const uint32_t q = done + lane;                                  // q-th chunk from the end
        const bool in = q < ce - cb;
        const uint32_t c = ce - 1u - q;
        const uint32_t v = in ? ch[c].tail_cnt : 0u;
        uint32_t x = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
        if (in) ch[c].tail_base = acc + x - v;
        acc += __shfl(x, 63);
    }
}
// Ranks of the set flags among kE * kT elements per step, element (j, t) = j * kT + t: one ballot per j, the four waves' totals
// through LDS, ONE barrier per step (the table is double-buffered by step parity) -- and kE loads per thread in flight before it.
template <int kE> struct StepRanks { uint32_t cnt[2][kE][4]; };
template <int kE> __device__ __forceinline__ void step_ranks(const bool (&f)[kE], uint32_t (&rank)[kE], uint32_t *total, StepRanks<kE> *sr, uint32_t step) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    unsigned long long m[kE];
#pragma unroll
    for (int j = 0; j < kE; j++) m[j] = __ballot(f[j]);
    uint32_t (*tab)[4] = sr->cnt[step & 1u];
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < kE; j++) tab[j][w] = (uint32_t)__popcll(m[j]);
    }
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < kE; j++) {
        const uint32_t c0 = tab[j][0], c1 = tab[j][1], c2 = tab[j][2], c3 = tab[j][3];
        rank[j] = run + (w > 0 ? c0 : 0u) + (w > 1 ? c1 : 0u) + (w > 2 ? c2 : 0u) + mask_rank(m[j]);
        run += c0 + c1 + c2 + c3;
    }
    *total = run;
}
__global__ __launch_bounds__(kT) void big_fill(const BigState *bs, const ChunkInfo *ch, const Proxy *__restrict__ pin, uint32_t *hole_pos, uint32_t *tail_pos,
                                               const Ctrl *ctrl) {
    constexpr int kE = 8;
    __shared__ StepRanks<kE> s_sr;
    if (blockIdx.x >= ctrl->n_chunks.v) return;
    const ChunkInfo c = ch[blockIdx.x];
    const BigState *b = bs + c.big;
    if (!b->split) return;
    const Proxy *in = pin + b->first + c.off;
    uint32_t *hp = hole_pos + b->first, *tp = tail_pos + b->first;
    const int axis = b->axis; const float pos = b->splitpos; const uint32_t k = b->k;
    const uint32_t n_lo = c.off < k ? (k - c.off < c.len ? k - c.off : c.len) : 0u;     // the chunk's elements at positions < k
    const uint32_t n_hi = c.len - n_lo;
    uint32_t hb = c.hole_base, tb = c.tail_base, step = 0;
    for (uint32_t base = 0; base < n_lo; base += kT * kE, step++) {      // holes: increasing p
        bool f[kE];
#pragma unroll
        for (int j = 0; j < kE; j++) { const uint32_t i = base + (uint32_t)j * kT + threadIdx.x; f[j] = i < n_lo && !(in[i].cax(axis) < pos); }
        uint32_t rank[kE], tot;
        step_ranks(f, rank, &tot, &s_sr, step);
#pragma unroll
        for (int j = 0; j < kE; j++) if (f[j]) hp[hb + rank[j]] = c.off + base + (uint32_t)j * kT + threadIdx.x;
        hb += tot;
    }
    for (uint32_t base = 0; base < n_hi; base += kT * kE, step++) {      // tail "<": decreasing p
        bool f[kE];
#pragma unroll
        for (int j = 0; j < kE; j++) { const uint32_t q = base + (uint32_t)j * kT + threadIdx.x; f[j] = q < n_hi && (in[c.len - 1u - q].cax(axis) < pos); }
        uint32_t rank[kE], tot;
        step_ranks(f, rank, &tot, &s_sr, step);
#pragma unroll
        for (int j = 0; j < kE; j++) if (f[j]) tp[tb + rank[j]] = c.off + c.len - 1u - (base + (uint32_t)j * kT + threadIdx.x);
        tb += tot;
    }
}
// box and centroid range of one side of a split, in scalars (an array the compiler cannot prove constant-indexed goes to scratch)
struct SideAcc {
    float l0 = F32_MAX, l1 = F32_MAX, l2 = F32_MAX, h0 = -F32_MAX, h1 = -F32_MAX, h2 = -F32_MAX;
    float cl0 = F32_MAX, cl1 = F32_MAX, cl2 = F32_MAX, ch0 = -F32_MAX, ch1 = -F32_MAX, ch2 = -F32_MAX;
    __device__ __forceinline__ void add(const Proxy &e) {
        l0 = fminf(l0, e.lo[0]); l1 = fminf(l1, e.lo[1]); l2 = fminf(l2, e.lo[2]);
        h0 = fmaxf(h0, e.hi[0]); h1 = fmaxf(h1, e.hi[1]); h2 = fmaxf(h2, e.hi[2]);
        const float c0 = e.c(0), c1 = e.c(1), c2 = e.c(2);
        cl0 = fminf(cl0, c0); cl1 = fminf(cl1, c1); cl2 = fminf(cl2, c2);
        ch0 = fmaxf(ch0, c0); ch1 = fmaxf(ch1, c1); ch2 = fmaxf(ch2, c2);
    }
    __device__ __forceinline__ void flush(uint32_t *box_keys, uint32_t *cen_keys) const {       // keys: min xyz, max xyz (LDS)
        atomicMin(&box_keys[0], fkey(l0)); atomicMin(&box_keys[1], fkey(l1)); atomicMin(&box_keys[2], fkey(l2));
        atomicMax(&box_keys[3], fkey(h0)); atomicMax(&box_keys[4], fkey(h1)); atomicMax(&box_keys[5], fkey(h2));
        atomicMin(&cen_keys[0], fkey(cl0)); atomicMin(&cen_keys[1], fkey(cl1)); atomicMin(&cen_keys[2], fkey(cl2));
        atomicMax(&cen_keys[3], fkey(ch0)); atomicMax(&cen_keys[4], fkey(ch1)); atomicMax(&cen_keys[5], fkey(ch2));
    }
};
__global__ __launch_bounds__(kT) void big_scatter(BigState *bs, const ChunkInfo *ch, const Proxy *__restrict__ pin, Proxy *__restrict__ pout,
                                                  const uint32_t *hole_pos, const uint32_t *tail_pos, const Ctrl *ctrl) {
    constexpr int kE = 4;
    __shared__ StepRanks<kE> s_sr;
    __shared__ uint32_t s_ckey[2][6], s_ccen[2][6];
    if (blockIdx.x >= ctrl->n_chunks.v) return;
    const ChunkInfo c = ch[blockIdx.x];
    BigState *b = bs + c.big;
    const Proxy *in = pin + b->first + c.off;
    Proxy *out = pout + b->first;
    if (!b->split) { for (uint32_t i = threadIdx.x; i < c.len; i += kT) out[c.off + i] = in[i]; return; }
    const uint32_t *hp = hole_pos + b->first, *tp = tail_pos + b->first;
    const int axis = b->axis; const float pos = b->splitpos; const uint32_t k = b->k, n = b->n, n_holes = b->n_holes;
    if (threadIdx.x < 12) { (&s_ckey[0][0])[threadIdx.x] = ((threadIdx.x % 6) < 3) ? 0xffffffffu : 0u; (&s_ccen[0][0])[threadIdx.x] = ((threadIdx.x % 6) < 3) ? 0xffffffffu : 0u; }
    __syncthreads();
    // Child boxes, and the children's centroid ranges (the next level's bvh.rs:67-77: they fall out of this pass, every element's side is
    // known here).  One accumulator set per side, indexed by constants only -- a `[side]` index would send the arrays to scratch.
    SideAcc A, B;
    const uint32_t t_last = n_holes ? tp[n_holes - 1u] : n;
    const uint32_t n_lo = c.off < k ? (k - c.off < c.len ? k - c.off : c.len) : 0u;     // the chunk's elements at positions < k
    const uint32_t n_hi = c.len - n_lo;
    uint32_t hb = c.hole_base, tb = c.tail_base, step = 0;
    for (uint32_t base = 0; base < n_lo; base += kT * kE, step++) {      // positions < k, increasing
        Proxy e[kE];
        bool f[kE], valid[kE];
#pragma unroll
        for (int j = 0; j < kE; j++) {
            const uint32_t i = base + (uint32_t)j * kT + threadIdx.x;
            valid[j] = i < n_lo; f[j] = false;
            if (valid[j]) { e[j] = in[i]; f[j] = !(e[j].cax(axis) < pos); }
        }
        uint32_t rank[kE], tot;
        step_ranks(f, rank, &tot, &s_sr, step);
#pragma unroll
        for (int j = 0; j < kE; j++) {
            if (!valid[j]) continue;
            uint32_t dest = c.off + base + (uint32_t)j * kT + threadIdx.x;
            if (f[j]) { const uint32_t m = hb + rank[j]; dest = (m ? tp[m - 1u] : n) - 1u; }
            out[dest] = e[j];
            if (f[j]) B.add(e[j]); else A.add(e[j]);
        }
        hb += tot;
    }
    for (uint32_t base = 0; base < n_hi; base += kT * kE, step++) {      // positions >= k, decreasing
        Proxy e[kE];
        bool f[kE], valid[kE];
#pragma unroll
        for (int j = 0; j < kE; j++) {
            const uint32_t q = base + (uint32_t)j * kT + threadIdx.x;
            valid[j] = q < n_hi; f[j] = false;
            if (valid[j]) { e[j] = in[c.len - 1u - q]; f[j] = e[j].cax(axis) < pos; }
        }
        uint32_t rank[kE], tot;
        step_ranks(f, rank, &tot, &s_sr, step);
#pragma unroll
        for (int j = 0; j < kE; j++) {
            if (!valid[j]) continue;
            const uint32_t p = c.off + c.len - 1u - (base + (uint32_t)j * kT + threadIdx.x);
            uint32_t dest;
            if (f[j]) dest = hp[tb + rank[j]];
            else if (p > t_last) dest = p - 1u;
            else dest = (p == k) ? (t_last - 1u) : (p - 1u);
            out[dest] = e[j];
            if (f[j]) A.add(e[j]); else B.add(e[j]);
        }
        tb += tot;
    }
    A.flush(s_ckey[0], s_ccen[0]);
    B.flush(s_ckey[1], s_ccen[1]);
    __syncthreads();
    if (threadIdx.x < 12) {
        const uint32_t v = (&s_ckey[0][0])[threadIdx.x];
        if ((threadIdx.x % 6) < 3) atomicMin(&(&b->ckey[0][0])[threadIdx.x], v); else atomicMax(&(&b->ckey[0][0])[threadIdx.x], v);
    } else if (threadIdx.x < 24) {
        const uint32_t t = threadIdx.x - 12u, v = (&s_ccen[0][0])[t];
        if ((t % 6) < 3) atomicMin(&(&b->ccen[0][0])[t], v); else atomicMax(&(&b->ccen[0][0])[t], v);
    }
}
// crange: the records of the NEXT level's big nodes (one region per level parity, like the work lists)
__global__ void big_finish(const BigState *bs, BNode *bn, Ctrl *ctrl, Lists ls, uint32_t next_parity, uint32_t nb, uint32_t *crange, uint32_t crange_cap) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nb || !bs[j].split) return;
    const BigState &s = bs[j];
    if (s.k == 0u || s.k == s.n) return;                                // bvh.rs:110-113: everything on one side -- a leaf, AFTER the partition loop has reordered its triangles (the scatter did)
    const Box3 A{funkey(s.ckey[0][0]), funkey(s.ckey[0][1]), funkey(s.ckey[0][2]), funkey(s.ckey[0][3]), funkey(s.ckey[0][4]), funkey(s.ckey[0][5])};
    const Box3 B{funkey(s.ckey[1][0]), funkey(s.ckey[1][1]), funkey(s.ckey[1][2]), funkey(s.ckey[1][3]), funkey(s.ckey[1][4]), funkey(s.ckey[1][5])};
    const uint32_t base = emit_children(bn, ctrl, ls, next_parity, s.node, A, B, s.first, s.k, s.n);
    const uint32_t cn[2] = {s.k, s.n - s.k};
    for (uint32_t sd = 0; sd < 2u; sd++) {                              // a child that is big itself starts its level with the range known
        if (cn[sd] <= kBig) continue;
        const uint32_t slot = atomicAdd(&ctrl->cnt[next_parity >> 8][kClasses].v, 1u);
        if (slot >= crange_cap) continue;                               // cannot happen: a level holds at most n / kBig big nodes (the host checks the counter all the same)
        for (int q = 0; q < 6; q++) crange[(size_t)slot * 6u + (uint32_t)q] = s.ccen[sd][q];
        bn[base + sd].dfs = slot + 1u;
    }
}

__global__ void sizes_level(BNode *bn, uint32_t begin, uint32_t end) {              // bottom-up: |desc(X)|
    for (uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        const uint32_t l = bn[i].left;
        if (l != kNone && (l & kSubFlag)) continue;                               // a finished subtree: size set by build_subtree_tiny
        bn[i].size = (l == kNone) ? 0u : 2u + bn[l].size + bn[l + 1].size;
    }
}
// Top-down: desc(X) = [A, B] ++ desc(A) ++ desc(B).  A node's own place (dfs) and its descendants' block (base) were set by its
// parent one level up, so the node is also WRITTEN here, in the reference's 32-byte format (a pass of its own over all nodes before).
__device__ __forceinline__ void place_node(BNode *bn, uint32_t i, const PoolNode *__restrict__ pool, MiptNode *nodes) {
    const BNode b = bn[i];
    MiptNode nd;
    nd.bounds_min = {b.lo[0], b.lo[1], b.lo[2]}; nd.bounds_max = {b.hi[0], b.hi[1], b.hi[2]};
    if (b.left == kNone) { nd.first_tri_or_child = b.first; nd.num_tris = b.n; }
    else { nd.first_tri_or_child = b.base; nd.num_tris = 0; }
    nodes[b.dfs] = nd;
    if (b.left == kNone) return;
    if (b.left & kSubFlag) {                                                       // its subtree: pool block -> nodes[base ...], child indices re-based
        const PoolNode *src = pool + (b.left & ~kSubFlag);
        for (uint32_t j = 0; j < b.size; j++) {
            const PoolNode q = src[j];
            MiptNode o;
            o.bounds_min = {q.lo[0], q.lo[1], q.lo[2]}; o.bounds_max = {q.hi[0], q.hi[1], q.hi[2]};
            o.first_tri_or_child = q.n ? q.a : b.base + q.a; o.num_tris = q.n;
            nodes[b.base + j] = o;
        }
        return;
    }
    const uint32_t l = b.left;
    bn[l].dfs = b.base; bn[l + 1].dfs = b.base + 1u;
    bn[l].base = b.base + 2u; bn[l + 1].base = b.base + 2u + bn[l].size;
}
__global__ void bases_level(BNode *bn, uint32_t begin, uint32_t end, const PoolNode *__restrict__ pool, MiptNode *nodes) {
    for (uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) place_node(bn, i, pool, nodes);
}
// The same two passes for a RUN of consecutive small levels in one launch (one workgroup, a barrier between levels): the top of the tree
// and its last few levels hold a handful of nodes each, and a launch per level and pass was 58 launches ~ 0.9 ms of a 17-ms build.
constexpr uint32_t kRunLevels = 32, kRunNodes = 4096;        // a run: up to 32 levels of at most 4 096 nodes each (wider levels pay more in one workgroup than a launch costs)
struct LevelRun { uint32_t b[kRunLevels + 1]; uint32_t n; };  // level i of the run = nodes [b[i], b[i + 1])
__global__ __launch_bounds__(1024) void sizes_run(BNode *bn, LevelRun r) {                // bottom-up
    for (int i = (int)r.n - 1; i >= 0; i--) {
        for (uint32_t j = r.b[i] + threadIdx.x; j < r.b[i + 1]; j += blockDim.x) {
            const uint32_t l = bn[j].left;
            if (l != kNone && (l & kSubFlag)) continue;
            bn[j].size = (l == kNone) ? 0u : 2u + bn[l].size + bn[l + 1].size;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void bases_run(BNode *bn, LevelRun r, const PoolNode *__restrict__ pool, MiptNode *nodes) {   // top-down
    for (uint32_t i = 0; i < r.n; i++) {
        for (uint32_t j = r.b[i] + threadIdx.x; j < r.b[i + 1]; j += blockDim.x) place_node(bn, j, pool, nodes);
        __syncthreads();
    }
}
__global__ void extract_order(const Proxy *px, uint32_t n, uint32_t *order) {        // reordered[t] = original[order[t]]
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) order[i] = px[i].idx;
}
__global__ void gather_tris(const MiptTriangle *src, const uint32_t *order, uint32_t n, MiptTriangle *dst) {
    // 112-B records as 7 x 16 B; one thread per (triangle, 16-B piece)
    const unsigned long long total = (unsigned long long)n * 7ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t t = (uint32_t)(i / 7ull), piece = (uint32_t)(i % 7ull);
        reinterpret_cast<float4 *>(dst)[(unsigned long long)t * 7ull + piece] = reinterpret_cast<const float4 *>(src)[(unsigned long long)order[t] * 7ull + piece];
    }
}

int fail(int code, const char *what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof buf, "mipt_bvh_build_device: %s: %s", what, hipGetErrorString(e));
    mipt_internal_set_error(buf);
    return code;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(MIPT_ERR_HIP, #x, e_); } } while (0)

} // namespace

// The build proper: triangles already in HBM (`d_tris`, original order), results stay in HBM -- the node array in the reference's
// order and the permutation BVH::build applied to the triangles (reordered[t] = original[tri_order[t]]).  Both are hipMalloc'ed here
// and owned by the caller.  Used by mipt_bvh_build_device (below) and by the device-resident scene setup (scene_device.hip).
// The runtime resolves a kernel the first time it is launched (~60 us each, 25 kernels: the first build of a process took 1.5-2 ms
// longer than the next).  A caller that is waiting for something else anyway (scene_device.hip: the upload) does it up front.
void mipt::bvh_builder_resolve_kernels() {
    const void *k[] = {(const void *)make_proxies, (const void *)init_root, (const void *)level_mark, (const void *)big_setup, (const void *)big_bin, (const void *)big_choose,
                       (const void *)big_count2, (const void *)big_scan, (const void *)big_fill, (const void *)big_scatter, (const void *)big_finish,
                       (const void *)build_level_wave<kWaveM, (kWaveMax > kWaveM ? kWaveMax : 2u * kWaveM), kWaveBigNodes, 2>, (const void *)build_level_wave<128u, kWaveM, 8, 3>,
                       (const void *)build_level_wave<kWaveS, 128u, 8, 6>, (const void *)build_level_wave<32u, kWaveS, 8, 8>, (const void *)build_level_group<32, 16u, 32u>,
                       (const void *)build_level_group<16, kTiny, 16u>, (const void *)build_level_tiny, (const void *)build_subtree_tiny, (const void *)sizes_level,
                       (const void *)sizes_run, (const void *)bases_level, (const void *)bases_run, (const void *)extract_order};
    hipFuncAttributes a;
    for (const void *f : k) (void)hipFuncGetAttributes(&a, f);
    (void)hipGetLastError();
}

int mipt::bvh_build_resident(const MiptTriangle *d_tris, uint32_t n_tris, int device_id, ResidentBvh *res) {
    if (!d_tris || !res || n_tris == 0) { mipt_internal_set_error("mipt_bvh_build_device: bad argument (empty scene: the reference panics)"); return MIPT_ERR_INVALID_ARG; }
    *res = ResidentBvh{};
    Proxy *d_px[2] = {nullptr, nullptr};
    BNode *d_bn = nullptr;
    MiptNode *d_nodes = nullptr;
    uint32_t *d_order = nullptr;
    PoolNode *d_pool = nullptr;
    uint32_t *d_hp = nullptr, *d_tp = nullptr, *d_root = nullptr, *d_cbeg = nullptr, *d_lists = nullptr, *d_crange = nullptr;
    Ctrl *d_ctrl = nullptr;
    LevelSnap *h_snap = nullptr;                            // pinned, written by level_mark
    BigState *d_big = nullptr;
    ChunkInfo *d_chunks = nullptr;
    ChunkBins *d_cbins = nullptr;
    const uint32_t big_cap = n_tris / kBig + 2u, chunk_cap = n_tris / kChunk + big_cap + 2u;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // The per-level kernels touch disjoint nodes and overlap on four streams: the chunked path; the 65..512 wave kernel (the longest of
    // a deep level); the other wave and group kernels; the two thread kernels.  (A stream costs ~6 ms to create the first time in a process.)
    hipStream_t sw = nullptr, sw2 = nullptr, sg = nullptr, ss = nullptr;
    auto cleanup = [&]() {
        hipStream_t all[] = {sw, sw2, sg, ss};
        for (hipStream_t x : all) if (x) (void)hipStreamDestroy(x);
        void *p[] = {d_px[0], d_px[1], d_bn, d_nodes, d_order, d_pool, d_hp, d_tp, d_ctrl, d_root, d_cbeg, d_big, d_chunks, d_cbins, d_crange, d_lists};
        for (void *q : p) if (q) (void)hipFree(q);
        if (h_snap) (void)hipHostFree(h_snap);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    HIP_TRY(hipSetDevice(device_id));
    const uint32_t max_nodes = 2u * n_tris;
    HIP_TRY(hipMalloc((void **)&d_px[0], (size_t)n_tris * sizeof(Proxy)));
    HIP_TRY(hipMalloc((void **)&d_px[1], (size_t)n_tris * sizeof(Proxy)));
    HIP_TRY(hipMalloc((void **)&d_bn, (size_t)max_nodes * sizeof(BNode)));
    HIP_TRY(hipMalloc((void **)&d_nodes, (size_t)max_nodes * sizeof(MiptNode)));
    HIP_TRY(hipMalloc((void **)&d_order, (size_t)n_tris * 4));
    HIP_TRY(hipMalloc((void **)&d_pool, (size_t)max_nodes * sizeof(PoolNode)));
    HIP_TRY(hipMalloc((void **)&d_hp, (size_t)n_tris * 4));
    HIP_TRY(hipMalloc((void **)&d_tp, (size_t)n_tris * 4));
    HIP_TRY(hipMalloc((void **)&d_ctrl, sizeof(Ctrl)));
    HIP_TRY(hipHostMalloc((void **)&h_snap, 2 * sizeof(LevelSnap), hipHostMallocDefault));
    memset(h_snap, 0, 2 * sizeof(LevelSnap));
    // work lists: a level has at most min(2^level, n_tris) nodes; a class list never holds more nodes than triangles / its
    // smallest node... sized by the simple bound n_tris + 1 per (parity, class)
    const size_t list_cap = (size_t)n_tris + 1u;
    HIP_TRY(hipMalloc((void **)&d_lists, (2 * (size_t)(kClasses - 1) * list_cap + 2 * (size_t)big_cap) * 4));
    Lists ls;
    for (int pa = 0; pa < 2; pa++) {
        int slot = 0;
        for (int c = 0; c < kClasses; c++) {
            if (c == CLS_BIG) continue;
            ls.l[pa][c] = d_lists + (size_t)(pa * (kClasses - 1) + slot++) * list_cap;
        }
        ls.l[pa][CLS_BIG] = d_lists + 2 * (size_t)(kClasses - 1) * list_cap + (size_t)pa * big_cap;
    }
    HIP_TRY(hipMalloc((void **)&d_root, 48));
    HIP_TRY(hipMalloc((void **)&d_cbins, (size_t)chunk_cap * sizeof(ChunkBins)));
    HIP_TRY(hipMalloc((void **)&d_crange, 2 * (size_t)big_cap * 6 * 4));
    HIP_TRY(hipMalloc((void **)&d_cbeg, (size_t)(big_cap + 1) * 4));
    HIP_TRY(hipMalloc((void **)&d_big, (size_t)big_cap * sizeof(BigState)));
    HIP_TRY(hipMalloc((void **)&d_chunks, (size_t)chunk_cap * sizeof(ChunkInfo)));
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    {   // blocking streams: ordered against the null stream's copies / launches.  The chunked path is a chain of eight dependent launches:
        // its stream gets the higher priority, so that a link of the chain is not left waiting for a CU under the wave kernels' workgroups
        // (-0.5 ms).  The others stay at the default priority: a stream at a priority the process has not used yet costs a new hardware
        // queue, 6-7 ms each the first time in a process (tools/setup_trace_first.py); creating them on a helper thread during the upload does
        // not hide that -- the runtime serialises queue creation with the copies (tried: the upload grew by what the creation took).
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_TRY(hipStreamCreateWithPriority(&sg, hipStreamDefault, prio_hi));
        HIP_TRY(hipStreamCreate(&sw));
        HIP_TRY(hipStreamCreate(&sw2));
        HIP_TRY(hipStreamCreate(&ss));
    }
    const uint32_t root_init[12] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIP_TRY(hipMemcpy(d_root, root_init, 48, hipMemcpyHostToDevice));
    Ctrl hc;
    memset(&hc, 0, sizeof hc);
    hc.n_nodes.v = 1u;
    {                                                       // the root goes straight into its class list (parity 0)
        const int cls = (int)node_class(n_tris);
        hc.cnt[0][cls].v = 1u;
        const uint32_t zero = 0u;
        HIP_TRY(hipMemcpy(ls.l[0][cls], &zero, 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(d_ctrl, &hc, sizeof hc, hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(make_proxies, dim3(2048), dim3(256), 0, nullptr, d_tris, n_tris, d_px[0], d_root);
    hipLaunchKernelGGL(init_root, dim3(1), dim3(1), 0, nullptr, d_bn, d_root, n_tris);
    std::vector<uint32_t> lvl_begin;
    uint32_t begin = 0, end = 1;
    int cur = 0;
    hipStream_t const sw3 = sw2, st = ss;
    uint32_t parity = 0, row = 0;                       // list parity = level & 1, counter row = level % 3
    while (begin < end) {                               // one round of launches per tree level; `end` strictly grows or the loop stops
        lvl_begin.push_back(begin);
        const uint32_t row_next = (row + 1u) % 3u, next = (parity ^ 1u) | (row_next << 8);
        const uint32_t nb = hc.cnt[row][CLS_BIG].v;
        if (nb) {                                       // top of the tree: nodes too large for one workgroup (chunks of kChunk);
                                                        // its own stream: these 8 launches overlap the level's wave / group / thread kernels
            const uint32_t nc = n_tris / kChunk + nb;   // bound on sum(ceil(n_j / kChunk)); the real count lives in ctrl->n_chunks
            if (nb > big_cap || nc > chunk_cap) { cleanup(); mipt_internal_set_error("mipt_bvh_build_device: internal capacity"); return MIPT_ERR_BVH; }
            const dim3 gb((nb + 63) / 64), tb(64);
            if (hc.cnt[row][kClasses].v > big_cap) { cleanup(); mipt_internal_set_error("mipt_bvh_build_device: internal capacity"); return MIPT_ERR_BVH; }
            hipLaunchKernelGGL(big_setup, dim3(1), dim3(kSetupT), 0, sg, d_big, d_bn, ls.l[parity][CLS_BIG], nb, d_chunks, d_cbeg, d_ctrl, d_root, d_crange + (size_t)parity * big_cap * 6);
            hipLaunchKernelGGL(big_bin, dim3(nc), dim3(kT), 0, sg, d_big, d_chunks, d_cbins, d_px[cur], d_ctrl);
            hipLaunchKernelGGL(big_choose, dim3(nb), dim3(64), 0, sg, d_big, nb, d_px[cur]);
            hipLaunchKernelGGL(big_count2, dim3(nc), dim3(kT), 0, sg, d_big, d_chunks, d_cbins, d_px[cur], d_ctrl);
            hipLaunchKernelGGL(big_scan, dim3(nb), dim3(64), 0, sg, d_big, d_chunks, d_cbeg, nb);
            hipLaunchKernelGGL(big_fill, dim3(nc), dim3(kT), 0, sg, d_big, d_chunks, d_px[cur], d_hp, d_tp, d_ctrl);
            hipLaunchKernelGGL(big_scatter, dim3(nc), dim3(kT), 0, sg, d_big, d_chunks, d_px[cur], d_px[cur ^ 1], d_hp, d_tp, d_ctrl);
            hipLaunchKernelGGL(big_finish, gb, tb, 0, sg, d_big, d_bn, d_ctrl, ls, next, nb, d_crange + (size_t)(parity ^ 1u) * big_cap * 6, big_cap);
        }
        const uint32_t ntin = hc.cnt[row][CLS_TINY].v, nsub = hc.cnt[row][CLS_SUB].v;
        const uint32_t nws = hc.cnt[row][CLS_WAVE].v, nwm = hc.cnt[row][CLS_WAVE_M].v, nwl = hc.cnt[row][CLS_WAVE_L].v;
        const uint32_t ng16 = hc.cnt[row][CLS_G16].v, ng32 = hc.cnt[row][CLS_G32].v, nwa = hc.cnt[row][CLS_WAVE_A].v;
        if (nwl) hipLaunchKernelGGL((build_level_wave<kWaveM, (kWaveMax > kWaveM ? kWaveMax : 2u * kWaveM), kWaveBigNodes, 2>), dim3((nwl + (uint32_t)kWaveBigNodes - 1u) / (uint32_t)kWaveBigNodes), dim3(64 * kWaveBigNodes), 0, sw3,
                                    d_bn, ls.l[parity][CLS_WAVE_L], nwl, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (nwm) hipLaunchKernelGGL((build_level_wave<128u, kWaveM, 8, 3>), dim3((nwm + 7u) / 8u), dim3(512), 0, sw, d_bn, ls.l[parity][CLS_WAVE_M], nwm, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (nwa) hipLaunchKernelGGL((build_level_wave<kWaveS, 128u, 8, 6>), dim3((nwa + 7u) / 8u), dim3(512), 0, sw2, d_bn, ls.l[parity][CLS_WAVE_A], nwa, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (nws) hipLaunchKernelGGL((build_level_wave<32u, kWaveS, 8, 8>), dim3((nws + 7u) / 8u), dim3(512), 0, sw2, d_bn, ls.l[parity][CLS_WAVE], nws, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (ng32) hipLaunchKernelGGL((build_level_group<32, 16u, 32u>), dim3((ng32 + 15u) / 16u), dim3(512), 0, sw2, d_bn, ls.l[parity][CLS_G32], ng32, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (ng16) hipLaunchKernelGGL((build_level_group<16, kTiny, 16u>), dim3((ng16 + 31u) / 32u), dim3(512), 0, sw2, d_bn, ls.l[parity][CLS_G16], ng16, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (ntin) hipLaunchKernelGGL(build_level_tiny, dim3((ntin + 63u) / 64u), dim3(64), 0, st, d_bn, ls.l[parity][CLS_TINY], ntin, d_px[cur], d_px[cur ^ 1], d_ctrl, ls, next);
        if (nsub) hipLaunchKernelGGL(build_subtree_tiny, dim3((nsub + 63u) / 64u), dim3(64), 0, ss, d_bn, ls.l[parity][CLS_SUB], nsub, d_px[cur], d_px[0], d_px[1], d_pool, d_ctrl);
        HIP_TRY(hipGetLastError());
        {   // the level's barrier and the next level's counts: level_mark on every stream that got work, then poll its flag
            hipStream_t used[4];
            uint32_t n_used = 0;
            if (nwm) used[n_used++] = sw;
            if (nwl || nwa || nws || ng32 || ng16) used[n_used++] = sw2;
            if (nb) used[n_used++] = sg;
            if (ntin || nsub) used[n_used++] = ss;
            for (int c = 0; c <= (int)kClasses; c++) hc.cnt[row][c].v = 0u;
            if (n_used) {
                const size_t lvl = lvl_begin.size();                    // level + 1: never 0
                LevelSnap *snap = h_snap + (lvl & 1u);
                for (uint32_t i = 0; i < n_used; i++) hipLaunchKernelGGL(level_mark, dim3(1), dim3(64), 0, used[i], d_ctrl, n_used, row_next, row, snap, (uint32_t)lvl);
                HIP_TRY(hipGetLastError());
                const auto t_spin = std::chrono::steady_clock::now();
                for (uint32_t spins = 0; snap->flag != (uint32_t)lvl; spins++) {
                    if ((spins & 0xfffffu) == 0xfffffu && std::chrono::steady_clock::now() - t_spin > std::chrono::seconds(20)) {
                        const hipError_t e = hipDeviceSynchronize();    // a kernel that faulted never raises the flag
                        cleanup();
                        mipt_internal_set_error(e != hipSuccess ? hipGetErrorString(e) : "mipt_bvh_build_device: level barrier timed out");
                        return MIPT_ERR_HIP;
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                hc.n_nodes.v = snap->n_nodes; hc.sub_nodes.v = snap->sub_nodes;
                for (int c = 0; c <= (int)kClasses; c++) hc.cnt[row_next][c].v = snap->cnt[c];
            }
        }
        const uint32_t total = hc.n_nodes.v;
        if (total > max_nodes) { cleanup(); mipt_internal_set_error("mipt_bvh_build_device: node overflow"); return MIPT_ERR_BVH; }
        begin = end; end = total; cur ^= 1; parity ^= 1u; row = row_next;
        if (lvl_begin.size() > 4096) { cleanup(); mipt_internal_set_error("mipt_bvh_build_device: tree deeper than 4096 levels"); return MIPT_ERR_BVH; }
    }
    const uint32_t n_bn = end;                                  // nodes built level by level; the finished subtrees' nodes live in the pool
    const uint32_t n_nodes = n_bn + hc.sub_nodes.v;
    lvl_begin.push_back(n_bn);
    if (n_nodes > max_nodes) { cleanup(); mipt_internal_set_error("mipt_bvh_build_device: node overflow"); return MIPT_ERR_BVH; }
    hipLaunchKernelGGL(extract_order, dim3(2048), dim3(256), 0, sw, d_px[cur], n_tris, d_order);      // beside the two tree passes below (every stream is idle here: the last level's barrier has been seen)
    {   // subtree sizes bottom-up, then depth-first bases top-down (which also writes the nodes): runs of small levels in one launch each, wide levels one by one
        const int n_lvl = (int)lvl_begin.size() - 1;
        auto small = [&](int l) { return lvl_begin[(size_t)l + 1] - lvl_begin[(size_t)l] <= kRunNodes; };
        for (int l = n_lvl - 1; l >= 0;) {
            if (!small(l)) {
                const uint32_t b = lvl_begin[(size_t)l], e = lvl_begin[(size_t)l + 1];
                hipLaunchKernelGGL(sizes_level, dim3((e - b + 255) / 256 < 2048 ? (e - b + 255) / 256 : 2048), dim3(256), 0, sg, d_bn, b, e);
                l--;
                continue;
            }
            int lo = l;
            while (lo - 1 >= 0 && small(lo - 1) && l - (lo - 1) + 1 <= (int)kRunLevels) lo--;
            LevelRun r;
            r.n = (uint32_t)(l - lo + 1);
            for (int i = 0; i <= l - lo + 1; i++) r.b[i] = lvl_begin[(size_t)(lo + i)];
            hipLaunchKernelGGL(sizes_run, dim3(1), dim3(1024), 0, sg, d_bn, r);
            l = lo - 1;
        }
        for (int l = 0; l < n_lvl;) {
            if (!small(l)) {
                const uint32_t b = lvl_begin[(size_t)l], e = lvl_begin[(size_t)l + 1];
                hipLaunchKernelGGL(bases_level, dim3((e - b + 255) / 256 < 2048 ? (e - b + 255) / 256 : 2048), dim3(256), 0, sg, d_bn, b, e, d_pool, d_nodes);
                l++;
                continue;
            }
            int hi = l;
            while (hi + 1 < n_lvl && small(hi + 1) && (hi + 1) - l + 1 <= (int)kRunLevels) hi++;
            LevelRun r;
            r.n = (uint32_t)(hi - l + 1);
            for (int i = 0; i <= hi - l + 1; i++) r.b[i] = lvl_begin[(size_t)(l + i)];
            hipLaunchKernelGGL(bases_run, dim3(1), dim3(1024), 0, sg, d_bn, r, d_pool, d_nodes);
            l = hi + 1;
        }
    }
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    res->d_nodes = d_nodes; res->n_nodes = n_nodes; res->d_tri_order = d_order; res->build_ms = ms;
    res->levels = (uint32_t)lvl_begin.size();
    d_nodes = nullptr; d_order = nullptr;                   // the caller's now
    cleanup();
    return MIPT_OK;
}

// The C ABI entry: host arrays in, host arrays out (reorders `tris` in place like bvh.rs:105).
extern "C" int mipt_bvh_build_device(MiptTriangle *tris, uint32_t n_tris, MiptNode *nodes_out, uint32_t nodes_cap,
                                     uint32_t *n_nodes_out, int device_id, double *build_ms_out) {
    if (!tris || !nodes_out || n_tris == 0 || nodes_cap == 0) { mipt_internal_set_error("mipt_bvh_build_device: bad argument (empty scene: the reference panics)"); return MIPT_ERR_INVALID_ARG; }
    MiptTriangle *d_tris = nullptr, *d_out = nullptr;
    mipt::ResidentBvh r;
    auto cleanup = [&]() {
        if (d_tris) (void)hipFree(d_tris);
        if (d_out) (void)hipFree(d_out);
        if (r.d_nodes) (void)hipFree(r.d_nodes);
        if (r.d_tri_order) (void)hipFree(r.d_tri_order);
    };
    HIP_TRY(hipSetDevice(device_id));
    const size_t nb = (size_t)n_tris * sizeof(MiptTriangle);
    HIP_TRY(hipMalloc((void **)&d_tris, nb));
    HIP_TRY(hipMalloc((void **)&d_out, nb));
    HIP_TRY(hipMemcpy(d_tris, tris, nb, hipMemcpyHostToDevice));
    const int rc = mipt::bvh_build_resident(d_tris, n_tris, device_id, &r);
    if (rc != MIPT_OK) { cleanup(); return rc; }
    if (r.n_nodes > nodes_cap) { cleanup(); mipt_internal_set_error("mipt_bvh_build_device: nodes_cap too small"); return MIPT_ERR_INVALID_ARG; }
    hipEvent_t g0 = nullptr, g1 = nullptr;                  // the 112-byte gather belongs to the build (bvh.rs:105 swaps the triangles themselves)
    HIP_TRY(hipEventCreate(&g0));
    if (hipEventCreate(&g1) != hipSuccess) { (void)hipEventDestroy(g0); cleanup(); mipt_internal_set_error("mipt_bvh_build_device: hipEventCreate failed"); return MIPT_ERR_HIP; }
    (void)hipEventRecord(g0, nullptr);
    hipLaunchKernelGGL(gather_tris, dim3(4096), dim3(256), 0, nullptr, d_tris, r.d_tri_order, n_tris, d_out);
    (void)hipEventRecord(g1, nullptr);
    hipError_t e = hipDeviceSynchronize();
    float gms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&gms, g0, g1);
    (void)hipEventDestroy(g0); (void)hipEventDestroy(g1);
    HIP_TRY(e);
    HIP_TRY(hipMemcpy(tris, d_out, nb, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nodes_out, r.d_nodes, (size_t)r.n_nodes * sizeof(MiptNode), hipMemcpyDeviceToHost));
    if (n_nodes_out) *n_nodes_out = r.n_nodes;
    if (build_ms_out) *build_ms_out = r.build_ms + gms;
    cleanup();
    return MIPT_OK;
}

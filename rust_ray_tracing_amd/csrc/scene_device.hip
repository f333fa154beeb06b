// scene_device.hip -- device-resident scene setup: mipt_scene_create_from_triangles.
//
// What the reference does between loading an OBJ and the first frame is BVH::build on the host (reference src/bvh.rs:13-54, minutes
// at 10 M triangles) and one upload of the flat arrays (src/renderer/backend/gpu.rs:329-339).  Here the triangle array crosses PCIe
// ONCE and everything else happens in HBM:
//   1. staged upload: a crew of copy threads moves chunks of the caller's (pageable) array into a ring of pinned buffers, the copy
//      engine drains the ring (measured: 22 ms for 1.12 GB against 213 ms for a plain hipMemcpy of untouched pageable memory,
//      tools/calib/h2d_rate.hip); the textures take the same road;
//   2. BVH::build on the GPU (bvh_build_device.hip, mipt::bvh_build_resident): node array in the reference's order + the triangle
//      permutation, both left in HBM;
//   3. the device layout of pt_kernel.h, built by the kernels below from those two arrays -- bit for bit what mipt_scene_create's
//      host code (mipt_api.cpp, bvh_build.cpp) builds from the same tree (tests/test_gpu_scene_device.py compares checksums):
//        * slots of the intersection stream (mipt::tri_slots): "doubles" first, in pair order, then the rest -- two prefix sums;
//        * order of the 64-B pair records (mipt::pair_order): the breadth-first walk of the tree, one prefix sum per level over
//          (couples, singles, parent+child lines, mate-less pairs); no host round trip per level, the level sizes stay on the device;
//        * the pair records, re-based on the new order; the two triangle streams, gathered straight from the caller's order through
//          the permutation (the 112-byte reordered triangle array of bvh.rs:105 is never materialised).
// The scan kernels are written here (reduce / scan-of-sums / apply over a fixed grid, 4 x u32 lanes per element) rather than taken
// from rocPRIM: they read their element count from device memory, which is what lets the level walk run without host synchronisation.
#include "../../include/mipt.h"
#include "mipt_internal.h"
#include "mipt_scene.h"
#include "copy_crew.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kT = 256;
constexpr int kScanGrid = 512;                  // workgroups of a scan pass (fixed: the element count lives on the device)
constexpr uint32_t kPad = 0xffffffffu;

int fail(int code, const std::string &msg) {
    mipt_internal_set_error(msg.c_str());
    return code;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- device-side state of the layout pass ------------------------------------------------------------------------------------
struct LevelState {
    uint32_t cnt;          // pairs in this level's list
    uint32_t order_base;   // entries of `order` in use before this level
    uint32_t lone_base;    // mate-less pairs collected before this level
    uint32_t start;        // where this level's entries of `order` begin (written by the level's scan-of-sums pass)
    uint32_t couples;      // entries of the next level that come from pairs with two inner children (they go first)
    uint32_t pad[3];
};
struct Ctl {
    LevelState lv[2];      // [depth & 1] = the level being processed, the other one = the next level
    uint32_t max_leaf, tiny_axes, bad_bound, bad_tri;
    uint32_t n_doubles, pad[3];
};

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 operator+(U4 a, U4 b) { return U4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

// block-wide exclusive scan of one U4 per thread (kT threads); *total = block sum
__device__ U4 block_exscan4(U4 v, U4 *s_warp, U4 *total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    U4 x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        U4 y{__shfl_up(x.x, o), __shfl_up(x.y, o), __shfl_up(x.z, o), __shfl_up(x.w, o)};
        if (lane >= (uint32_t)o) x = x + y;
    }
    if (lane == 63u) s_warp[w] = x;
    __syncthreads();
    U4 add{0, 0, 0, 0}, tot{0, 0, 0, 0};
    for (uint32_t i = 0; i < (uint32_t)(kT / 64); i++) { const U4 sw = s_warp[i]; if (i < w) add = add + sw; tot = tot + sw; }
    __syncthreads();
    *total = tot;
    return U4{add.x + x.x - v.x, add.y + x.y - v.y, add.z + x.z - v.z, add.w + x.w - v.w};
}
__device__ __forceinline__ uint32_t scan_chunk(uint32_t count) {           // elements per workgroup, a multiple of kT; same in every pass
    const uint32_t per = (count + (uint32_t)kScanGrid - 1u) / (uint32_t)kScanGrid;
    return (per + (uint32_t)kT - 1u) / (uint32_t)kT * (uint32_t)kT;
}

// Op: count() elements; value(i) -> U4; apply(i, value, exclusive prefix); finish(total) once, between the passes.
template <class Op> __global__ __launch_bounds__(kT) void scan_reduce(Op op, U4 *sums) {
    __shared__ U4 s_warp[kT / 64];
    const uint32_t count = op.count(), chunk = scan_chunk(count);
    const unsigned long long b = (unsigned long long)blockIdx.x * chunk;
    U4 acc{0, 0, 0, 0};
    if (b < count) {
        const uint32_t e = b + chunk < count ? (uint32_t)(b + chunk) : count;
        for (uint32_t i = (uint32_t)b + threadIdx.x; i < e; i += kT) acc = acc + op.value(i);
    }
    U4 tot;
    (void)block_exscan4(acc, s_warp, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
template <class Op> __global__ __launch_bounds__(kT) void scan_sums(Op op, U4 *sums) {       // one workgroup
    __shared__ U4 s_warp[kT / 64];
    U4 carry{0, 0, 0, 0};
    for (uint32_t base = 0; base < (uint32_t)kScanGrid; base += kT) {
        const uint32_t i = base + threadIdx.x;
        const U4 v = i < (uint32_t)kScanGrid ? sums[i] : U4{0, 0, 0, 0};
        U4 tot;
        const U4 ex = block_exscan4(v, s_warp, &tot);
        if (i < (uint32_t)kScanGrid) sums[i] = carry + ex;
        carry = carry + tot;
    }
    if (threadIdx.x == 0) op.finish(carry);
}
template <class Op> __global__ __launch_bounds__(kT) void scan_apply(Op op, const U4 *sums) {
    __shared__ U4 s_warp[kT / 64];
    const uint32_t count = op.count(), chunk = scan_chunk(count);
    const unsigned long long b = (unsigned long long)blockIdx.x * chunk;
    if (b >= count) return;
    const uint32_t e = b + chunk < count ? (uint32_t)(b + chunk) : count;
    U4 carry = sums[blockIdx.x];
    for (uint32_t base = (uint32_t)b; base < e; base += kT) {
        const uint32_t i = base + threadIdx.x;
        const U4 v = i < e ? op.value(i) : U4{0, 0, 0, 0};
        U4 tot;
        const U4 ex = block_exscan4(v, s_warp, &tot);
        if (i < e) op.apply(i, v, carry + ex);
        carry = carry + tot;
    }
}
template <class Op> void run_scan(const Op &op, U4 *sums, hipStream_t s) {
    hipLaunchKernelGGL(scan_reduce<Op>, dim3(kScanGrid), dim3(kT), 0, s, op, sums);
    hipLaunchKernelGGL(scan_sums<Op>, dim3(1), dim3(kT), 0, s, op, sums);
    hipLaunchKernelGGL(scan_apply<Op>, dim3(kScanGrid), dim3(kT), 0, s, op, sums);
}

// ---- 1. node checks: what mipt_scene_create validates / derives while it walks the node array (mipt_api.cpp) ----
__global__ void check_nodes(const MiptNode *nodes, uint32_t n_nodes, Ctl *ctl) {
    const float lim = 1.0995116e12f, tiny = 1.3234890e-23f /* 2^-76 */;
    uint32_t max_leaf = 0, tiny_axes = 0, bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += gridDim.x * blockDim.x) {
        const MiptNode n = nodes[i];
        const float b[3] = {n.bounds_min.x, n.bounds_min.y, n.bounds_min.z}, c[3] = {n.bounds_max.x, n.bounds_max.y, n.bounds_max.z};
        for (int k = 0; k < 3; k++) {
            if (!(fabsf(b[k]) <= lim) || !(fabsf(c[k]) <= lim)) bad = 1u;
            if ((b[k] != 0.0f && fabsf(b[k]) < tiny) || (c[k] != 0.0f && fabsf(c[k]) < tiny)) tiny_axes |= 1u << k;
        }
        if (n.num_tris > max_leaf) max_leaf = n.num_tris;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t m = __shfl_xor(max_leaf, o);
        max_leaf = m > max_leaf ? m : max_leaf;
        tiny_axes |= __shfl_xor(tiny_axes, o);
        bad |= __shfl_xor(bad, o);
    }
    if ((threadIdx.x & 63u) == 0u) {
        if (max_leaf) atomicMax(&ctl->max_leaf, max_leaf);
        if (tiny_axes) atomicOr(&ctl->tiny_axes, tiny_axes);
        if (bad) atomicOr(&ctl->bad_bound, 1u);
    }
}

// ---- 2. slots of the intersection stream (the order of mipt::tri_slots, bvh_build.cpp) ----
struct DoublesOp {          // over pairs k, in pair order: 2-triangle leaves and adjacent 1+1 leaf couples get a 128-B line to themselves
    const MiptNode *nodes; uint32_t n_pairs; uint32_t *slot; uint8_t *placed; Ctl *ctl;
    __device__ uint32_t count() const { return n_pairs; }
    __device__ U4 value(uint32_t k) const {
        const MiptNode l = nodes[2u * k + 1u], r = nodes[2u * k + 2u];
        if (l.num_tris == 1u && r.num_tris == 1u && r.first_tri_or_child == l.first_tri_or_child + 1u) return U4{2u, 0, 0, 0};
        return U4{(l.num_tris == 2u ? 2u : 0u) + (r.num_tris == 2u ? 2u : 0u), 0, 0, 0};
    }
    __device__ void apply(uint32_t k, U4 v, U4 ex) const {
        if (v.x == 0u) return;
        const MiptNode l = nodes[2u * k + 1u], r = nodes[2u * k + 2u];
        uint32_t next = ex.x;
        if (l.num_tris == 1u && r.num_tris == 1u && r.first_tri_or_child == l.first_tri_or_child + 1u) {
            slot[l.first_tri_or_child] = next; slot[r.first_tri_or_child] = next + 1u;
            placed[l.first_tri_or_child] = 1; placed[r.first_tri_or_child] = 1;
            return;
        }
        if (l.num_tris == 2u) { slot[l.first_tri_or_child] = next; slot[l.first_tri_or_child + 1u] = next + 1u; placed[l.first_tri_or_child] = 1; placed[l.first_tri_or_child + 1u] = 1; next += 2u; }
        if (r.num_tris == 2u) { slot[r.first_tri_or_child] = next; slot[r.first_tri_or_child + 1u] = next + 1u; placed[r.first_tri_or_child] = 1; placed[r.first_tri_or_child + 1u] = 1; }
    }
    __device__ void finish(U4 tot) const { ctl->n_doubles = tot.x; }
};
struct RestOp {             // over triangles: everything not placed above follows in the reference order
    uint32_t n_tris; uint32_t *slot; const uint8_t *placed; const Ctl *ctl;
    __device__ uint32_t count() const { return n_tris; }
    __device__ U4 value(uint32_t i) const { return U4{placed[i] ? 0u : 1u, 0, 0, 0}; }
    __device__ void apply(uint32_t i, U4 v, U4 ex) const { if (v.x) slot[i] = ctl->n_doubles + ex.x; }
    __device__ void finish(U4) const {}
};

// ---- 3. order of the pair records (the order of mipt::pair_order, bvh_build.cpp), one tree level per scan ----
struct LevelOp {
    const MiptNode *nodes; const uint32_t *cur; uint32_t *next; uint8_t *taken; uint32_t *order; uint32_t *lone; Ctl *ctl;
    uint32_t parity;        // depth & 1
    uint32_t bottom;        // depth >= mipt::pair_order_top()
    __device__ uint32_t count() const { return ctl->lv[parity].cnt; }
    __device__ __forceinline__ void children(uint32_t k, bool *ha, bool *hb, uint32_t *ca, uint32_t *cb) const {
        const uint32_t nl = nodes[2u * k + 1u].num_tris, nr = nodes[2u * k + 2u].num_tris;
        *ha = nl == 0u; *hb = nr == 0u;
        *ca = (nodes[2u * k + 1u].first_tri_or_child - 1u) / 2u;
        *cb = (nodes[2u * k + 2u].first_tri_or_child - 1u) / 2u;
    }
    __device__ U4 value(uint32_t i) const {
        const uint32_t k = cur[i];
        bool ha, hb; uint32_t ca, cb;
        children(k, &ha, &hb, &ca, &cb);
        U4 v{(ha && hb) ? 2u : 0u, (ha != hb) ? 1u : 0u, 0u, 0u};
        if (bottom && !taken[k]) { if (ha || hb) v.z = 1u; else v.w = 1u; }
        return v;
    }
    __device__ double half_area(uint32_t node) const {
        const MiptNode n = nodes[node];
        const double ex = (double)n.bounds_max.x - n.bounds_min.x, ey = (double)n.bounds_max.y - n.bounds_min.y, ez = (double)n.bounds_max.z - n.bounds_min.z;
        return ex * ey + ey * ez + ez * ex;
    }
    __device__ void apply(uint32_t i, U4 v, U4 ex) const {
        const LevelState st = ctl->lv[parity];
        const uint32_t k = cur[i];
        bool ha, hb; uint32_t ca, cb;
        children(k, &ha, &hb, &ca, &cb);
        if (ha && hb) { next[ex.x] = ca; next[ex.x + 1u] = cb; }      // couples first (an even count keeps them line-aligned), then singles
        else if (ha) next[st.couples + ex.y] = ca;
        else if (hb) next[st.couples + ex.y] = cb;
        if (!bottom) { order[st.start + i] = k; return; }              // tree top: breadth-first, the level in list order
        if (v.z) {                                                       // a line of its own with the child pair of its larger inner child
            uint32_t pick = ha ? ca : cb;
            if (ha && hb && half_area(2u * k + 2u) > half_area(2u * k + 1u)) pick = cb;
            order[st.start + 2u * ex.z] = k; order[st.start + 2u * ex.z + 1u] = pick;
            taken[pick] = 1;
        } else if (v.w) {
            lone[st.lone_base + ex.w] = k;
        }
    }
    __device__ void finish(U4 tot) const {
        LevelState &c = ctl->lv[parity], &n = ctl->lv[parity ^ 1u];
        n.cnt = 0; n.order_base = c.order_base; n.lone_base = c.lone_base;
        if (c.cnt == 0u) return;
        const uint32_t writes = bottom ? 2u * tot.z : c.cnt;
        c.start = writes ? ((c.order_base + 1u) & ~1u) : c.order_base;   // a level (top) / the first line (below) starts on a line boundary
        c.couples = tot.x;
        n.cnt = tot.x + tot.y;
        n.order_base = c.start + writes;
        n.lone_base = c.lone_base + tot.w;
    }
};
__global__ void append_lone(const uint32_t *lone, uint32_t *order, const Ctl *ctl, uint32_t parity) {
    const LevelState st = ctl->lv[parity];
    const uint32_t start = (st.order_base + 1u) & ~1u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < st.lone_base; i += gridDim.x * blockDim.x) order[start + i] = lone[i];
}

// ---- 4. records ----
__global__ void write_new_of(const uint32_t *order, uint32_t n_records, uint32_t *new_of) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_records; j += gridDim.x * blockDim.x) {
        const uint32_t k = order[j];
        if (k != kPad) new_of[k] = j;
    }
}
__global__ void write_pairs(const MiptNode *nodes, const uint32_t *order, uint32_t n_records, uint32_t n_records_padded, const uint32_t *new_of,
                            const uint32_t *slot, float4 *pairs) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_records_padded; j += gridDim.x * blockDim.x) {
        const uint32_t k = j < n_records ? order[j] : kPad;
        if (k == kPad) {
            for (int q = 0; q < 4; q++) pairs[(size_t)j * 4 + q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            continue;
        }
        for (uint32_t w = 0; w < 2u; w++) {
            const MiptNode n = nodes[2u * k + 1u + w];
            const uint32_t a = n.num_tris > 0u ? slot[n.first_tri_or_child] : new_of[(n.first_tri_or_child - 1u) / 2u];
            pairs[(size_t)j * 4 + w * 2u + 0u] = make_float4(n.bounds_min.x, n.bounds_min.y, n.bounds_min.z, __uint_as_float(a));
            pairs[(size_t)j * 4 + w * 2u + 1u] = make_float4(n.bounds_max.x, n.bounds_max.y, n.bounds_max.z, __uint_as_float(n.num_tris));
        }
    }
}
// Both triangle streams straight from the caller's order: reordered triangle t = tris[tri_order[t]].  The edges are one rounded f32
// subtraction each, the value the reference recomputes per test (ray.rs:24-25; -ffp-contract=off).
__global__ void write_tris(const MiptTriangle *tris, const uint32_t *tri_order, uint32_t n_tris, uint32_t n_materials, const uint32_t *slot,
                           float4 *tri_pos, float4 *tri_attr, Ctl *ctl) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tris; t += gridDim.x * blockDim.x) {
        const float4 *src = reinterpret_cast<const float4 *>(tris + (tri_order ? tri_order[t] : t));   // no permutation: the caller's triangles are in the tree's order already
        const float4 a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3], a4 = src[4], a5 = src[5], a6 = src[6];
        // Vertex = {position.xyz, u, normal.xyz, v}: a0 a1 | a2 a3 | a4 a5; a6.x = material_id
        const uint32_t mat = __float_as_uint(a6.x);
        if (mat >= n_materials) { atomicMin(&ctl->bad_tri, t); continue; }
        const float e1x = a2.x - a0.x, e1y = a2.y - a0.y, e1z = a2.z - a0.z;
        const float e2x = a4.x - a0.x, e2y = a4.y - a0.y, e2z = a4.z - a0.z;
        const size_t q = (size_t)slot[t] * 4;
        tri_pos[q + 0] = make_float4(a0.x, a0.y, a0.z, e1x);
        tri_pos[q + 1] = make_float4(e1y, e1z, e2x, e2y);
        tri_pos[q + 2] = make_float4(e2z, __uint_as_float(t), 0.0f, 0.0f);
        tri_pos[q + 3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        tri_attr[(size_t)t * 4 + 0] = make_float4(a1.x, a1.y, a1.z, a3.x);
        tri_attr[(size_t)t * 4 + 1] = make_float4(a3.y, a3.z, a5.x, a5.y);
        tri_attr[(size_t)t * 4 + 2] = make_float4(a5.z, a0.w, a1.w, a2.w);
        tri_attr[(size_t)t * 4 + 3] = make_float4(a3.w, a4.w, a5.w, a6.x);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) tri_pos[(size_t)n_tris * 4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the kernel's unconditional 4th-float4 load of the last record's neighbour
}

// ---- staged host -> device copies of large pageable arrays ----
// A crew of copy threads (started once per scene) moves chunk after chunk of the caller's memory into a ring of pinned buffers; the copy
// engine drains the ring.  8-MB chunks: small enough that pinning the ring is cheap (32 MB: ~8 ms to allocate, ~5 ms to free -- a
// 64-MB ring cost 26 ms per scene), large enough for full PCIe rate once the threads no longer have to be spawned per chunk.
class StagedUploader {
  public:
    static constexpr size_t kChunk = (size_t)8 << 20;
    static constexpr int kRing = 4;
    ~StagedUploader() { shut(); }
    // one copy; small ones go straight through hipMemcpy
    int copy(void *d_dst, const void *h_src, size_t bytes) {
        if (bytes == 0) return MIPT_OK;
        if (bytes < 2 * kChunk && !ready_) return direct(d_dst, h_src, bytes);
        if (!ready_) { const int rc = init(); if (rc) return rc == 1 ? direct(d_dst, h_src, bytes) : rc; }
        const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
        hipError_t e = hipSuccess;
        for (size_t c = 0; c < n_chunks && e == hipSuccess; c++, seq_++) {
            const int slot = (int)(seq_ % kRing);
            if (seq_ >= (size_t)kRing) e = hipEventSynchronize(ev_[slot]);     // the copy engine is done with this buffer
            if (e != hipSuccess) break;
            const size_t off = c * kChunk, len = off + kChunk <= bytes ? kChunk : bytes - off;
            crew_.copy((const char *)h_src + off, pin_[slot], len);
            e = hipMemcpyAsync((char *)d_dst + off, pin_[slot], len, hipMemcpyHostToDevice, stream_);
            if (e == hipSuccess) e = hipEventRecord(ev_[slot], stream_);
        }
        return e == hipSuccess ? MIPT_OK : fail(MIPT_ERR_HIP, std::string("staged upload: ") + hipGetErrorString(e));
    }
    int finish() {                                             // everything queued so far has arrived
        if (!ready_) return MIPT_OK;
        const hipError_t e = hipStreamSynchronize(stream_);
        return e == hipSuccess ? MIPT_OK : fail(MIPT_ERR_HIP, std::string("staged upload: ") + hipGetErrorString(e));
    }
    void pause() { crew_.stop(); }                            // the ring stays; the helpers stop spinning until the next copy
    void shut() {
        crew_.stop();
        if (stream_) (void)hipStreamSynchronize(stream_);
        for (int i = 0; i < kRing; i++) { if (pin_[i]) (void)hipHostFree(pin_[i]); if (ev_[i]) (void)hipEventDestroy(ev_[i]); pin_[i] = nullptr; ev_[i] = nullptr; }
        if (stream_) (void)hipStreamDestroy(stream_);
        stream_ = nullptr; ready_ = false;
    }

  private:
    int direct(void *d_dst, const void *h_src, size_t bytes) {
        const hipError_t e = hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice);
        return e == hipSuccess ? MIPT_OK : fail(MIPT_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
    }
    int init() {                                               // 0 ok, 1 = no pinned memory / stream: fall back to hipMemcpy, < 0 error
        hipError_t e = hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking);
        for (int i = 0; i < kRing && e == hipSuccess; i++) {
            e = hipHostMalloc((void **)&pin_[i], kChunk, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); shut(); return 1; }
        crew_.start();
        ready_ = true;
        return 0;
    }
    hipStream_t stream_ = nullptr;
    char *pin_[kRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_[kRing] = {nullptr, nullptr, nullptr, nullptr};
    mipt::CopyCrew crew_{4};
    size_t seq_ = 0;
    bool ready_ = false;
};

#define S_HIP(expr)                                                                                                    \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess) { cleanup(); return fail(MIPT_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e__)); } \
    } while (0)

} // namespace

// Both ways in end here.  `host_nodes`: the caller's node array (validated by mipt_scene_create, mipt_api.cpp) is uploaded beside the
// triangles, which are in the tree's order already; else BVH::build runs on the GPU.  The layout kernels are the same.
static int create_on_device(const MiptSceneDesc *desc, int device_id, bool host_nodes, MiptScene **out) {
    if (!desc || !out) return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_create_from_triangles: null argument");
    *out = nullptr;
    if (!desc->tris || desc->n_tris == 0) return fail(MIPT_ERR_INVALID_ARG, "scene has no triangles (the reference panics in BVH::build)");
    if (desc->n_tris > mipt::kMaxTris) return fail(MIPT_ERR_SCENE_LIMIT, std::to_string(desc->n_tris) + " triangles exceed the 2^25 device-format limit");
    if (host_nodes && (!desc->nodes || (desc->n_nodes & 1u) == 0u)) return fail(MIPT_ERR_BVH, "scene has no BVH nodes, or an even number of them");
    const double t_begin = now_ms();
    mipt::MaterialTables tables;
    { const int rc = mipt::build_material_tables(desc, &tables, false); if (rc) return rc; }     // the textures are staged below, not gathered on the host
    const uint32_t n_tris = desc->n_tris;

    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess) return fail(MIPT_ERR_HIP, std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e));
        if (device_id < 0 || device_id >= ndev) return fail(MIPT_ERR_HIP, "HIP device " + std::to_string(device_id) + " not available (" + std::to_string(ndev) + " visible)");
    }
    MiptTriangle *d_tris = nullptr;
    mipt::ResidentBvh bvh;
    char *arena = nullptr;
    MiptScene *s = nullptr;
    StagedUploader up_ring;
    auto cleanup = [&]() {
        up_ring.shut();
        if (d_tris) (void)hipFree(d_tris);
        if (bvh.d_nodes) (void)hipFree(bvh.d_nodes);
        if (bvh.d_tri_order) (void)hipFree(bvh.d_tri_order);
        if (arena) (void)hipFree(arena);
        if (s) mipt::free_scene(s);
    };
    S_HIP(hipSetDevice(device_id));
    // ---- 1. the one host -> device copy ----
    S_HIP(hipMalloc((void **)&d_tris, (size_t)n_tris * sizeof(MiptTriangle)));
    if (host_nodes) S_HIP(hipMalloc((void **)&bvh.d_nodes, (size_t)desc->n_nodes * sizeof(MiptNode)));
    {
        std::thread warm;
        if (!host_nodes) warm = std::thread([device_id]() { if (hipSetDevice(device_id) == hipSuccess) mipt::bvh_builder_resolve_kernels(); });   // beside the copies
        int rc = up_ring.copy(d_tris, desc->tris, (size_t)n_tris * sizeof(MiptTriangle));
        if (rc == MIPT_OK && host_nodes) rc = up_ring.copy(bvh.d_nodes, desc->nodes, (size_t)desc->n_nodes * sizeof(MiptNode));
        if (rc == MIPT_OK) rc = up_ring.finish();
        if (warm.joinable()) warm.join();
        up_ring.pause();                                      // back for the textures, after the build
        if (rc) { cleanup(); return rc; }
    }
    const double t_up = now_ms();
    // ---- 2. BVH::build in HBM (or the caller's tree) ----
    if (host_nodes) bvh.n_nodes = desc->n_nodes;
    else { const int rc = mipt::bvh_build_resident(d_tris, n_tris, device_id, &bvh); if (rc) { cleanup(); return rc; } }
    const double t_build = now_ms();
    const uint32_t n_nodes = bvh.n_nodes;
    if ((n_nodes & 1u) == 0u) { cleanup(); return fail(MIPT_ERR_BVH, "device builder returned an even node count"); }
    const uint32_t n_pairs = (n_nodes - 1u) / 2u;
    if (n_pairs > mipt::kMaxPairs) { cleanup(); return fail(MIPT_ERR_SCENE_LIMIT, std::to_string(n_pairs) + " node pairs exceed the 2^24 device-format limit"); }

    // ---- 3. layout kernels.  One arena for the temporaries ----
    const size_t np = n_pairs ? n_pairs : 1, order_cap = 2 * (size_t)n_pairs + 4;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_ctl = 0, o_sums = up(sizeof(Ctl)), o_slot = o_sums + up(sizeof(U4) * kScanGrid), o_placed = o_slot + up((size_t)n_tris * 4),
                 o_lvl0 = o_placed + up(n_tris), o_lvl1 = o_lvl0 + up(np * 4), o_lone = o_lvl1 + up(np * 4), o_taken = o_lone + up(np * 4),
                 o_order = o_taken + up(np), o_newof = o_order + up(order_cap * 4), arena_bytes = o_newof + up(np * 4);
    S_HIP(hipMalloc((void **)&arena, arena_bytes));
    Ctl *ctl = (Ctl *)(arena + o_ctl);
    U4 *sums = (U4 *)(arena + o_sums);
    uint32_t *slot = (uint32_t *)(arena + o_slot), *lvl[2] = {(uint32_t *)(arena + o_lvl0), (uint32_t *)(arena + o_lvl1)},
             *lone = (uint32_t *)(arena + o_lone), *order = (uint32_t *)(arena + o_order), *new_of = (uint32_t *)(arena + o_newof);
    uint8_t *placed = (uint8_t *)(arena + o_placed), *taken = (uint8_t *)(arena + o_taken);
    hipStream_t st = nullptr;                                      // the null stream: ordered after the builder's work
    {
        Ctl h;
        memset(&h, 0, sizeof h);
        h.lv[0].cnt = n_pairs ? 1u : 0u;                           // level 0 = { pair 0 } (the root's children)
        h.bad_tri = 0xffffffffu;
        S_HIP(hipMemcpy(ctl, &h, sizeof h, hipMemcpyHostToDevice));
        S_HIP(hipMemsetAsync(placed, 0, n_tris, st));
        S_HIP(hipMemsetAsync(taken, 0, np, st));
        S_HIP(hipMemsetAsync(order, 0xff, order_cap * 4, st));     // every entry a pad until a pair is written there
        S_HIP(hipMemsetAsync(lvl[0], 0, 4, st));
    }
    hipLaunchKernelGGL(check_nodes, dim3(1024), dim3(kT), 0, st, bvh.d_nodes, n_nodes, ctl);
    if (n_pairs) run_scan(DoublesOp{bvh.d_nodes, n_pairs, slot, placed, ctl}, sums, st);
    run_scan(RestOp{n_tris, slot, placed, ctl}, sums, st);
    // the level walk: batches of levels without a host round trip, then one look at the size of the next level
    const uint32_t top = mipt::pair_order_top();
    uint32_t depth = 0;
    Ctl hctl;
    memset(&hctl, 0, sizeof hctl);
    if (n_pairs) {
        for (;;) {
            for (int b = 0; b < 16; b++, depth++)
                run_scan(LevelOp{bvh.d_nodes, lvl[depth & 1u], lvl[(depth & 1u) ^ 1u], taken, order, lone, ctl, depth & 1u, depth >= top ? 1u : 0u}, sums, st);
            S_HIP(hipMemcpyAsync(&hctl, ctl, sizeof hctl, hipMemcpyDeviceToHost, st));
            S_HIP(hipStreamSynchronize(st));
            if (hctl.lv[depth & 1u].cnt == 0u) break;
            if (depth > 8192u) { cleanup(); return fail(MIPT_ERR_BVH, "BVH deeper than 8192 levels"); }
        }
        hipLaunchKernelGGL(append_lone, dim3(1024), dim3(kT), 0, st, lone, order, ctl, depth & 1u);
    } else {
        S_HIP(hipMemcpyAsync(&hctl, ctl, sizeof hctl, hipMemcpyDeviceToHost, st));
        S_HIP(hipStreamSynchronize(st));
    }
    if (hctl.bad_bound) { cleanup(); return fail(MIPT_ERR_SCENE_LIMIT, "a node has a non-finite bound or one beyond 2^40"); }
    const LevelState fin = hctl.lv[depth & 1u];
    uint32_t n_records = n_pairs ? ((fin.order_base + 1u) & ~1u) + fin.lone_base : 0u;
    if ((size_t)n_records + 1 > order_cap) { cleanup(); return fail(MIPT_ERR_BVH, "pair-record order overflow (internal)"); }
    const uint32_t n_records_padded = (n_records + 1u) & ~1u;      // one zero pad record: the triangle stream behind starts on a 128-B line
    if (n_records_padded > mipt::kMaxPairs) { cleanup(); return fail(MIPT_ERR_SCENE_LIMIT, "pair records (with line padding) exceed the 2^24 device-format limit"); }
    const size_t pairs_bytes = (size_t)n_records_padded * 64, pos_bytes = (size_t)n_tris * 64 + 16;
    if (pairs_bytes + pos_bytes >= 0xffffffffull) { cleanup(); return fail(MIPT_ERR_SCENE_LIMIT, "BVH + triangle stream exceed 4 GiB"); }

    s = new (std::nothrow) MiptScene();
    if (!s) { cleanup(); return fail(MIPT_ERR_INVALID_ARG, "out of host memory"); }
    s->device = device_id;
    s->n_tris = n_tris;
    s->geom_alloc = pairs_bytes + pos_bytes + 64;
    s->attr_bytes = (size_t)n_tris * 64;
    S_HIP(hipMalloc(&s->d_geom, s->geom_alloc));
    S_HIP(hipMalloc(&s->d_tri_attr, s->attr_bytes));
    float4 *d_pairs = (float4 *)s->d_geom, *d_pos = (float4 *)((char *)s->d_geom + pairs_bytes);
    if (n_records) hipLaunchKernelGGL(write_new_of, dim3(2048), dim3(kT), 0, st, order, n_records, new_of);
    if (n_records_padded) hipLaunchKernelGGL(write_pairs, dim3(2048), dim3(kT), 0, st, bvh.d_nodes, order, n_records, n_records_padded, new_of, slot, d_pairs);
    hipLaunchKernelGGL(write_tris, dim3(4096), dim3(kT), 0, st, d_tris, bvh.d_tri_order, n_tris, desc->n_materials, slot, d_pos, (float4 *)s->d_tri_attr, ctl);
    S_HIP(hipGetLastError());
    MiptNode root;
    uint32_t root_slot = 0;
    S_HIP(hipMemcpyAsync(&hctl, ctl, sizeof hctl, hipMemcpyDeviceToHost, st));
    S_HIP(hipMemcpyAsync(&root, bvh.d_nodes, sizeof root, hipMemcpyDeviceToHost, st));
    S_HIP(hipStreamSynchronize(st));
    if (hctl.bad_tri != 0xffffffffu) {
        cleanup();
        return fail(MIPT_ERR_INVALID_ARG, host_nodes ? "triangle " + std::to_string(hctl.bad_tri) + " has material_id >= n_materials " + std::to_string(desc->n_materials)
                                                     : "a triangle has material_id >= n_materials " + std::to_string(desc->n_materials) + " (position " + std::to_string(hctl.bad_tri) + " of the BVH order)");
    }
    if (root.num_tris > 0u) S_HIP(hipMemcpy(&root_slot, slot + root.first_tri_or_child, 4, hipMemcpyDeviceToHost));
    else if (root.first_tri_or_child != 1u) { cleanup(); return fail(MIPT_ERR_BVH, "root's children must be nodes 1 and 2 (bvh.rs:121)"); }
    (void)hipFree(arena); arena = nullptr;
    (void)hipFree(d_tris); d_tris = nullptr;
    const double t_layout = now_ms();

    s->max_leaf = hctl.max_leaf;
    s->n_nodes = n_nodes;
    if (host_nodes) { (void)hipFree(bvh.d_nodes); bvh.d_nodes = nullptr; }          // the caller has them
    else {                                                                           // kept for mipt_scene_get_bvh
        s->d_nodes = bvh.d_nodes; bvh.d_nodes = nullptr;
        s->d_tri_order = bvh.d_tri_order; bvh.d_tri_order = nullptr;
    }
    {   // materials + the texel pool; the textures go through the same pinned ring, straight from the caller's buffers
        int rc = mipt::upload_material_tables(s, tables);
        for (uint32_t i = 0; i < desc->n_textures && rc == MIPT_OK; i++)
            rc = up_ring.copy((uint32_t *)s->d_texels + tables.tex_offset[i], desc->textures[i].rgba8, (size_t)desc->textures[i].width * desc->textures[i].height * 4);
        if (rc == MIPT_OK) rc = up_ring.finish();
        up_ring.shut();
        if (rc) { cleanup(); return rc; }
    }
    { const int rc = mipt::scene_finish_workspace(s); if (rc) { cleanup(); return rc; } }
    s->dev.pairs = d_pairs;
    s->dev.tri_pos = d_pos;
    s->dev.tri_off_bytes = (uint32_t)pairs_bytes;
    s->dev.geom_bytes = (uint32_t)(pairs_bytes + pos_bytes);
    s->dev.tiny_axes = hctl.tiny_axes;
    s->dev.tri_attr = (const float4 *)s->d_tri_attr;
    s->dev.n_pairs = n_records_padded; s->dev.n_tris = n_tris; s->dev.n_mats = desc->n_materials; s->dev.n_texs = desc->n_textures;
    s->dev.root_a = root.num_tris > 0u ? root_slot : 0u;
    s->dev.root_n = root.num_tris;
    const double t_end = now_ms();
    s->info.n_tris = n_tris; s->info.n_nodes = n_nodes; s->info.n_pair_records = n_records_padded; s->info.max_leaf = hctl.max_leaf;
    s->info.geometry_bytes = (uint64_t)pairs_bytes + pos_bytes + s->attr_bytes;
    s->info.built_on_device = host_nodes ? 0u : 1u;
    s->info.upload_ms = (t_up - t_begin) + (t_end - t_layout);
    s->info.build_ms = host_nodes ? 0.0 : bvh.build_ms;
    s->info.layout_ms = t_layout - t_build;
    s->info.total_ms = t_end - t_begin;
    *out = s;
    s = nullptr;
    return MIPT_OK;
}

int mipt::scene_create_from_triangles(const MiptSceneDesc *desc, int device_id, MiptScene **out) { return create_on_device(desc, device_id, false, out); }
int mipt::scene_create_from_nodes(const MiptSceneDesc *desc, int device_id, MiptScene **out) { return create_on_device(desc, device_id, true, out); }

extern "C" int mipt_scene_create_from_triangles(const MiptSceneDesc *desc, int device_id, MiptScene **out) {
    try { return mipt::scene_create_from_triangles(desc, device_id, out); }
    catch (const std::bad_alloc &) { return fail(MIPT_ERR_INVALID_ARG, "out of host memory"); }
    catch (const std::exception &e) { return fail(MIPT_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); }
}

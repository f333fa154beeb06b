// host_layout.cpp -- part of libmipt_diag.so (test infrastructure): the HOST layout of a scene's geometry, byte for byte the buffers
// mipt_scene_create built on host threads in rounds 1-3 (pair records in mipt::pair_order's order with re-based child references |
// the 64-byte intersection stream in mipt::tri_slots' order, and the 64-byte attribute stream).  The product now produces these
// buffers with GPU kernels for both of its entries (rust_ray_tracing_amd/csrc/scene_device.hip); this independent restatement is what
// tests/test_gpu_scene_device.py, test_gpu_fullsize.py and tests/tools/soak_scene_device.py compare the device's bytes with.
// The caller's node array must be a well-formed tree (mipt_scene_create's checks); triangles in the tree's order.
#include "../../include/mipt_diag.h"
#include "../../rust_ray_tracing_amd/csrc/mipt_internal.h"
#include "../../rust_ray_tracing_amd/csrc/pt_kernel.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {
struct F4 { float x, y, z, w; };
inline F4 mk4(float x, float y, float z, float w) { return F4{x, y, z, w}; }
template <class T> struct HostBuf {
    T *p = nullptr;
    size_t n = 0;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { free(p); }
    bool alloc(size_t count, bool zeroed) {
        free(p);
        n = count;
        p = (T *)(zeroed ? calloc(count ? count : 1, sizeof(T)) : malloc((count ? count : 1) * sizeof(T)));
        return p != nullptr;
    }
    void swap(HostBuf &o) { T *tp = p; p = o.p; o.p = tp; size_t tn = n; n = o.n; o.n = tn; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *data() { return p; }
    size_t size() const { return n; }
};
template <class F> void parallel_for(size_t n, F body) {                 // body(begin, end) on disjoint ranges; results must not depend on the split
    unsigned t = std::thread::hardware_concurrency();
    if (t > 16u) t = 16u;
    if (t < 2u || n < (size_t)1 << 16) { body((size_t)0, n); return; }
    const size_t per = (n + t - 1) / t;
    std::vector<std::thread> th;
    for (size_t b = per; b < n; b += per) { const size_t e = b + per < n ? b + per : n; th.emplace_back([&body, b, e]() { body(b, e); }); }
    body((size_t)0, per < n ? per : n);
    for (auto &x : th) x.join();
}

int host_layout(const MiptSceneDesc *desc, uint8_t *geom_out, uint64_t geom_cap, uint8_t *attr_out, uint64_t attr_cap, uint64_t *sizes_out, uint32_t *info_out) {
    if (!desc || !desc->tris || !desc->nodes || !sizes_out || (desc->n_nodes & 1u) == 0u) return MIPT_ERR_INVALID_ARG;
    const uint32_t n_pairs = (desc->n_nodes - 1u) / 2u;
    uint32_t max_leaf = 0;
    for (uint32_t i = 0; i < desc->n_nodes; i++) if (desc->nodes[i].num_tris > max_leaf) max_leaf = desc->nodes[i].num_tris;
    // ---- slots of the intersection stream (mipt::tri_slots, layout_order.cpp): where triangle i's 64-B record sits ----
    std::vector<uint32_t> slot_of_tri(desc->n_tris);
    uint32_t n_slots = 0;
    if (mipt::tri_slots(desc->nodes, desc->n_nodes, desc->n_tris, slot_of_tri.data(), &n_slots) != MIPT_OK)
        return MIPT_ERR_BVH;
    HostBuf<F4> pairs;                                                  // (+4: room for the pad record below)
    if (!pairs.alloc((size_t)n_pairs * 4 + 4, false)) return MIPT_ERR_INVALID_ARG;
    pairs.n = (size_t)n_pairs * 4;
    for (int q = 0; q < 4; q++) pairs[(size_t)n_pairs * 4 + q] = mk4(0, 0, 0, 0);
    parallel_for(n_pairs, [&](size_t kb, size_t ke) {
        for (size_t k = kb; k < ke; k++) {
            for (uint32_t w = 0; w < 2; w++) {
                const MiptNode &n = desc->nodes[2 * k + 1 + w];
                const uint32_t a = n.num_tris > 0 ? slot_of_tri[n.first_tri_or_child] : (n.first_tri_or_child - 1u) / 2u;
                F4 lo, hi;
                lo.x = n.bounds_min.x; lo.y = n.bounds_min.y; lo.z = n.bounds_min.z; memcpy(&lo.w, &a, 4);
                hi.x = n.bounds_max.x; hi.y = n.bounds_max.y; hi.z = n.bounds_max.z; memcpy(&hi.w, &n.num_tris, 4);
                pairs[k * 4 + w * 2 + 0] = lo;
                pairs[k * 4 + w * 2 + 1] = hi;
            }
        }
    });
    // ---- order of the pair records in HBM (mipt::pair_order, layout_order.cpp): the tree top breadth-first, below it every
    // pair in one 128-B line with the child pair of its larger inner child.  Topology, visit order and results are untouched; only
    // `a` of the inner children is renumbered.
    std::vector<uint32_t> new_of(n_pairs);                                  // reference pair index -> record index in HBM
    for (uint32_t k = 0; k < n_pairs; k++) new_of[k] = k;
    if (n_pairs > 0) {
        std::vector<uint32_t> order(2 * (size_t)n_pairs + 2);               // record index -> reference pair index (0xffffffff = pad; at most one pad per level)
        uint32_t n_records = 0;
        {
            const uint32_t cap = order.size() < (size_t)mipt::kMaxPairs ? (uint32_t)order.size() : mipt::kMaxPairs;
            const int rc = mipt::pair_order(desc->nodes, desc->n_nodes, order.data(), cap, &n_records);
            if (rc != MIPT_OK) return rc;
        }
        order.resize(n_records);
        auto child_pair = [&](uint32_t k, uint32_t w, uint32_t *out) -> bool {
            const MiptNode &n = desc->nodes[2 * k + 1 + w];
            if (n.num_tris != 0u) return false;
            *out = (n.first_tri_or_child - 1u) / 2u;
            return true;
        };
        parallel_for(order.size(), [&](size_t jb, size_t je) {
            for (size_t j = jb; j < je; j++) if (order[j] != 0xffffffffu) new_of[order[j]] = (uint32_t)j;      // every pair appears once: disjoint writes
        });
        HostBuf<F4> re;                                                 // pad records stay zero (calloc)
        if (!re.alloc(order.size() * 4 + 4, true)) return MIPT_ERR_INVALID_ARG;
        re.n = order.size() * 4;
        parallel_for(order.size(), [&](size_t jb, size_t je) {
            for (size_t j = jb; j < je; j++) {
                if (order[j] == 0xffffffffu) continue;
                for (int q = 0; q < 4; q++) re[j * 4 + q] = pairs[(size_t)order[j] * 4 + q];
                for (uint32_t w = 0; w < 2; w++) {
                    uint32_t c;
                    if (child_pair(order[j], w, &c)) memcpy(&re[j * 4 + w * 2].w, &new_of[c], 4);
                }
            }
        });
        pairs.swap(re);
    }
    if ((pairs.size() / 4) & 1u) pairs.n += 4;                             // one zero pad record (allocated above): the triangle stream behind it starts on a 128-B line
    const uint32_t n_pair_records = (uint32_t)(pairs.size() / 4);
    // ---- triangles: 64-B-strided intersection stream + 64-B shading stream ----
    if (n_slots > mipt::kMaxTris) return MIPT_ERR_SCENE_LIMIT;
    HostBuf<F4> tri_pos, tri_attr;                                      // tri_pos: +1 for the kernel's unconditional 4th F4 load; unused slots / words stay zero
    if (!tri_pos.alloc((size_t)n_slots * mipt::kTriPosStride / 16 + 1, true) || !tri_attr.alloc((size_t)desc->n_tris * 4, false))
        return MIPT_ERR_INVALID_ARG;
    std::atomic<uint32_t> bad_tri{UINT32_MAX};
    parallel_for(desc->n_tris, [&](size_t ib, size_t ie) {
        for (size_t ii = ib; ii < ie; ii++) {
            const uint32_t i = (uint32_t)ii;
            const MiptTriangle &t = desc->tris[i];
            if (t.material_id >= desc->n_materials) {                      // reported below: the lowest such triangle, as a sequential scan would
                uint32_t cur = bad_tri.load();
                while (i < cur && !bad_tri.compare_exchange_weak(cur, i)) {}
                continue;
            }
            const MiptVec3 v0 = t.vertices[0].position, v1 = t.vertices[1].position, v2 = t.vertices[2].position;
            // edge_1 = v_2 - v_1, edge_2 = v_3 - v_1 (ray.rs:24-25): one rounded f32 subtraction each,
            // the same value the reference recomputes per test (-ffp-contract=off; no fusing possible here).
            const float e1x = v1.x - v0.x, e1y = v1.y - v0.y, e1z = v1.z - v0.z;
            const float e2x = v2.x - v0.x, e2y = v2.y - v0.y, e2z = v2.z - v0.z;
            const size_t q = (size_t)slot_of_tri[i] * (mipt::kTriPosStride / 16);
            float idf;
            memcpy(&idf, &i, 4);
            tri_pos[q + 0] = mk4(v0.x, v0.y, v0.z, e1x);
            tri_pos[q + 1] = mk4(e1y, e1z, e2x, e2y);
            tri_pos[q + 2] = mk4(e2z, idf, 0.0f, 0.0f);
            const MiptVec3 n0 = t.vertices[0].normal, n1 = t.vertices[1].normal, n2 = t.vertices[2].normal;
            float mid;
            memcpy(&mid, &t.material_id, 4);
            tri_attr[(size_t)i * 4 + 0] = mk4(n0.x, n0.y, n0.z, n1.x);
            tri_attr[(size_t)i * 4 + 1] = mk4(n1.y, n1.z, n2.x, n2.y);
            tri_attr[(size_t)i * 4 + 2] = mk4(n2.z, t.vertices[0].tex_coord_x, t.vertices[0].tex_coord_y, t.vertices[1].tex_coord_x);
            tri_attr[(size_t)i * 4 + 3] = mk4(t.vertices[1].tex_coord_y, t.vertices[2].tex_coord_x, t.vertices[2].tex_coord_y, mid);
        }
    });
    if (bad_tri.load() != UINT32_MAX)
        return MIPT_ERR_INVALID_ARG;
    const size_t pairs_bytes = pairs.size() * sizeof(F4), pos_bytes = tri_pos.size() * sizeof(F4), attr_bytes = tri_attr.size() * sizeof(F4);
    sizes_out[0] = pairs_bytes + pos_bytes;
    sizes_out[1] = attr_bytes;
    if (info_out) {
        info_out[0] = n_pair_records;
        info_out[1] = max_leaf;
        info_out[2] = desc->nodes[0].num_tris > 0 ? slot_of_tri[desc->nodes[0].first_tri_or_child] : 0u;   // root_a
        info_out[3] = desc->nodes[0].num_tris;                                                              // root_n
    }
    if (geom_out) {
        if (geom_cap < pairs_bytes + pos_bytes) return MIPT_ERR_INVALID_ARG;
        memcpy(geom_out, pairs.data(), pairs_bytes);
        memcpy(geom_out + pairs_bytes, tri_pos.data(), pos_bytes);
    }
    if (attr_out) {
        if (attr_cap < attr_bytes) return MIPT_ERR_INVALID_ARG;
        memcpy(attr_out, tri_attr.data(), attr_bytes);
    }
    return MIPT_OK;
}
} // namespace

// the fingerprint of tests/cpp/scene_hooks.hip's hash_words, on the host: sum (w_i + c) * (2 i + 1) mod 2^64
extern "C" int mipt_diag_hash_words(const void *words, uint64_t n_words, uint64_t *out) {
    if (!words || !out) return MIPT_ERR_INVALID_ARG;
    const uint32_t *w = (const uint32_t *)words;
    std::atomic<uint64_t> acc{0};
    try {
        parallel_for((size_t)n_words, [&](size_t b, size_t e) {
            uint64_t a = 0;
            for (size_t i = b; i < e; i++) a += ((uint64_t)w[i] + 0x9E3779B97F4A7C15ull) * (2ull * (uint64_t)i + 1ull);
            acc.fetch_add(a);
        });
    } catch (...) { return MIPT_ERR_INVALID_ARG; }
    *out = acc.load();
    return MIPT_OK;
}

extern "C" int mipt_diag_host_layout(const void *desc_, uint8_t *geom_out, uint64_t geom_cap, uint8_t *attr_out, uint64_t attr_cap,
                                     uint64_t *sizes_out, uint32_t *info_out) {
    const MiptSceneDesc *desc = (const MiptSceneDesc *)desc_;
    try { return host_layout(desc, geom_out, geom_cap, attr_out, attr_cap, sizes_out, info_out); }
    catch (...) { return MIPT_ERR_INVALID_ARG; }
}

// bvh_build.cpp -- host-side restatement of BVH::build (reference src/bvh.rs:13-161) that emits
// the IDENTICAL node array and triangle order, but in O(2 passes) per node instead of the
// reference's ~27 and with independent subtrees built on separate threads.
//
// Why the output is identical (each point is checked against the CPU oracle in tests/):
//  * A triangle's centroid (Triangle::bounds_mid, scene.rs:114-126) and AABB depend only on the
//    triangle, so they are computed once with the same f32 operations and cached.
//  * evaluate_sah (bvh.rs:138-161) grows two boxes with f32 min/max and counts triangles on each
//    side of `centroid < split_pos`.  min/max are exact and order-independent, and the 7 candidate
//    planes cmin + i*scale are non-decreasing in i, so the set left(i) = {c < pos_i} is exactly
//    the union of "first plane the centroid is below" bins 1..i.  The bins are filled with the
//    reference's own comparisons against the reference's own plane values (no floor/divide
//    binning), then prefix/suffix-merged; the cost expression is evaluated with the same f32
//    operations, including 0*inf = NaN -> f32::MAX for an empty side (bvh.rs:153-160).
//  * The partition loop (bvh.rs:99-108) is replayed verbatim on 40-byte proxy records; the fat
//    112-byte triangles are permuted once at the end.
//  * Node indices: split_node pushes both children at the current end of the vector and then
//    recurses left, right (bvh.rs:131-135), i.e. desc(X) = [A, B] ++ desc(A) ++ desc(B).  Subtrees
//    are built into relative-indexed blocks and stitched in that order.
#include "../../include/mipt.h"
#include "mipt_internal.h"

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <exception>
#include <thread>
#include <vector>

namespace {

constexpr float F32_MAX = FLT_MAX;

struct Proxy {              // 40 B
    float c[3];             // Triangle::bounds_mid
    float lo[3], hi[3];     // vertex AABB
    uint32_t idx;           // original triangle index
};

struct Box {
    float lo[3], hi[3];
    void reset() { for (int i = 0; i < 3; i++) { lo[i] = F32_MAX; hi[i] = -F32_MAX; } }   // Node::default, bvh.rs:173-182
    void grow(const Proxy &p) { for (int i = 0; i < 3; i++) { lo[i] = fminf(lo[i], p.lo[i]); hi[i] = fmaxf(hi[i], p.hi[i]); } }
    void grow(const Box &b) { for (int i = 0; i < 3; i++) { lo[i] = fminf(lo[i], b.lo[i]); hi[i] = fmaxf(hi[i], b.hi[i]); } }
    float area() const {    // Node::surface_area (half area), bvh.rs:196-203
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return (ex * ez) + (ex * ey) + (ez * ey);
    }
};

inline MiptNode make_node(const Box &b, uint32_t first, uint32_t n) {
    MiptNode nd;
    nd.bounds_min = {b.lo[0], b.lo[1], b.lo[2]}; nd.first_tri_or_child = first;
    nd.bounds_max = {b.hi[0], b.hi[1], b.hi[2]}; nd.num_tris = n;
    return nd;
}
inline Box node_box(const MiptNode &n) {
    Box b;
    b.lo[0] = n.bounds_min.x; b.lo[1] = n.bounds_min.y; b.lo[2] = n.bounds_min.z;
    b.hi[0] = n.bounds_max.x; b.hi[1] = n.bounds_max.y; b.hi[2] = n.bounds_max.z;
    return b;
}

struct Bins { Box box[3][8]; uint32_t cnt[3][8]; };

struct Split { bool ok; int axis; float pos; uint32_t a_count; };

struct Builder {
    Proxy *px;
    std::atomic<int> spare_threads{0};
    std::atomic<int> err{0};
};

// Chooses the split for proxies [first, first+n) (bvh.rs:58-97) and partitions them (bvh.rs:99-113).
Split split_range(Builder &B, uint32_t first, uint32_t n, const Box &bounds) {
    Proxy *px = B.px;
    Split s{false, 0, 0.0f, 0};
    const float parent_cost = (float)n * bounds.area();
    // pass 1: centroid ranges for all three axes (bvh.rs:68-77; f32::MIN == -f32::MAX)
    float cmin[3] = {F32_MAX, F32_MAX, F32_MAX}, cmax[3] = {-F32_MAX, -F32_MAX, -F32_MAX};
    for (uint32_t i = 0; i < n; i++) {
        const Proxy &p = px[first + i];
        for (int a = 0; a < 3; a++) { cmin[a] = fminf(cmin[a], p.c[a]); cmax[a] = fmaxf(cmax[a], p.c[a]); }
    }
    float pos[3][8];
    bool use[3];
    for (int a = 0; a < 3; a++) {
        use[a] = !(cmin[a] == cmax[a]);                                  // bvh.rs:78
        const float scale = (cmax[a] - cmin[a]) / 8.0f;                  // bvh.rs:82
        for (int i = 1; i < 8; i++) pos[a][i] = cmin[a] + (float)i * scale;   // bvh.rs:84
    }
    // pass 2: bin k = first plane i (1..7) with c < pos_i, else 8 (stored at k-1)
    std::unique_ptr<Bins> bins(new Bins);
    for (int a = 0; a < 3; a++)
        for (int k = 0; k < 8; k++) { bins->box[a][k].reset(); bins->cnt[a][k] = 0; }
    for (uint32_t i = 0; i < n; i++) {
        const Proxy &p = px[first + i];
        for (int a = 0; a < 3; a++) {
            if (!use[a]) continue;
            int k = 8;
            for (int j = 1; j < 8; j++) if (p.c[a] < pos[a][j]) { k = j; break; }
            bins->box[a][k - 1].grow(p);
            bins->cnt[a][k - 1] += 1;
        }
    }
    float best_cost = F32_MAX, best_pos = 0.0f;
    int best_axis = 0;
    for (int a = 0; a < 3; a++) {
        if (!use[a]) continue;
        Box right[8]; uint32_t rcnt[8];
        Box acc; acc.reset(); uint32_t c = 0;
        for (int k = 7; k >= 0; k--) { acc.grow(bins->box[a][k]); c += bins->cnt[a][k]; right[k] = acc; rcnt[k] = c; }
        Box left; left.reset(); uint32_t lc = 0;
        for (int i = 1; i < 8; i++) {
            left.grow(bins->box[a][i - 1]); lc += bins->cnt[a][i - 1];
            // evaluate_sah (bvh.rs:150-160): right(i) = bins i+1..8 = stored indices i..7
            const float cost = (float)lc * left.area() + (float)rcnt[i] * right[i].area();
            const float split_cost = (cost > 0.0f) ? cost : F32_MAX;
            if (split_cost < best_cost) { best_axis = a; best_pos = pos[a][i]; best_cost = split_cost; }   // bvh.rs:86-90
        }
    }
    if (best_cost >= parent_cost) return s;                              // bvh.rs:94
    // partition (bvh.rs:99-108), verbatim on the proxies
    uint32_t i = first, j = first + n - 1;
    while (i <= j) {
        if (px[i].c[best_axis] < best_pos) { i += 1; }
        else {
            Proxy t = px[i]; px[i] = px[j]; px[j] = t;
            if (j == 0) { B.err.store(-1); return s; }                  // Rust would panic on the u32 underflow
            j -= 1;
        }
    }
    const uint32_t a_count = i - first;
    if (a_count == 0 || a_count == n) return s;                          // bvh.rs:110-113
    s.ok = true; s.axis = best_axis; s.pos = best_pos; s.a_count = a_count;
    return s;
}

// desc(X) for the node covering [first, first+n) into `out`, child indices relative to out[0].
void build_block(Builder &B, uint32_t first, uint32_t n, const Box &bounds, std::vector<MiptNode> &out) {
    struct Item { uint32_t first, n; Box box; int64_t self; };   // self: index in `out` of this node, -1 = block root
    std::vector<Item> todo;
    todo.push_back({first, n, bounds, -1});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const Split s = split_range(B, it.first, it.n, it.box);
        if (!s.ok) continue;
        Box a, b; a.reset(); b.reset();                                   // bvh.rs:115-129
        for (uint32_t k = 0; k < s.a_count; k++) a.grow(B.px[it.first + k]);
        for (uint32_t k = s.a_count; k < it.n; k++) b.grow(B.px[it.first + k]);
        const uint32_t used = (uint32_t)out.size();
        if (it.self >= 0) { out[(size_t)it.self].first_tri_or_child = used; out[(size_t)it.self].num_tris = 0; }
        out.push_back(make_node(a, it.first, s.a_count));
        out.push_back(make_node(b, it.first + s.a_count, it.n - s.a_count));
        // left subtree is completed before the right child allocates (bvh.rs:134-135): LIFO, right first
        todo.push_back({it.first + s.a_count, it.n - s.a_count, b, (int64_t)used + 1});
        todo.push_back({it.first, s.a_count, a, (int64_t)used});
    }
}

struct Sub {                          // a subtree built as its own task
    bool split = false;               // true: node has children A, B handled as tasks
    MiptNode a{}, b{};
    std::unique_ptr<Sub> sa, sb;
    std::vector<MiptNode> block;      // !split: desc(X), relative indices (may be empty = leaf)
    size_t size = 0;                  // |desc(X)|
};

constexpr uint32_t kTaskMinTris = 1u << 15;

void build_task(Builder &B, uint32_t first, uint32_t n, const Box &bounds, Sub &sub) {
    if (n < kTaskMinTris) {
        build_block(B, first, n, bounds, sub.block);
        sub.size = sub.block.size();
        return;
    }
    const Split s = split_range(B, first, n, bounds);
    if (!s.ok) { sub.size = 0; return; }
    Box a, b; a.reset(); b.reset();
    for (uint32_t k = 0; k < s.a_count; k++) a.grow(B.px[first + k]);
    for (uint32_t k = s.a_count; k < n; k++) b.grow(B.px[first + k]);
    sub.split = true;
    sub.a = make_node(a, first, s.a_count);
    sub.b = make_node(b, first + s.a_count, n - s.a_count);
    sub.sa.reset(new Sub); sub.sb.reset(new Sub);
    bool forked = false;
    std::thread th;
    if (B.spare_threads.fetch_sub(1) > 0) {
        forked = true;
        th = std::thread([&] { build_task(B, first, s.a_count, a, *sub.sa); });
    } else {
        B.spare_threads.fetch_add(1);
    }
    if (!forked) build_task(B, first, s.a_count, a, *sub.sa);
    build_task(B, first + s.a_count, n - s.a_count, b, *sub.sb);
    if (forked) { th.join(); B.spare_threads.fetch_add(1); }
    sub.size = 2 + sub.sa->size + sub.sb->size;
}

// writes desc(X) at nodes[off ...]; `self` is X's own slot (its first_tri_or_child is set here)
void place(const Sub &sub, MiptNode *nodes, uint32_t off, MiptNode &self) {
    if (sub.size == 0) return;                                            // X stays a leaf
    self.first_tri_or_child = off; self.num_tris = 0;
    if (!sub.split) {
        for (size_t i = 0; i < sub.block.size(); i++) {
            MiptNode nd = sub.block[i];
            if (nd.num_tris == 0) nd.first_tri_or_child += off;
            nodes[off + i] = nd;
        }
        return;
    }
    nodes[off] = sub.a; nodes[off + 1] = sub.b;
    place(*sub.sa, nodes, off + 2, nodes[off]);
    place(*sub.sb, nodes, off + 2 + (uint32_t)sub.sa->size, nodes[off + 1]);
}

} // namespace

static int bvh_build_impl(MiptTriangle *tris, uint32_t n_tris, MiptNode *nodes_out, uint32_t nodes_cap,
                          uint32_t *n_nodes_out, uint32_t threads) {
    if (!tris || !nodes_out || n_tris == 0 || nodes_cap == 0) return MIPT_ERR_INVALID_ARG;   // empty scene: the reference panics
    std::vector<Proxy> px(n_tris);
    Box root; root.reset();
    for (uint32_t i = 0; i < n_tris; i++) {
        Proxy &p = px[i];
        p.idx = i;
        for (int a = 0; a < 3; a++) {
            float mn = F32_MAX, mx = -F32_MAX;                            // scene.rs:115-123, bvh.rs:185-194
            for (int v = 0; v < 3; v++) {
                const float q = (&tris[i].vertices[v].position.x)[a];
                mn = fminf(mn, q); mx = fmaxf(mx, q);
            }
            p.lo[a] = mn; p.hi[a] = mx;
            p.c[a] = (mn + mx) / 2.0f;                                    // scene.rs:125
        }
        root.grow(p);                                                     // bvh.rs:20-22
    }
    Builder B;
    B.px = px.data();
    unsigned hw = threads ? threads : std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    B.spare_threads.store((int)hw - 1);
    Sub top;
    build_task(B, 0, n_tris, root, top);
    if (B.err.load()) return MIPT_ERR_BVH;
    const size_t total = 1 + top.size;
    if (total > nodes_cap) return MIPT_ERR_INVALID_ARG;
    nodes_out[0] = make_node(root, 0, n_tris);                            // bvh.rs:23-24
    place(top, nodes_out, 1, nodes_out[0]);
    if (n_nodes_out) *n_nodes_out = (uint32_t)total;
    // apply the permutation to the fat triangles (the reference swaps them in place, bvh.rs:105)
    std::vector<MiptTriangle> tmp(tris, tris + n_tris);
    for (uint32_t i = 0; i < n_tris; i++) tris[i] = tmp[px[i].idx];
    return MIPT_OK;
}

// No C++ exception may cross the C ABI (allocation or thread-creation failure on the calling thread -> status code).
extern "C" int mipt_bvh_build(MiptTriangle *tris, uint32_t n_tris, MiptNode *nodes_out, uint32_t nodes_cap,
                              uint32_t *n_nodes_out, uint32_t threads) {
    try {
        return bvh_build_impl(tris, n_tris, nodes_out, nodes_cap, n_nodes_out, threads);
    } catch (const std::exception &) {
        return MIPT_ERR_INVALID_ARG;
    }
}

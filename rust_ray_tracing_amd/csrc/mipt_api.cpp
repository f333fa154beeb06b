// mipt_api.cpp -- the extern "C" boundary of libmipt.so (include/mipt.h): scene upload into the
// HBM layout of pt_kernel.h, render dispatch, status / error plumbing.  Host C++ only; every
// device operation is stream-ordered HIP.  There is deliberately no CPU rendering path here:
// without a HIP device the entry points return MIPT_ERR_HIP.
#include "../../include/mipt.h"
#include "pt_kernel.h"
#include "mipt_internal.h"
#include "mipt_scene.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

static_assert(sizeof(MiptVec3) == 12, "Vec3f");
static_assert(sizeof(MiptVertex) == 32, "Vertex");
static_assert(sizeof(MiptTriangle) == 112 && offsetof(MiptTriangle, material_id) == 96, "Triangle");
static_assert(sizeof(MiptNode) == 32 && offsetof(MiptNode, first_tri_or_child) == 12 &&
                  offsetof(MiptNode, bounds_max) == 16 && offsetof(MiptNode, num_tris) == 28, "Node");
static_assert(sizeof(MiptMaterial) == 80 && offsetof(MiptMaterial, ior) == 28 && offsetof(MiptMaterial, emission) == 32 &&
                  offsetof(MiptMaterial, roughness) == 44 && offsetof(MiptMaterial, base_color_tex_id) == 56, "Material");
static_assert(sizeof(MiptCamera) == 80 && offsetof(MiptCamera, position) == 64, "UniformCamera");
static_assert(sizeof(MiptStats) == 8 + 22 * 8 && sizeof(MiptOptions) == 64, "ABI struct sizes (tests/test_abi.py, rust_ray_tracing_amd/_lib.py)");
static_assert(sizeof(MiptSceneInfo) == 64, "MiptSceneInfo");
static_assert(sizeof(mipt::DevMaterial) == 64 && sizeof(mipt::DevMaterialFull) == 128, "device records");

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(MIPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));          \
    } while (0)

} // namespace

void mipt_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }

void mipt::free_scene(MiptScene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    void *ptrs[] = {s->d_geom, s->d_tri_attr, s->d_mats, s->d_mats_full, s->d_texels, s->d_nodes, s->d_tri_order,
                    s->d_stats, s->d_ovf, s->d_hdr, s->d_rgba, s->d_touched};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

namespace {

using mipt::free_scene;

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T>
int upload(void **dst, const T *src, size_t count, size_t min_bytes = 16) {
    size_t bytes = count * sizeof(T);
    size_t alloc = bytes < min_bytes ? min_bytes : bytes;
    HIP_TRY(hipMalloc(dst, alloc));
    if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return MIPT_OK;
}
template <class T>
int upload(void **dst, const std::vector<T> &src, size_t min_bytes = 16) { return upload(dst, src.data(), src.size(), min_bytes); }

int ensure(void **p, size_t *have, size_t want_bytes) {
    if (*have >= want_bytes && *p) return MIPT_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    HIP_TRY(hipMalloc(p, want_bytes));
    *have = want_bytes;
    return MIPT_OK;
}

// host loops over the caller's arrays (the material-id check) on up to 16 threads
template <class F> void parallel_for(size_t n, F body) {                 // body(begin, end) on disjoint ranges; results must not depend on the split
    unsigned t = std::thread::hardware_concurrency();
    if (t > 16u) t = 16u;
    if (t < 2u || n < (size_t)1 << 16) { body((size_t)0, n); return; }
    const size_t per = (n + t - 1) / t;
    std::vector<std::thread> th;
    size_t b = per;                                                       // [0, per) runs on the calling thread
    try {
        for (; b < n; b += per) { const size_t e = b + per < n ? b + per : n; th.emplace_back([&body, b, e]() { body(b, e); }); }
    } catch (...) {                                                       // could not start a thread: the caller does the rest
        body((size_t)0, per < n ? per : n);
        for (auto &x : th) x.join();
        body(b, n);
        return;
    }
    body((size_t)0, per < n ? per : n);
    for (auto &x : th) x.join();
}

} // namespace

// ---- pieces shared with the device-resident setup (scene_device.hip) ----
int mipt::build_material_tables(const MiptSceneDesc *desc, MaterialTables *out, bool gather_texels) {
    struct TexDesc { uint32_t offset, width, height; };
    if (!desc->materials || desc->n_materials == 0) return fail(MIPT_ERR_INVALID_ARG, "scene has no materials");
    if (desc->n_textures && !desc->textures) return fail(MIPT_ERR_INVALID_ARG, "n_textures > 0 but textures == NULL");
    std::vector<TexDesc> texs(desc->n_textures);
    uint64_t n_texels = 0;
    for (uint32_t i = 0; i < desc->n_textures; i++) {
        const MiptTexture &t = desc->textures[i];
        if (!t.rgba8 || t.width == 0 || t.height == 0) return fail(MIPT_ERR_INVALID_ARG, "texture %u is empty", i);
        if (n_texels + (uint64_t)t.width * t.height > 0xffffffffull) return fail(MIPT_ERR_SCENE_LIMIT, "texture pool exceeds 2^32 texels");
        texs[i] = {(uint32_t)n_texels, t.width, t.height};
        n_texels += (uint64_t)t.width * t.height;
    }
    out->mats.assign(desc->n_materials, mipt::DevMaterial{});
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        const MiptMaterial &m = desc->materials[i];
        if ((m.base_color_tex_id != UINT32_MAX && m.base_color_tex_id >= desc->n_textures) ||
            (m.emission_tex_id != UINT32_MAX && m.emission_tex_id >= desc->n_textures))
            return fail(MIPT_ERR_INVALID_ARG, "material %u references a texture >= n_textures %u", i, desc->n_textures);
        mipt::DevMaterial d{};
        d.base[0] = m.base_color.x; d.base[1] = m.base_color.y; d.base[2] = m.base_color.z;
        d.emis[0] = m.emission.x; d.emis[1] = m.emission.y; d.emis[2] = m.emission.z;
        if (m.base_color_tex_id != UINT32_MAX) { const TexDesc &t = texs[m.base_color_tex_id]; d.base_off = t.offset; d.base_w = t.width; d.base_h = t.height; }
        if (m.emission_tex_id != UINT32_MAX) { const TexDesc &t = texs[m.emission_tex_id]; d.emis_off = t.offset; d.emis_w = t.width; d.emis_h = t.height; }
        out->mats[i] = d;
    }
    out->mats_full.assign(desc->n_materials, mipt::DevMaterialFull{});
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        const MiptMaterial &m = desc->materials[i];
        mipt::DevMaterialFull f{};
        f.base[0] = m.base_color.x; f.base[1] = m.base_color.y; f.base[2] = m.base_color.z; f.transmission = m.transmission;
        f.emission[0] = m.emission.x; f.emission[1] = m.emission.y; f.emission[2] = m.emission.z; f.ior = m.ior;
        f.roughness = m.roughness; f.metallic = m.metallic; f.transparency = m.transparency;
        const uint32_t ids[6] = {m.base_color_tex_id, m.transparency_tex_id, m.roughness_tex_id, m.metallic_tex_id, m.emission_tex_id, m.normal_tex_id};
        for (int k = 0; k < 6; k++) {
            if (ids[k] == UINT32_MAX) continue;
            if (ids[k] >= desc->n_textures) return fail(MIPT_ERR_INVALID_ARG, "material %u references a texture >= n_textures %u", i, desc->n_textures);
            f.tex[k][0] = texs[ids[k]].offset; f.tex[k][1] = texs[ids[k]].width; f.tex[k][2] = texs[ids[k]].height;
        }
        out->mats_full[i] = f;
    }
    out->n_texels = n_texels;
    out->tex_offset.resize(desc->n_textures);
    for (uint32_t i = 0; i < desc->n_textures; i++) out->tex_offset[i] = texs[i].offset;
    out->texels.clear();
    if (gather_texels) {
        out->texels.resize((size_t)n_texels);
        for (uint32_t i = 0; i < desc->n_textures; i++)
            memcpy(out->texels.data() + texs[i].offset, desc->textures[i].rgba8, (size_t)texs[i].width * texs[i].height * 4);
    }
    return MIPT_OK;
}

int mipt::upload_material_tables(MiptScene *s, const MaterialTables &t) {
    int rc;
    if ((rc = upload(&s->d_mats, t.mats, 64)) || (rc = upload(&s->d_mats_full, t.mats_full, 128))) return rc;
    if (!t.texels.empty() || t.n_texels == 0) { if ((rc = upload(&s->d_texels, t.texels, 16))) return rc; }
    else HIP_TRY(hipMalloc(&s->d_texels, (size_t)t.n_texels * 4));              // filled by the caller (scene_device.hip stages the textures itself)
    s->mats_bytes = t.mats.size() * sizeof(mipt::DevMaterial) < 64 ? 64 : t.mats.size() * sizeof(mipt::DevMaterial);
    s->mats_full_bytes = t.mats_full.size() * sizeof(mipt::DevMaterialFull) < 128 ? 128 : t.mats_full.size() * sizeof(mipt::DevMaterialFull);
    s->texel_bytes = (size_t)t.n_texels * 4 < 16 ? 16 : (size_t)t.n_texels * 4;
    s->dev.mats = (const mipt::DevMaterial *)s->d_mats;
    s->dev.mats_full = (const mipt::DevMaterialFull *)s->d_mats_full;
    s->dev.texels = (const uint32_t *)s->d_texels;
    return MIPT_OK;
}

int mipt::scene_finish_workspace(MiptScene *s) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, s->device);
    if (e != hipSuccess) return fail(MIPT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    s->n_cu = prop.multiProcessorCount;
    if ((e = hipMalloc((void **)&s->d_stats, sizeof(mipt::DevStats))) != hipSuccess ||
        (e = hipEventCreate(&s->ev0)) != hipSuccess || (e = hipEventCreate(&s->ev1)) != hipSuccess)
        return fail(MIPT_ERR_HIP, "workspace allocation: %s", hipGetErrorString(e));
    return MIPT_OK;
}

// Replica by device-to-device copy.  hipMemcpyPeer needs no peer mapping; with one enabled (tried, failure ignored) the copy engines
// move the data directly over the xGMI link between the two GPUs instead of staging it.
// Two phases, so that several replicas are in flight at once (mipt::scene_clone_many): `clone_issue` allocates on the destination and
// queues the copies on ITS null stream, `clone_finish` waits for them.  Seven pulls from device 0 use seven different xGMI links.
static int clone_issue(const MiptScene *src, int device, MiptScene **out) {
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    if (device != src->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src->device, 0);
            if (pe != hipSuccess) (void)hipGetLastError();          // hipErrorPeerAccessAlreadyEnabled included
        }
    }
    MiptScene *s = new (std::nothrow) MiptScene();
    if (!s) return fail(MIPT_ERR_INVALID_ARG, "out of host memory");
    s->device = device;
    s->dev = src->dev;
    s->max_leaf = src->max_leaf; s->n_tris = src->n_tris; s->n_nodes = src->n_nodes; s->info = src->info;
    s->geom_alloc = src->geom_alloc; s->attr_bytes = src->attr_bytes; s->mats_bytes = src->mats_bytes;
    s->mats_full_bytes = src->mats_full_bytes; s->texel_bytes = src->texel_bytes;
    struct Part { void **dst; const void *from; size_t alloc, copy; };
    const Part parts[] = {{&s->d_geom, src->d_geom, src->geom_alloc, (size_t)src->dev.geom_bytes},
                          {&s->d_tri_attr, src->d_tri_attr, src->attr_bytes < 16 ? 16 : src->attr_bytes, src->attr_bytes},
                          {&s->d_mats, src->d_mats, src->mats_bytes, src->mats_bytes},
                          {&s->d_mats_full, src->d_mats_full, src->mats_full_bytes, src->mats_full_bytes},
                          {&s->d_texels, src->d_texels, src->texel_bytes, src->texel_bytes},
                          {(void **)&s->d_nodes, src->d_nodes, (size_t)src->n_nodes * sizeof(MiptNode), src->d_nodes ? (size_t)src->n_nodes * sizeof(MiptNode) : 0},
                          {(void **)&s->d_tri_order, src->d_tri_order, src->n_tris * 4, src->d_tri_order ? src->n_tris * 4 : 0}};
    for (const Part &p : parts) {
        if (!p.from) continue;
        hipError_t e = hipMalloc(p.dst, p.alloc ? p.alloc : 16);
        if (e == hipSuccess && p.copy) e = hipMemcpyPeerAsync(*p.dst, device, p.from, src->device, p.copy, nullptr);
        if (e != hipSuccess) { (void)hipStreamSynchronize(nullptr); free_scene(s); return fail(MIPT_ERR_HIP, "replica copy %d -> %d: %s", src->device, device, hipGetErrorString(e)); }
    }
    *out = s;
    return MIPT_OK;
}
static int clone_finish(const MiptScene *src, MiptScene *s, double t0) {
    const int device = s->device;
    HIP_TRY(hipSetDevice(device));
    {
        const hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) return fail(MIPT_ERR_HIP, "replica copy %d -> %d: %s", src->device, device, hipGetErrorString(e));
    }
    const int rc = mipt::scene_finish_workspace(s);
    if (rc) return rc;
    const size_t pairs_bytes = src->dev.tri_off_bytes;
    s->dev.pairs = (const float4 *)s->d_geom;
    s->dev.tri_pos = (const float4 *)((const char *)s->d_geom + pairs_bytes);
    s->dev.tri_attr = (const float4 *)s->d_tri_attr;
    s->dev.mats = (const mipt::DevMaterial *)s->d_mats;
    s->dev.mats_full = (const mipt::DevMaterialFull *)s->d_mats_full;
    s->dev.texels = (const uint32_t *)s->d_texels;
    s->info.replica_of_device = (uint32_t)src->device + 1u;
    s->info.upload_ms = now_ms() - t0; s->info.build_ms = 0.0; s->info.layout_ms = 0.0; s->info.total_ms = s->info.upload_ms;
    return MIPT_OK;
}
int mipt::scene_clone_to(const MiptScene *src, int device, MiptScene **out) {
    const double t0 = now_ms();
    int rc = clone_issue(src, device, out);
    if (rc == MIPT_OK && (rc = clone_finish(src, *out, t0))) { free_scene(*out); *out = nullptr; }
    return rc;
}
// outs[1 .. n) = replicas of `src` (= outs[0]) on device_ids[1 .. n), all copies queued before the first is waited for; a replica's
// upload_ms runs from the common start to ITS completion.  On failure every replica made here is freed and outs[i >= 1] are null.
int mipt::scene_clone_many(const MiptScene *src, const int *device_ids, int n_dev, MiptScene **outs) {
    const double t0 = now_ms();
    int rc = MIPT_OK, fail_dev = -1;
    for (int i = 1; i < n_dev && rc == MIPT_OK; i++) if ((rc = clone_issue(src, device_ids[i], &outs[i]))) fail_dev = device_ids[i];
    for (int i = 1; i < n_dev; i++) {
        if (!outs[i]) continue;
        const int r = clone_finish(src, outs[i], t0);            // also when an earlier one failed: the copies are in flight
        if (r && rc == MIPT_OK) { rc = r; fail_dev = device_ids[i]; }
    }
    if (rc) {
        const std::string msg = g_err;
        for (int i = 1; i < n_dev; i++) { free_scene(outs[i]); outs[i] = nullptr; }
        return fail(rc, "replica on device %d: %s", fail_dev, msg.c_str());
    }
    return MIPT_OK;
}


extern "C" {

const char *mipt_last_error(void) { return g_err.c_str(); }
int mipt_abi_version(void) { return MIPT_ABI_VERSION; }

int mipt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(MIPT_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return n;
}

void mipt_material_default(MiptMaterial *m) {        // scene.rs:148-167
    if (!m) return;
    m->base_color = {0.8f, 0.8f, 0.8f}; m->transmission = 0.0f;
    m->specular_tint = {1.0f, 1.0f, 1.0f}; m->ior = 1.45f;
    m->emission = {0.0f, 0.0f, 0.0f}; m->roughness = 1.0f;
    m->metallic = 0.0f; m->transparency = 1.0f;
    m->base_color_tex_id = m->transparency_tex_id = m->roughness_tex_id = m->metallic_tex_id =
        m->emission_tex_id = m->normal_tex_id = UINT32_MAX;
}

// Builds the device layout ONCE on the host, uploads it to device_ids[0] and replicates it from there (device-to-device copies: xGMI
// between the GPUs of a node) for mipt_multi_create.
static int scene_create_many(const MiptSceneDesc *desc, const int *device_ids, int n_dev, MiptScene **outs) {
    if (!desc || !outs || !device_ids || n_dev < 1) return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_create: null argument");
    for (int i = 0; i < n_dev; i++) outs[i] = nullptr;
    if (!desc->tris || desc->n_tris == 0) return fail(MIPT_ERR_INVALID_ARG, "scene has no triangles (the reference panics in BVH::build)");
    if (!desc->nodes || desc->n_nodes == 0) return fail(MIPT_ERR_INVALID_ARG, "scene has no BVH nodes");
    if (!desc->materials || desc->n_materials == 0) return fail(MIPT_ERR_INVALID_ARG, "scene has no materials");
    if (desc->n_textures && !desc->textures) return fail(MIPT_ERR_INVALID_ARG, "n_textures > 0 but textures == NULL");
    if (desc->n_tris > mipt::kMaxTris) return fail(MIPT_ERR_SCENE_LIMIT, "%u triangles exceed the 2^25 device-format limit", desc->n_tris);
    if ((desc->n_nodes & 1u) == 0u) return fail(MIPT_ERR_BVH, "node count %u is even: children are pushed in pairs after the root (bvh.rs:131-132)", desc->n_nodes);
    const uint32_t n_pairs = (desc->n_nodes - 1u) / 2u;
    if (n_pairs > mipt::kMaxPairs) return fail(MIPT_ERR_SCENE_LIMIT, "%u node pairs exceed the 2^24 device-format limit", n_pairs);
    const double t_begin = now_ms();

    // ---- validate the BVH on the host, before anything touches the device (the layout kernels index with these fields) ----
    // On up to 16 threads, and with the verdict of a sequential scan over the nodes: the error reported is the one at the LOWEST node
    // index (a node's own checks before its claims).  Every leaf claims its triangles and every inner node its child pair with an
    // atomic min of the node index; a second pass finds the claims a node lost -- "belongs to more than one ... (node i is the
    // second)" is then node i's error exactly when a lower node claimed the same thing, as in the sequential scan.
    {
        constexpr uint32_t kFree = 0xffffffffu;
        struct Err { uint32_t node = kFree; int kind = 0; uint32_t a = 0; };     // kind 1 bound, 2 leaf range, 3 child index, 4 shared pair, 5 shared triangle
        std::unique_ptr<uint32_t[]> pair_owner(new uint32_t[n_pairs ? n_pairs : 1]), tri_owner(new uint32_t[desc->n_tris]);
        parallel_for(n_pairs, [&](size_t b, size_t e) { for (size_t k = b; k < e; k++) pair_owner[k] = kFree; });
        parallel_for(desc->n_tris, [&](size_t b, size_t e) { for (size_t t = b; t < e; t++) tri_owner[t] = kFree; });
        std::mutex mu;
        Err first;
        auto report = [&](uint32_t node, int kind, uint32_t a) {
            std::lock_guard<std::mutex> lock(mu);
            if (node < first.node) { first.node = node; first.kind = kind; first.a = a; }
        };
        auto claim = [](uint32_t *p, uint32_t i) {
            uint32_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
            while (i < cur && !__atomic_compare_exchange_n(p, &cur, i, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        };
        auto own_error = [&](uint32_t i) -> int {                          // the checks that need nothing but node i
            const MiptNode &n = desc->nodes[i];
            // the kernel's exact-division fast path assumes finite bounds of magnitude <= 2^40 (beyond: refused).  (Axes with tiny
            // non-zero plane coordinates and the largest leaf are found by the device's own pass over the nodes, scene_device.hip.)
            const float lim = 1.0995116e12f;
            const float *b = &n.bounds_min.x, *c = &n.bounds_max.x;
            for (int k = 0; k < 3; k++) if (!(fabsf(b[k]) <= lim) || !(fabsf(c[k]) <= lim)) return 1;
            if (n.num_tris > 0) return (uint64_t)n.first_tri_or_child + n.num_tris > desc->n_tris ? 2 : 0;
            const uint32_t c2 = n.first_tri_or_child;
            return ((c2 & 1u) == 0u || (uint64_t)c2 + 1u >= desc->n_nodes || c2 <= i) ? 3 : 0;
        };
        parallel_for(desc->n_nodes, [&](size_t nb, size_t ne) {
            for (size_t ii = nb; ii < ne; ii++) {
                const uint32_t i = (uint32_t)ii;
                const int k = own_error(i);
                if (k) { report(i, k, 0); return; }                       // the rest of this range lies above an error: irrelevant
                const MiptNode &n = desc->nodes[i];
                // leaves partition the triangle array (bvh.rs:99-115 splits a node's range in place); the device stream re-packs leaf by
                // leaf, so a triangle in two leaves cannot be represented.  A tree, not a DAG: BVH::split_node pushes every child pair once
                // (bvh.rs:115-132).
                if (n.num_tris > 0) for (uint32_t t = n.first_tri_or_child; t < n.first_tri_or_child + n.num_tris; t++) claim(&tri_owner[t], i);
                else claim(&pair_owner[(n.first_tri_or_child - 1u) / 2u], i);
            }
        });
        const uint32_t upto = first.node == kFree ? desc->n_nodes : first.node;     // nodes above an error no longer matter
        parallel_for(upto, [&](size_t nb, size_t ne) {
            for (size_t ii = nb; ii < ne; ii++) {
                const uint32_t i = (uint32_t)ii;
                const MiptNode &n = desc->nodes[i];
                if (n.num_tris > 0) {
                    for (uint32_t t = n.first_tri_or_child; t < n.first_tri_or_child + n.num_tris; t++)
                        if (tri_owner[t] != i) { report(i, 5, t); return; }
                } else if (pair_owner[(n.first_tri_or_child - 1u) / 2u] != i) { report(i, 4, n.first_tri_or_child); return; }
            }
        });
        if (first.node != kFree) {
            const uint32_t i = first.node;
            const MiptNode &n = desc->nodes[i];
            switch (first.kind) {
            case 1: return fail(MIPT_ERR_SCENE_LIMIT, "node %u has a non-finite bound or one beyond 2^40", i);
            case 2: return fail(MIPT_ERR_BVH, "leaf node %u covers triangles [%u, %u+%u) beyond n_tris=%u", i, n.first_tri_or_child, n.first_tri_or_child, n.num_tris, desc->n_tris);
            case 3: return fail(MIPT_ERR_BVH, "inner node %u has child index %u (must be odd, > parent, and c+1 < n_nodes=%u)", i, n.first_tri_or_child, desc->n_nodes);
            case 4: return fail(MIPT_ERR_BVH, "child pair at node %u is referenced by more than one inner node (node %u is the second)", first.a, i);
            default: return fail(MIPT_ERR_BVH, "triangle %u belongs to more than one leaf (node %u is the second)", first.a, i);
            }
        }
        std::atomic<uint32_t> orphan{kFree};
        parallel_for(n_pairs, [&](size_t b, size_t e) {
            for (size_t k = b; k < e; k++) {
                if (pair_owner[k] != kFree) continue;
                uint32_t cur = orphan.load();
                while ((uint32_t)k < cur && !orphan.compare_exchange_weak(cur, (uint32_t)k)) {}
                return;
            }
        });
        if (orphan.load() != kFree) return fail(MIPT_ERR_BVH, "nodes %u and %u are not the children of any inner node", 2 * orphan.load() + 1, 2 * orphan.load() + 2);
    }
    // the reference indexes `materials[material_id]` and would panic; the lowest offender is reported, as a sequential scan would
    {
        std::atomic<uint32_t> bad_tri{UINT32_MAX};
        parallel_for(desc->n_tris, [&](size_t ib, size_t ie) {
            for (size_t ii = ib; ii < ie; ii++) {
                if (desc->tris[ii].material_id < desc->n_materials) continue;
                uint32_t cur = bad_tri.load();
                while ((uint32_t)ii < cur && !bad_tri.compare_exchange_weak(cur, (uint32_t)ii)) {}
            }
        });
        if (bad_tri.load() != UINT32_MAX)
            return fail(MIPT_ERR_INVALID_ARG, "triangle %u has material_id %u >= n_materials %u", bad_tri.load(), desc->tris[bad_tri.load()].material_id, desc->n_materials);
    }
    if (desc->nodes[0].num_tris == 0 && desc->nodes[0].first_tri_or_child != 1u)
        return fail(MIPT_ERR_BVH, "root's children must be nodes 1 and 2 (bvh.rs:121)");
    // ---- device(s).  Triangles and nodes cross PCIe once, to device_ids[0], and the layout -- triangle slots, the order of the pair
    // records, both triangle streams -- is produced there by the kernels mipt_scene_create_from_triangles uses after its build
    // (scene_device.hip); further replicas are device-to-device copies.  (Rounds 1-3 laid the scene out on host threads: 0.55 s for
    // 10 M triangles; that code now lives in libmipt_diag.so as the byte-for-byte reference of the kernels, tests/cpp/host_layout.cpp.)
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    for (int i = 0; i < n_dev; i++)
        if (device_ids[i] < 0 || device_ids[i] >= ndev) return fail(MIPT_ERR_HIP, "HIP device %d not available (%d visible)", device_ids[i], ndev);
    MiptScene *s = nullptr;
    int rc = mipt::scene_create_from_nodes(desc, device_ids[0], &s);
    if (rc) return rc;
    s->info.layout_ms += now_ms() - t_begin - s->info.total_ms;             // + the host checks above
    s->info.total_ms = now_ms() - t_begin;
    outs[0] = s;
    if ((rc = mipt::scene_clone_many(s, device_ids, n_dev, outs))) {
        const std::string msg = g_err;
        free_scene(outs[0]); outs[0] = nullptr;
        g_err = msg;
    }
    return rc;
}

static int scene_create_impl(const MiptSceneDesc *desc, int device_id, MiptScene **out) {
    if (!out) return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_create: null argument");
    *out = nullptr;
    return scene_create_many(desc, &device_id, 1, out);
}

// No C++ exception may cross the C ABI (the caller may be Rust or C): allocation failures become status codes.
#define MIPT_NO_THROW(call)                                                                                             \
    try { return call; }                                                                                               \
    catch (const std::bad_alloc &) { return fail(MIPT_ERR_INVALID_ARG, "out of host memory"); }                        \
    catch (const std::exception &e) { return fail(MIPT_ERR_INVALID_ARG, "internal error: %s", e.what()); }
int mipt_scene_create(const MiptSceneDesc *desc, int device_id, MiptScene **out) { MIPT_NO_THROW(scene_create_impl(desc, device_id, out)) }
} // extern "C"
// internal (mipt_multi.cpp): one scene on device_ids[0], device-to-device replicas on the others
static int scene_create_many_from_triangles(const MiptSceneDesc *desc, const int *device_ids, int n_dev, MiptScene **outs) {
    if (!desc || !outs || !device_ids || n_dev < 1) return fail(MIPT_ERR_INVALID_ARG, "mipt_multi_create_from_triangles: null argument");
    for (int i = 0; i < n_dev; i++) outs[i] = nullptr;
    int rc = mipt::scene_create_from_triangles(desc, device_ids[0], &outs[0]);
    if (rc == MIPT_OK) rc = mipt::scene_clone_many(outs[0], device_ids, n_dev, outs);
    if (rc) {
        const std::string msg = g_err;
        free_scene(outs[0]); outs[0] = nullptr;
        g_err = msg;
    }
    return rc;
}
int mipt::scene_create_replicas(const MiptSceneDesc *desc, const int *device_ids, int n_dev, MiptScene **outs, bool from_triangles) {
    MIPT_NO_THROW(from_triangles ? scene_create_many_from_triangles(desc, device_ids, n_dev, outs) : scene_create_many(desc, device_ids, n_dev, outs))
}
extern "C" {

void mipt_scene_destroy(MiptScene *scene) { free_scene(scene); }

int mipt_scene_info(const MiptScene *scene, MiptSceneInfo *out) {
    if (!scene || !out) return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_info: null argument");
    *out = scene->info;
    return MIPT_OK;
}

int mipt_scene_get_bvh(MiptScene *scene, MiptNode *nodes_out, uint32_t nodes_cap, uint32_t *n_nodes_out, uint32_t *tri_order_out) {
    if (!scene || !nodes_out) return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_get_bvh: null argument");
    if (!scene->d_nodes || !scene->d_tri_order)
        return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_get_bvh: this scene was created from host-built nodes (the caller already has them)");
    if (nodes_cap < scene->n_nodes) return fail(MIPT_ERR_INVALID_ARG, "mipt_scene_get_bvh: nodes_cap %u < %u nodes", nodes_cap, scene->n_nodes);
    HIP_TRY(hipSetDevice(scene->device));
    HIP_TRY(hipMemcpy(nodes_out, scene->d_nodes, (size_t)scene->n_nodes * sizeof(MiptNode), hipMemcpyDeviceToHost));
    if (tri_order_out) HIP_TRY(hipMemcpy(tri_order_out, scene->d_tri_order, scene->n_tris * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (n_nodes_out) *n_nodes_out = scene->n_nodes;
    return MIPT_OK;
}

uint64_t mipt_packed_pixels(uint32_t width, uint32_t height, uint32_t tile_world) {
    if (tile_world == 0) tile_world = 1;
    const uint64_t tiles = (uint64_t)((width + 7u) / 8u) * ((height + 7u) / 8u);
    return ((tiles + tile_world - 1u) / tile_world) * 64ull;
}

static int validate_options(const MiptOptions *opt) {
    if (!opt) return fail(MIPT_ERR_INVALID_ARG, "options == NULL");
    // Renderer::new (renderer.rs:15-26)
    if (opt->width == 0 || opt->height == 0) return fail(MIPT_ERR_INVALID_ARG, "Width and height must be greater than 0");
    if (opt->max_ray_depth == 0) return fail(MIPT_ERR_INVALID_ARG, "Max ray depth must be greater than 0");
    if (opt->samples == 0) return fail(MIPT_ERR_INVALID_ARG, "Sample count must be greater than 0");
    // seed wrap / absorbing zero seed: index + 87636354 must stay below 2^31
    if ((uint64_t)opt->width * opt->height >= 2147483648ull - 87636354ull)
        return fail(MIPT_ERR_INVALID_ARG, "width*height too large for the reference's 32-bit pixel seed");
    if (opt->seed_mode > MIPT_SEED_PER_SAMPLE) return fail(MIPT_ERR_INVALID_ARG, "unknown seed_mode %u", opt->seed_mode);
    if (opt->traversal > MIPT_TRAVERSAL_CULLED) return fail(MIPT_ERR_INVALID_ARG, "unknown traversal %u", opt->traversal);
    const uint32_t world = opt->tile_world ? opt->tile_world : 1u;
    if (opt->tile_rank >= world) return fail(MIPT_ERR_INVALID_ARG, "tile_rank %u >= tile_world %u", opt->tile_rank, world);
    if ((opt->flags & MIPT_FLAG_SUM) && opt->seed_mode != MIPT_SEED_PER_SAMPLE && opt->sample_begin > 1)
        return fail(MIPT_ERR_INVALID_ARG, "sample_begin needs MIPT_SEED_PER_SAMPLE (the pixel stream cannot be entered mid-way)");
    if (opt->shading > MIPT_SHADING_WGPU) return fail(MIPT_ERR_INVALID_ARG, "unknown shading mode %u", opt->shading);
    if (!(opt->cull_margin >= 0.0f) || opt->cull_margin > 1.0f) return fail(MIPT_ERR_INVALID_ARG, "cull_margin must be in [0, 1]");
    for (uint32_t r : opt->reserved)
        if (r) return fail(MIPT_ERR_INVALID_ARG, "reserved option fields must be 0");
    return MIPT_OK;
}

// `pack_single`: honour MIPT_FLAG_PACKED also at tile_world == 1 (mipt_render_multi with one device keeps the same
// gather + unpack path as with eight); through the public entry PACKED at world 1 means full-frame, as documented.
} // extern "C"
int mipt::render_device_impl(MiptScene *scene, const MiptCamera *camera, const MiptOptions *opt,
                             float *d_hdr_rgb, uint8_t *d_rgba8, void *hip_stream, MiptStats *stats, bool pack_single) {
    if (!scene || !camera) return fail(MIPT_ERR_INVALID_ARG, "mipt_render_device: null scene or camera");
    int rc = validate_options(opt);
    if (rc) return rc;
    if (!d_hdr_rgb) return fail(MIPT_ERR_INVALID_ARG, "mipt_render_device: d_hdr_rgb == NULL");
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t stream = (hipStream_t)hip_stream;

    const uint32_t world = opt->tile_world ? opt->tile_world : 1u;
    const bool packed = (opt->flags & MIPT_FLAG_PACKED) != 0 && (world > 1 || pack_single);
    if ((opt->flags & MIPT_FLAG_ACCUM) && !(opt->flags & MIPT_FLAG_SUM))
        return fail(MIPT_ERR_INVALID_ARG, "MIPT_FLAG_ACCUM needs MIPT_FLAG_SUM (a running sum, divided once at the end)");
    if (d_rgba8 && (packed || (opt->flags & MIPT_FLAG_SUM)))
        return fail(MIPT_ERR_INVALID_ARG, "RGBA8 output needs a full-frame mean buffer (not PACKED / SUM)");

    mipt::DevParams pr{};
    pr.width = opt->width; pr.height = opt->height; pr.samples = opt->samples; pr.max_depth = opt->max_ray_depth;
    pr.seed_mode = opt->shading == MIPT_SHADING_WGPU ? (uint32_t)MIPT_SEED_PER_SAMPLE : opt->seed_mode;   // the shader seeds per sample
    pr.sample_begin = opt->sample_begin ? opt->sample_begin : 1u;
    pr.sum_only = (opt->flags & MIPT_FLAG_SUM) ? 1u : 0u;
    pr.packed = packed ? 1u : 0u;
    pr.accumulate = (opt->flags & MIPT_FLAG_ACCUM) ? 1u : 0u;
    pr.tile_rank = opt->tile_rank; pr.tile_world = world;
    pr.tiles_x = (opt->width + 7u) / 8u; pr.tiles_y = (opt->height + 7u) / 8u;
    const uint64_t tiles = (uint64_t)pr.tiles_x * pr.tiles_y;
    // tiles owned by this rank: t in [0, tiles) with t % world == rank
    pr.n_local_tiles = (uint32_t)((tiles + world - 1u - opt->tile_rank) / world);
    pr.total_work = (unsigned long long)pr.n_local_tiles * 64ull;
    pr.aspect = (float)opt->width / (float)opt->height;       // cpu.rs:34
    pr.samples_f = (float)opt->samples;                       // cpu.rs:60
    pr.cull_scale = 1.0f + opt->cull_margin;
    pr.service_num = 3; pr.service_den = 8; pr.reverse_tiles = 0;   // service pass when >= 3/8 of the live lanes wait for one (tools/sweep_service.py)
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) pr.cam[c * 3 + r] = camera->look_at[c][r];
    pr.cam[9] = camera->position.x; pr.cam[10] = camera->position.y; pr.cam[11] = camera->position.z;
    pr.hdr = d_hdr_rgb;
    pr.stats = scene->d_stats;

    const bool count = (opt->flags & MIPT_FLAG_COUNT) != 0;
    const bool cull = opt->traversal == MIPT_TRAVERSAL_CULLED;
    const int occ = mipt::trace_blocks_per_cu(count, cull, (int)opt->shading);
    int bpc = occ;
    // Small shards (multi-GPU tile split: fewer pixels than resident lanes) are bound by the longest per-pixel chain --
    // a pixel's samples are sequential on one RNG stream -- and each chain steps faster with fewer co-resident waves:
    // keep >= 1.25 pixels per lane (measured on 1/8 of a 1080p frame: 25.4 ms at 3 blocks/CU vs 32.1 ms at 5).
    {
        const long long fit = (long long)(pr.total_work / (unsigned long long)(1.25 * scene->n_cu * mipt::kBlockThreads));
        const int cap = (int)(fit < 2 ? 2 : fit);
        if (cap < bpc) bpc = cap;
    }
    // Leaf phases (pt_kernel.hip): worth it when the SIMDs are the limit, i.e. every lane has many pixels to work through (full
    // 1080p frame, 6.3 pixels per resident lane: 83.0 -> 78.8 ms at period 4); a tile shard with 3.2 / 1.6 / 0.8 pixels per lane
    // (world 2 / 4 / 8) spends most of its time in the latency-bound tail and is 2 - 8 % slower with it: period 1 there
    pr.leaf_period = (pr.total_work >= 4ull * (unsigned long long)scene->n_cu * (unsigned long long)occ * mipt::kBlockThreads) ? 4u : 1u;
    pr.leaf_den = 4u;
    long long grid = (long long)scene->n_cu * bpc;
    const long long need_blocks = (long long)((pr.total_work + mipt::kBlockThreads - 1) / mipt::kBlockThreads);
    if (grid > need_blocks) grid = need_blocks;
    if (grid < 1) grid = 1;
    const size_t waves = (size_t)grid * mipt::kWavesPerBlock;
    if (waves > scene->ovf_waves) {
        if (scene->d_ovf) { (void)hipFree(scene->d_ovf); scene->d_ovf = nullptr; scene->ovf_waves = 0; }
        HIP_TRY(hipMalloc((void **)&scene->d_ovf, waves * (size_t)mipt::kStackOvf * 64 * sizeof(uint32_t)));
        scene->ovf_waves = waves;
    }
    pr.ovf = scene->d_ovf;

    // MIPT_FLAG_TOUCHED (diagnostic, counting build only): one bit per 128-B line of [pairs | tri_pos] and of tri_attr
    const bool touched = count && (opt->flags & MIPT_FLAG_TOUCHED) != 0 && opt->shading == 0;
    const size_t geom_lines = ((size_t)scene->dev.geom_bytes + 127) / 128, attr_lines = (scene->n_tris + 1) / 2;
    const size_t geom_words = (geom_lines + 31) / 32, attr_words = (attr_lines + 31) / 32;
    pr.touched = nullptr; pr.touched_attr_base = (uint32_t)(geom_words * 32);
    if (touched) {
        if (!scene->d_touched) HIP_TRY(hipMalloc((void **)&scene->d_touched, (geom_words + attr_words) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(scene->d_touched, 0, (geom_words + attr_words) * sizeof(uint32_t), stream));
        pr.touched = scene->d_touched;
    }
    HIP_TRY(hipMemsetAsync(scene->d_stats, 0, sizeof(mipt::DevStats), stream));
    HIP_TRY(hipEventRecord(scene->ev0, stream));
    HIP_TRY(mipt::launch_trace(scene->dev, pr, count, cull, (int)opt->shading, (int)grid, stream));
    HIP_TRY(hipEventRecord(scene->ev1, stream));
    if (touched) {
        HIP_TRY(mipt::launch_popcount(scene->d_touched, geom_words, &scene->d_stats->touched_geom, stream));
        HIP_TRY(mipt::launch_popcount(scene->d_touched + geom_words, attr_words, &scene->d_stats->touched_attr, stream));
    }
    if (d_rgba8) HIP_TRY(mipt::launch_tonemap(d_hdr_rgb, (unsigned long long)opt->width * opt->height, 1.0f, d_rgba8, stream));
    mipt::DevStats hs;
    HIP_TRY(hipMemcpyAsync(&hs, scene->d_stats, sizeof hs, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, scene->ev0, scene->ev1));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->kernel_ms = ms;
        stats->rays = hs.rays; stats->inner_steps = hs.inner_steps; stats->tri_tests = hs.tri_tests;
        stats->hits = hs.hits; stats->texel_fetches = hs.texel_fetches; stats->stack_overflows = hs.stack_overflows;
        stats->tex_clamped = hs.tex_clamped; stats->max_stack = hs.max_stack; stats->pixels = hs.pixels;
        stats->diag[0] = hs.d_iters; stats->diag[1] = hs.d_inner_lanes; stats->diag[2] = hs.d_leaf_lanes;
        stats->diag[3] = hs.d_iters_inner; stats->diag[4] = hs.d_iters_leaf; stats->diag[5] = hs.d_services;
        stats->diag[6] = hs.d_service_lanes; stats->diag[7] = hs.d_cycles_service; stats->diag[8] = hs.d_cycles_total; stats->diag[9] = hs.d_cycles_mem; stats->diag[10] = hs.d_cycles_tail;
        stats->touched_lines[0] = hs.touched_geom; stats->touched_lines[1] = hs.touched_attr;
    }
    if (hs.stack_overflows)
        return fail(MIPT_ERR_STACK, "traversal stack overflowed %llu times (capacity %d; the reference panics at 32, ray.rs:85)",
                    hs.stack_overflows, mipt::kStackLds + mipt::kStackOvf);
    return MIPT_OK;
}

extern "C" {

int mipt_render_device(MiptScene *scene, const MiptCamera *camera, const MiptOptions *opt,
                       float *d_hdr_rgb, uint8_t *d_rgba8, void *hip_stream, MiptStats *stats) {
    return mipt::render_device_impl(scene, camera, opt, d_hdr_rgb, d_rgba8, hip_stream, stats, false);
}

int mipt_render(MiptScene *scene, const MiptCamera *camera, const MiptOptions *opt,
                float *hdr_rgb, uint8_t *rgba8, MiptStats *stats) {
    if (!scene) return fail(MIPT_ERR_INVALID_ARG, "mipt_render: null scene");
    int rc = validate_options(opt);
    if (rc) return rc;
    if (opt->flags & MIPT_FLAG_ACCUM) return fail(MIPT_ERR_INVALID_ARG, "MIPT_FLAG_ACCUM needs a caller-owned device buffer: use mipt_render_device");
    HIP_TRY(hipSetDevice(scene->device));
    const uint32_t world = opt->tile_world ? opt->tile_world : 1u;
    const bool packed = (opt->flags & MIPT_FLAG_PACKED) != 0 && world > 1;
    const uint64_t n_pix = (uint64_t)opt->width * opt->height;
    const uint64_t n_out = packed ? mipt_packed_pixels(opt->width, opt->height, world) : n_pix;
    size_t have = scene->hdr_floats * sizeof(float);
    if ((rc = ensure((void **)&scene->d_hdr, &have, (size_t)n_out * 3 * sizeof(float)))) return rc;
    scene->hdr_floats = have / sizeof(float);
    if (world > 1 && !packed) HIP_TRY(hipMemsetAsync(scene->d_hdr, 0, (size_t)n_out * 3 * sizeof(float), nullptr));
    uint8_t *d_rgba = nullptr;
    if (rgba8) {
        if ((rc = ensure((void **)&scene->d_rgba, &scene->rgba_bytes, (size_t)n_pix * 4))) return rc;
        d_rgba = scene->d_rgba;
    }
    rc = mipt_render_device(scene, camera, opt, scene->d_hdr, d_rgba, nullptr, stats);
    if (rc && rc != MIPT_ERR_STACK) return rc;
    if (hdr_rgb) HIP_TRY(hipMemcpy(hdr_rgb, scene->d_hdr, (size_t)n_out * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (rgba8) HIP_TRY(hipMemcpy(rgba8, scene->d_rgba, (size_t)n_pix * 4, hipMemcpyDeviceToHost));
    return rc;
}

int mipt_unpack_tiles(const float *d_packed_all, uint32_t width, uint32_t height, uint32_t tile_world,
                      float *d_hdr_rgb, void *hip_stream) {
    if (!d_packed_all || !d_hdr_rgb || width == 0 || height == 0 || tile_world == 0)
        return fail(MIPT_ERR_INVALID_ARG, "mipt_unpack_tiles: bad argument");
    HIP_TRY(mipt::launch_unpack_tiles(d_packed_all, width, height, tile_world, d_hdr_rgb, (hipStream_t)hip_stream));
    return MIPT_OK;
}

int mipt_tonemap_device(const float *d_hdr_rgb, uint64_t n_pixels, float divisor, uint8_t *d_rgba8, void *hip_stream) {
    if (!d_hdr_rgb || !d_rgba8 || n_pixels == 0) return fail(MIPT_ERR_INVALID_ARG, "mipt_tonemap_device: bad argument");
    HIP_TRY(mipt::launch_tonemap(d_hdr_rgb, n_pixels, divisor, d_rgba8, (hipStream_t)hip_stream));
    return MIPT_OK;
}

int mipt_postprocess_device(const float *d_hdr_rgb, uint64_t n_pixels, float divisor, uint16_t *d_rgba16, void *hip_stream) {
    if (!d_hdr_rgb || !d_rgba16 || n_pixels == 0) return fail(MIPT_ERR_INVALID_ARG, "mipt_postprocess_device: bad argument");
    HIP_TRY(mipt::launch_postprocess(d_hdr_rgb, n_pixels, divisor, d_rgba16, (hipStream_t)hip_stream));
    return MIPT_OK;
}

// Camera::update_view + Mat4f::look_at (scene.rs:181-194, mat4.rs:25-44).  Host-only, once per
// frame; sin/cos are the platform libm's, as in the reference.
int mipt_camera_from_pose(const float position[3], float pitch_deg, float yaw_deg, MiptCamera *out) {
    if (!position || !out) return fail(MIPT_ERR_INVALID_ARG, "mipt_camera_from_pose: null argument");
    struct V { float x, y, z; };
    auto sub = [](V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto add = [](V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; };
    auto cross = [](V a, V b) { return V{(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; };
    auto norm = [](V a) { float l = sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); return V{a.x / l, a.y / l, a.z / l}; };
    const float k = 0.017453292519943295769236907684886f;        // f32::to_radians
    const float yaw = yaw_deg * k, pitch = pitch_deg * k;
    const V direction{cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch)};
    const V pos{position[0], position[1], position[2]};
    const V forward = norm(direction);
    const V right = norm(cross(V{0.0f, 1.0f, 0.0f}, forward));
    const V up = cross(forward, right);
    const V f = norm(sub(pos, add(pos, forward)));                // look_at(from, to = from + forward, up)
    const V r = norm(cross(up, f));
    const V u = cross(f, r);
    memset(out, 0, sizeof *out);
    out->look_at[0][0] = r.x; out->look_at[0][1] = r.y; out->look_at[0][2] = r.z;
    out->look_at[1][0] = u.x; out->look_at[1][1] = u.y; out->look_at[1][2] = u.z;
    out->look_at[2][0] = f.x; out->look_at[2][1] = f.y; out->look_at[2][2] = f.z;
    out->look_at[3][0] = pos.x; out->look_at[3][1] = pos.y; out->look_at[3][2] = pos.z; out->look_at[3][3] = 1.0f;
    out->position = {pos.x, pos.y, pos.z};
    return MIPT_OK;
}

} // extern "C"

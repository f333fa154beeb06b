// mipt_scene.h -- the device-resident scene behind the opaque MiptScene handle, shared by the translation units that build it
// (mipt_api.cpp: host layout + upload; scene_device.hip: layout built on the GPU; both: replicas by peer copy) and by the test
// library's checksum hook (tests/cpp/scene_hooks.hip, libmipt_diag.so).  Internal: HIP types, not part of include/mipt.h.
#pragma once
#include "../../include/mipt.h"
#include "pt_kernel.h"

#include <stddef.h>

#include <vector>

struct MiptScene {
    int device = 0;
    mipt::DevScene dev{};
    void *d_geom = nullptr, *d_tri_attr = nullptr, *d_mats = nullptr, *d_mats_full = nullptr, *d_texels = nullptr;
    size_t geom_alloc = 0, attr_bytes = 0, mats_bytes = 0, mats_full_bytes = 0, texel_bytes = 0;   // allocation / payload sizes (replicas copy these)
    // a scene whose BVH was built on the device keeps the tree for mipt_scene_get_bvh: nodes in the reference's order and the
    // triangle permutation BVH::build applied (reordered[t] = original[tri_order[t]])
    MiptNode *d_nodes = nullptr;
    uint32_t n_nodes = 0;
    uint32_t *d_tri_order = nullptr;
    MiptSceneInfo info{};
    // workspace
    mipt::DevStats *d_stats = nullptr;
    uint32_t *d_ovf = nullptr;
    size_t ovf_waves = 0;
    uint32_t *d_touched = nullptr;          // MIPT_FLAG_TOUCHED: line bitmap, allocated on first use
    size_t n_tris = 0;
    float *d_hdr = nullptr;
    size_t hdr_floats = 0;
    uint8_t *d_rgba = nullptr;
    size_t rgba_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 0;
    uint32_t max_leaf = 0;
};

namespace mipt {

void free_scene(MiptScene *s);
// stats buffer, events, CU count: everything a scene needs besides its geometry.  On failure the scene is left for free_scene.
int scene_finish_workspace(MiptScene *s);
// A replica of `src` in the memory of `device` by device-to-device copies (xGMI between GPUs of a node; a plain copy on the same
// device), queued on `stream` of... the current device is left at `device`.  No host staging.
int scene_clone_to(const MiptScene *src, int device, MiptScene **out);
// outs[1 .. n_dev) = replicas of `src` on device_ids[1 .. n_dev), every copy queued before the first is waited for (seven pulls from
// one GPU run on seven xGMI links at once); all-or-nothing.
int scene_clone_many(const MiptScene *src, const int *device_ids, int n_dev, MiptScene **outs);

// Host tables of a scene's materials and textures (mipt_api.cpp): the 64-B CPU-shading record, the 128-B record of the wgpu
// material model and the texel pool.
struct MaterialTables {
    std::vector<DevMaterial> mats;
    std::vector<DevMaterialFull> mats_full;
    std::vector<uint32_t> texels;             // the pool, when build_material_tables was asked to gather it
    std::vector<uint32_t> tex_offset;         // first texel of texture i in the pool
    uint64_t n_texels = 0;
};
// validates the texture references; gather_texels = false leaves `texels` empty (the caller moves the textures into the pool itself)
int build_material_tables(const MiptSceneDesc *desc, MaterialTables *out, bool gather_texels = true);
// on the current device.  With an empty `texels` the pool is allocated (n_texels) but not filled.
int upload_material_tables(MiptScene *s, const MaterialTables &t);

// BVH::build (bvh.rs:13-161) on the GPU with everything staying in HBM (bvh_build_device.hip): `d_tris` in, the node array in the
// reference's order and the triangle permutation out (reordered[t] = original[d_tri_order[t]]); both hipMalloc'ed, owned by the caller.
struct ResidentBvh {
    MiptNode *d_nodes = nullptr;
    uint32_t n_nodes = 0;
    uint32_t *d_tri_order = nullptr;
    uint32_t levels = 0;                  // levels the level-synchronous part ran
    double build_ms = 0.0;                // HIP events around the build kernels
};
int bvh_build_resident(const MiptTriangle *d_tris, uint32_t n_tris, int device_id, ResidentBvh *out);
void bvh_builder_resolve_kernels();         // optional: what the first launch of each builder kernel would pay, up front (call with the device set)

} // namespace mipt

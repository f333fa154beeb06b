// mipt_internal.h -- functions shared between the translation units of libmipt.so.  None of them is exported: the library is built
// with -fvisibility=hidden and only the MIPT_API declarations of include/mipt.h leave it.  (Host C++ only: no HIP types here, so the
// CPU sanitizer builds of tests/cpp/ can include it.)
#pragma once
#include "../../include/mipt.h"

void mipt_internal_set_error(const char *msg);          // sets the calling thread's mipt_last_error() text (mipt_api.cpp)

namespace mipt {

// ---- device-layout orders ----
// Number of breadth-first levels at the top of the pair-record order (scene_device.hip LevelOp; 8 ... 14 measure the same, 2.005 - 2.03 G
// line fills per frame of config M; 16: 2.20, 18: 2.31, 0: 2.23).
constexpr uint32_t kPairLayoutTop = 12;
inline uint32_t pair_order_top() { return kPairLayoutTop; }
// The host restatements of both orders -- NOT in libmipt.so: tests/cpp/layout_order.cpp, linked into libmipt_diag.so only, where they
// are the reference the layout kernels are checked against.
// Order of the 64-B pair records in HBM: order_out[j] = reference pair index of record j, 0xffffffff = pad record.
int pair_order(const MiptNode *nodes, uint32_t n_nodes, uint32_t *order_out, uint32_t cap, uint32_t *n_records_out);
// Slot of every triangle's 64-B record in the intersection stream.
int tri_slots(const MiptNode *nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t *slot_out, uint32_t *n_slots_out);

// ---- scene_device.hip ----
// mipt_scene_create_from_triangles without the exception fence.
int scene_create_from_triangles(const MiptSceneDesc *desc, int device_id, MiptScene **out);
// the same with the caller's node array (ALREADY validated: mipt_scene_create's host checks) instead of a build; triangles in the tree's order
int scene_create_from_nodes(const MiptSceneDesc *desc, int device_id, MiptScene **out);

// ---- mipt_api.cpp, used by mipt_multi.cpp ----
// One scene on device_ids[0] -- from the caller's nodes or, with from_triangles, built on that device -- and device-to-device
// replicas on the others.
int scene_create_replicas(const MiptSceneDesc *desc, const int *device_ids, int n_dev, MiptScene **outs, bool from_triangles);
// mipt_render_device with `pack_single`: honour MIPT_FLAG_PACKED also at tile_world == 1.
int render_device_impl(MiptScene *scene, const MiptCamera *camera, const MiptOptions *opt, float *d_hdr_rgb, uint8_t *d_rgba8,
                       void *hip_stream, MiptStats *stats, bool pack_single);

} // namespace mipt

// pt_kernel.h -- device-side scene layout and launch interface (internal to libmipt.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

// ---- HBM layout (see DESIGN.md "Data layout") ------------------------------------------
// pairs    : n_pairs x 64 B.  pair k = { nodes[2k+1], nodes[2k+2] } of the reference array
//            (children are always pushed adjacently at odd indices, bvh.rs:121,131-132), each
//            child kept in the reference's Node layout {min.xyz, a, max.xyz, n}: n = num_tris;
//            a = first triangle (leaf) or the PAIR index of its own children (inner).
// tri_pos  : n_tris x 64 B intersection stream {v0.xyz, e1.xyz, e2.xyz, 7 pad words} (a 48-B stride straddles 128-B lines:
//            1.41 memory requests per record instead of 1),
//            e1 = v1 - v0, e2 = v2 - v0 rounded once on the host exactly as ray.rs:24-25 does.
// tri_attr : n_tris x 64 B shading stream {n0,n1,n2 (9 f32), uv0,uv1,uv2 (6 f32), material_id}.
// mats     : n_materials x 64 B {base_color.xyz, emission.xyz} -- the only Material fields cpu/ray.rs:162-176 reads --
//            plus the two textures' {texel offset into `texels`, width, height} inlined; texels = RGBA8 as u32.
// 64 B: the texture descriptors (texel offset, width, height) ride in the material record, so a textured hit costs
// attr -> material -> texel (three dependent fetches) instead of four.  tex_w == 0 means "no texture" (u32::MAX id).
struct DevMaterial {
    float base[3]; uint32_t base_off;
    float emis[3]; uint32_t emis_off;
    uint32_t base_w, base_h, emis_w, emis_h;
    uint32_t pad[4];
};

// 128 B: every Material field the wgpu backend's shader reads (rt_compute.wgsl:41-56) + the six textures' descriptors
// {texel offset, width, height}; width == 0 = no texture.  Order: base, transparency, roughness, metallic, emission, normal.
struct DevMaterialFull {
    float base[3]; float transmission;
    float emission[3]; float ior;
    float roughness, metallic, transparency, pad0;
    uint32_t tex[6][3];
    uint32_t pad1[2];
};

struct DevScene {
    const float4 *pairs;          // = geom: [pairs | tri_pos] live in ONE allocation so the traversal step can address either
    const float4 *tri_pos;        //   through one buffer descriptor with a 32-bit byte offset (tri_off_bytes = n_pairs * 64)
    uint32_t tri_off_bytes, geom_bytes;
    uint32_t tiny_axes;           // bit c: some bounding plane has a coordinate 0 < |p_c| < 2^-76 on axis c (see ray_safe, pt_kernel.hip)
    const float4 *tri_attr;
    const DevMaterial *mats;
    const DevMaterialFull *mats_full;   // shading mode 1 (rt_compute.wgsl material model)
    const uint32_t *texels;
    uint32_t n_pairs, n_tris, n_mats, n_texs;
    uint32_t root_a, root_n;      // root node: leaf (root_n > 0: tris [root_a, root_a+root_n)) or inner (pair 0)
};

struct DevStats {                 // zeroed before every launch
    unsigned long long queue;     // next work index
    unsigned long long rays, inner_steps, tri_tests, hits, texel_fetches;
    unsigned long long stack_overflows, tex_clamped, max_stack, pixels;
    // wave-level occupancy diagnostics (COUNT build): traversal iterations, lanes doing inner / leaf work,
    // iterations that ran the inner / leaf branch, service passes, lanes serviced
    unsigned long long d_iters, d_inner_lanes, d_leaf_lanes, d_iters_inner, d_iters_leaf, d_services, d_service_lanes;
    unsigned long long d_cycles_service, d_cycles_total, d_cycles_mem, d_cycles_tail;   // per-wave s_memtime cycles spent in service passes / alive
    unsigned long long touched_geom, touched_attr;                                     // distinct 128-B lines read (MIPT_FLAG_TOUCHED)
};

struct DevParams {
    uint32_t width, height, samples, max_depth;
    uint32_t seed_mode, sample_begin, sum_only, packed;
    uint32_t accumulate, pad1;              // accumulate: hdr += this call's result (progressive rendering)
    uint32_t tile_rank, tile_world, tiles_x, tiles_y;
    uint32_t n_local_tiles, pad0;
    unsigned long long total_work;          // n_local_tiles * 64
    float aspect;                           // width as f32 / height as f32 (cpu.rs:34)
    float samples_f;
    float cull_scale;                       // 1 + cull_margin
    uint32_t reverse_tiles;                 // hand out the tile list back to front (bottom rows first)
    uint32_t service_num, service_den;      // run the service pass when need/live >= num/den
    uint32_t leaf_period, leaf_den;         // triangle tests run when leaf_period iterations have passed since the last ones or
                                            // 1/leaf_den of the traversing lanes wait for one (pt_kernel.hip "leaf phases"); period 1 = always
    float cam[12];                          // look_at columns 0..2 (xyz each), position
    float *hdr;                             // full-frame or rank-packed, 3 f32 per pixel
    uint32_t *ovf;                          // traversal-stack overflow area [wave][entry][lane]
    DevStats *stats;
    uint32_t *touched;                      // COUNT build + MIPT_FLAG_TOUCHED: one bit per 128-B line of [geom | tri_attr] (else NULL)
    uint32_t touched_attr_base;             // first bit of the tri_attr stream in `touched`
};

constexpr int kStackLds = 16;               // per-lane traversal-stack entries held in LDS
constexpr int kStackOvf = 48;               // further entries spilled to HBM (rarely touched)
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kTriPosStride = 64;   // bytes per record of the intersection stream: 48 packed, 64 = never straddles a 128-B line
constexpr int kBlockThreads = 64 * kWavesPerBlock;
constexpr uint32_t kMaxTris = 1u << 25;     // stack-entry encoding: 25-bit triangle index
constexpr uint32_t kMaxPairs = 1u << 24;    // 24-bit pair index in the child-ref form

// Launchers (stream-ordered; no allocation, no synchronisation inside).
hipError_t launch_trace(const DevScene &sc, const DevParams &pr, bool count, bool cull, int shading,
                        int grid_blocks, hipStream_t stream);
int trace_blocks_per_cu(bool count, bool cull, int shading);     // occupancy query
hipError_t launch_unpack_tiles(const float *packed_all, uint32_t width, uint32_t height,
                               uint32_t tile_world, float *hdr, hipStream_t stream);
hipError_t launch_tonemap(const float *hdr, unsigned long long n_pixels, float divisor,
                          uint8_t *rgba8, hipStream_t stream);

// number of set bits in words [0, n_words) of `bitmap`, added to *out (a device counter)
hipError_t launch_popcount(const uint32_t *bitmap, unsigned long long n_words, unsigned long long *out, hipStream_t stream);
hipError_t launch_divide(float *hdr, unsigned long long n_floats, float divisor, hipStream_t stream);
hipError_t launch_postprocess(const float *hdr, unsigned long long n_pixels, float divisor, uint16_t *rgba16,
                              hipStream_t stream);

} // namespace mipt

/*
 * mipt.h -- C ABI of the MI355X path-tracing backend (libmipt.so).
 *
 * This is the drop-in boundary for the reference's renderer seam: the
 * `match self.options.backend` in Renderer::render (reference src/renderer.rs:57-63)
 * whose arms all have the shape (Renderer, &Scene) -> Vec<u8>
 * (src/renderer/backend/cpu.rs:13, src/renderer/backend/gpu.rs:14).  A third arm
 * `RendererBackend::MI355X` calls mipt_scene_create / mipt_render / mipt_scene_destroy
 * (binding shown in INTEGRATION.md).  The payload is exactly what the reference's wgpu
 * backend already ships to a device (src/renderer/backend/gpu.rs:329-339,356-367,455-459):
 * bytemuck::Pod arrays of Triangle (112 B), Node (32 B), Material (80 B), RGBA8 textures
 * and the 80-byte UniformCamera.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types; every entry point returns
 * 0 on success or a negative MiptStatus and never unwinds or aborts across the ABI; the
 * message for the last failure on the calling thread is mipt_last_error().  Inputs are
 * borrowed for the duration of the call only (mipt_scene_create copies to HBM); output
 * buffers are caller-allocated.  The library fails loudly (MIPT_ERR_HIP) when no gfx950
 * device or code object is available: there is no CPU fallback inside libmipt.so.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_ABI_VERSION 4

/* libmipt.so is built with -fvisibility=hidden: exactly the functions declared here are exported
 * (tests/test_abi.py compares `nm -D --defined-only` with this header). */
#if defined(__GNUC__)
#define MIPT_API __attribute__((visibility("default")))
#else
#define MIPT_API
#endif

/* ---- PODs, byte-identical to the reference's #[repr(C, align(16))] structs ---------- */
typedef struct { float x, y, z; } MiptVec3;                 /* src/math/vec3.rs:55-59  (12 B) */

typedef struct {                                            /* src/scene.rs:87-94      (32 B) */
    MiptVec3 position; float tex_coord_x;
    MiptVec3 normal;   float tex_coord_y;
} MiptVertex;

typedef struct {                                            /* src/scene.rs:97-103    (112 B) */
    MiptVertex vertices[3];
    uint32_t   material_id;
    uint8_t    _pad[12];
} MiptTriangle;

typedef struct {                                            /* src/bvh.rs:164-171      (32 B) */
    MiptVec3 bounds_min; uint32_t first_tri_or_child;
    MiptVec3 bounds_max; uint32_t num_tris;                 /* leaf iff num_tris > 0 */
} MiptNode;

typedef struct {                                            /* src/scene.rs:129-146    (80 B) */
    MiptVec3 base_color;    float transmission;
    MiptVec3 specular_tint; float ior;
    MiptVec3 emission;      float roughness;
    float    metallic, transparency;
    uint32_t base_color_tex_id, transparency_tex_id, roughness_tex_id,
             metallic_tex_id, emission_tex_id, normal_tex_id;   /* UINT32_MAX = none */
} MiptMaterial;

typedef struct {                                            /* src/renderer/backend/gpu.rs:480-486 (80 B) */
    float    look_at[4][4];                                 /* Mat4f data[col][row], src/math/mat4.rs:6-10 */
    MiptVec3 position; float _pad;
} MiptCamera;

typedef struct {                                            /* src/texture.rs:4-10; payload as gpu.rs:360-367 */
    uint32_t width, height;
    const uint8_t *rgba8;                                   /* width*height*4 bytes, rows as stored by Texture::load (v-flipped) */
} MiptTexture;

typedef struct {                                            /* what StorageBuffers::new uploads, gpu.rs:329-391 */
    const MiptTriangle *tris;      uint32_t n_tris;
    const MiptNode     *nodes;     uint32_t n_nodes;        /* BVH::build output, src/bvh.rs:13-54 */
    const MiptMaterial *materials; uint32_t n_materials;    /* indexed by Triangle.material_id */
    const MiptTexture  *textures;  uint32_t n_textures;
} MiptSceneDesc;

typedef struct MiptScene MiptScene;                         /* opaque, device-resident */

/* ---- options: RendererOptions (src/renderer.rs:96-104) + what the MI355X path adds ---- */
enum MiptSeedMode {
    MIPT_SEED_PIXEL_STREAM = 0,  /* cpu.rs:28-29: one xorshift stream per pixel, all samples in sequence */
    MIPT_SEED_PER_SAMPLE   = 1   /* rt_compute.wgsl:102: reseed per (sample, x, y); splittable by sample */
};
enum MiptTraversal {
    MIPT_TRAVERSAL_REFERENCE = 0, /* cpu/ray.rs:69-81: slab test without t-max cull (the CPU backend) */
    MIPT_TRAVERSAL_CULLED    = 1  /* rt_compute.wgsl:341-349: + `t_near < best*(1+cull_margin)` cull (margin 0 = the wgpu backend's rule) */
};
/* Best-hit culling is not result-identical to the CPU backend at margin 0: Moller-Trumbore is not
 * watertight, so a ray on a shared edge can hit both neighbours with distances one ulp apart, and
 * culling the second leaf (slab t_near >= best) keeps the first-found instead of the closest
 * (measured: ~1e-5 of rays on the 10 M-triangle scene).  A relative margin keeps every near-tie
 * candidate in play; with 2^-7 the culled frame is bit-identical to the reference traversal on
 * every scene tested (tests/, bench.py re-checks it on each run) at +1 % node visits. */
#define MIPT_CULL_MARGIN_SAFE 0.0078125f
enum MiptShading {
    MIPT_SHADING_CPU  = 0,        /* cpu/ray.rs:141-202: the rayon backend's trace (the parity target) */
    MIPT_SHADING_WGPU = 1         /* rt_compute.wgsl:126-294: the wgpu shader's material model (GGX-VNDF specular, Schlick Fresnel,
                                   * refraction + Beer absorption, alpha cut-out, Russian roulette from depth 4, normal maps,
                                   * bilinear/repeat textures with 2.2 gamma); per-sample seeds always (rt_compute.wgsl:102) */
};
enum MiptFlags {
    MIPT_FLAG_COUNT  = 1u << 0,   /* counting build: fill rays / inner_steps / tri_tests / ... in MiptStats */
    MIPT_FLAG_PACKED = 1u << 1,   /* tile-sharded output is rank-packed (tile-major) instead of full-frame */
    MIPT_FLAG_SUM    = 1u << 2,   /* hdr = sum over samples (no division): sample-sharded accumulation */
    MIPT_FLAG_ACCUM  = 1u << 3,   /* with SUM, device buffers only: hdr += this call's samples (progressive rendering,
                                   * the resumable form of the per-sample loop at gpu.rs:17-77) */
    MIPT_FLAG_TOUCHED = 1u << 4   /* with COUNT: also mark every 128-byte line of the BVH / triangle streams the launch reads in a
                                   * device bitmap and report the number of distinct lines (MiptStats.touched_lines): the
                                   * compulsory memory traffic of the frame.  Diagnostic: slows the counting launch down */
};

typedef struct {
    uint32_t width, height;       /* output_image_dimensions, renderer.rs:100 */
    uint32_t samples;             /* renderer.rs:98  (> 0) */
    uint32_t max_ray_depth;       /* renderer.rs:99  (> 0) */
    uint32_t seed_mode;           /* MiptSeedMode */
    uint32_t traversal;           /* MiptTraversal */
    uint32_t flags;               /* MiptFlags */
    uint32_t tile_rank;           /* image-tile shard: this rank ...                        */
    uint32_t tile_world;          /* ... of this many (0 or 1 = whole image); 8x8 tiles, round-robin */
    uint32_t sample_begin;        /* PER_SAMPLE: first sample number (0 -> 1, as gpu.rs:252 starts at 1) */
    float    cull_margin;         /* MIPT_TRAVERSAL_CULLED: relative margin (>= 0); see MIPT_CULL_MARGIN_SAFE */
    uint32_t shading;             /* MiptShading */
    uint32_t reserved[4];         /* must be 0 */
} MiptOptions;

typedef struct {
    double   kernel_ms;           /* HIP-event time of the trace kernel on its launch stream */
    uint64_t rays;                /* traverse_bvh invocations (ray.rs:150) [COUNT] */
    uint64_t inner_steps;         /* inner-node visits, two child records each [COUNT] */
    uint64_t tri_tests;           /* intersect_tri calls [COUNT] */
    uint64_t hits;                /* rays that hit [COUNT] */
    uint64_t texel_fetches;       /* Texture::color_at calls [COUNT] */
    uint64_t stack_overflows;     /* pushes dropped: traversal stack full (reference panics, ray.rs:85) */
    uint64_t tex_clamped;         /* texel index clamped (reference panics, texture.rs:37) */
    uint64_t max_stack;           /* deepest stack occupancy [COUNT] */
    uint64_t pixels;              /* pixels this call produced */
    /* wave-occupancy diagnostics [COUNT]: traversal iterations (per wave); sum of lanes on an inner step; sum of lanes on
     * a leaf step; iterations that executed the inner branch; the leaf branch; service passes; lanes serviced;
     * wave-cycles inside service passes; wave-cycles alive; wave-cycles waiting for the traversal loads (only in a
     * -DMIPT_DIAG_STAMPS=1 build); wave-cycles between a wave first finding the queue empty and its exit (tail) */
    uint64_t diag[11];
    /* [COUNT | TOUCHED] distinct 128-byte lines read: [0] of the BVH pair records + triangle intersection stream (one
     * allocation), [1] of the triangle attribute stream.  x 128 = the bytes a launch must move at least once. */
    uint64_t touched_lines[2];
} MiptStats;

enum MiptStatus {
    MIPT_OK = 0,
    MIPT_ERR_INVALID_ARG   = -1,  /* renderer.rs:15-26 invariants, null pointers, bad ranges */
    MIPT_ERR_HIP           = -2,  /* HIP runtime error / no gfx950 device */
    MIPT_ERR_SCENE_LIMIT   = -3,  /* scene exceeds device-format limits */
    MIPT_ERR_BVH           = -4,  /* malformed BVH handed to mipt_scene_create */
    MIPT_ERR_IO            = -5,  /* file not found / parse error (OBJ loader) */
    MIPT_ERR_STACK         = -6,  /* traversal stack overflowed during render (result incomplete) */
    MIPT_ERR_RCCL          = -7   /* RCCL communicator / collective error (mipt_render_multi) */
};

/* ---- the seam ----------------------------------------------------------------------- */

/* Copies the scene to HBM of HIP device `device_id`, re-basing the BVH into 64-byte child-pair
 * records and splitting triangles into an intersection stream (40 B of payload -- v0, e1, e2 and the triangle's
 * reference index -- at a 64-byte stride, so a record never straddles a 128-byte line) and a 64-byte shading stream.  The tree is
 * validated on the host first (MIPT_ERR_BVH / _SCENE_LIMIT / _INVALID_ARG before any device call); triangles and nodes then cross PCIe
 * once and the layout is produced by GPU kernels (10 M triangles: 0.07 s).  Replaces State::new / StorageBuffers::new (gpu.rs:96-118, 329-401). */
MIPT_API int mipt_scene_create(const MiptSceneDesc *desc, int device_id, MiptScene **out);
MIPT_API void mipt_scene_destroy(MiptScene *scene);

/* The same scene from its TRIANGLES alone -- desc->nodes / n_nodes are ignored (may be NULL / 0): BVH::build (bvh.rs:13-161) runs on
 * the GPU and everything after the one host -> device copy of the triangle array stays in HBM: the node array in the reference's
 * order, the re-based pair records, both triangle streams.  The tree, the triangle order and every byte of the device layout are
 * identical to mipt_bvh_build + mipt_scene_create (sign of zero in a bound aside); only the time differs (10 M triangles: ~0.08 s
 * against ~10 s of host build or 1.3 s of mipt_bvh_build_device + mipt_scene_create).  The triangle array is read in the caller's
 * order and not modified; mipt_scene_get_bvh returns what BVH::build would have left in the host's Scene. */
MIPT_API int mipt_scene_create_from_triangles(const MiptSceneDesc *desc, int device_id, MiptScene **out);

/* The tree of a scene made by mipt_scene_create_from_triangles: nodes_out receives the node array (BVH::build's output, nodes_cap >=
 * 2 * n_tris - 1 is always enough), tri_order_out (n_tris entries, may be NULL) the permutation bvh.rs:99-108 applied to the
 * triangles: reordered[t] = original[tri_order_out[t]].  MIPT_ERR_INVALID_ARG for a scene made from host-built nodes. */
MIPT_API int mipt_scene_get_bvh(MiptScene *scene, MiptNode *nodes_out, uint32_t nodes_cap, uint32_t *n_nodes_out, uint32_t *tri_order_out);

/* What a scene holds and what it took to get it there (host clock, ms; build_ms: HIP events). */
typedef struct {
    uint32_t n_tris, n_nodes;
    uint32_t n_pair_records;      /* 64-byte pair records incl. line padding */
    uint32_t max_leaf;            /* largest leaf (triangles) */
    uint64_t geometry_bytes;      /* pair records + both triangle streams in HBM */
    uint32_t built_on_device;     /* 1: mipt_scene_create_from_triangles */
    uint32_t replica_of_device;   /* for a replica made by device-to-device copy: the source device ordinal + 1; else 0 */
    double   upload_ms;           /* host -> device copies (geometry + materials + textures) */
    double   build_ms;            /* device BVH build, kernels only; 0 for host-built nodes */
    double   layout_ms;           /* device layout: host re-layout (mipt_scene_create) or layout kernels */
    double   total_ms;            /* the whole create call */
} MiptSceneInfo;
MIPT_API int mipt_scene_info(const MiptScene *scene, MiptSceneInfo *out);

/* Renders into HOST buffers and blocks until done.  Replaces cpu::render_scene
 * (cpu.rs:13-68) / gpu::render_scene_to_buffer (gpu.rs:14-94).
 *   hdr_rgb : width*height*3 f32, linear mean radiance per pixel (the value cpu.rs:60 holds
 *             before sRGB), row 0 = top; may be NULL.
 *   rgba8   : width*height*4 bytes, exactly the Vec<u8> cpu.rs:63-67 returns; may be NULL. */
MIPT_API int mipt_render(MiptScene *scene, const MiptCamera *camera, const MiptOptions *opt,
                float *hdr_rgb, uint8_t *rgba8, MiptStats *stats);

/* Same, into DEVICE buffers (e.g. torch tensors), launched on `hip_stream` (hipStream_t, may
 * be NULL = the null stream).  Blocks until the kernel has finished (stats are read back).
 * With tile sharding and MIPT_FLAG_PACKED, d_hdr_rgb holds mipt_packed_pixels() * 3 floats. */
MIPT_API int mipt_render_device(MiptScene *scene, const MiptCamera *camera, const MiptOptions *opt,
                       float *d_hdr_rgb, uint8_t *d_rgba8, void *hip_stream, MiptStats *stats);

/* Tile-shard helpers (image tiles shard across GPUs; one RCCL all-gather of packed slices). */
MIPT_API uint64_t mipt_packed_pixels(uint32_t width, uint32_t height, uint32_t tile_world);
/* d_packed_all: tile_world slices of mipt_packed_pixels()*3 floats, rank-major (the layout an
 * all-gather produces); writes the full width*height*3 frame. */
MIPT_API int mipt_unpack_tiles(const float *d_packed_all, uint32_t width, uint32_t height,
                      uint32_t tile_world, float *d_hdr_rgb, void *hip_stream);
/* linear HDR -> sRGB -> RGBA8 epilogue on device (vec3.rs:80-90, 262-270; cpu.rs:61-64).
 * The radiance is first divided by `divisor` (the sample count for a summed buffer, cpu.rs:60;
 * 1 for a buffer that already holds the mean). */
MIPT_API int mipt_tonemap_device(const float *d_hdr_rgb, uint64_t n_pixels, float divisor,
                        uint8_t *d_rgba8, void *hip_stream);

/* The wgpu backend's post-process pass (pp_compute.wgsl:7-34): radiance / divisor, clamped to [0,1] (its accumulator is
 * rgba16unorm), linear_to_srgb, THEN aces_filmic, written as RGBA16 unorm (4 x u16 per pixel, alpha 65535) -- the pixel
 * format Renderer::render saves (renderer.rs:67-73, ColorType::Rgba16).  The CPU backend's epilogue is mipt_tonemap_device. */
MIPT_API int mipt_postprocess_device(const float *d_hdr_rgb, uint64_t n_pixels, float divisor, uint16_t *d_rgba16, void *hip_stream);

/* ---- all GPUs of one node behind one call --------------------------------------------- */

/* The reference's host is a single process with one Rc<RefCell<Scene>> (src/main.rs:46) and a blocking
 * `match backend` arm (src/renderer.rs:57-63), so its multi-GPU arm must be one call from one thread.
 * A MiptMulti holds a scene replica, a HIP stream and an RCCL communicator (ncclCommInitAll) per device. */
typedef struct MiptMulti MiptMulti;

enum MiptMultiMode {
    MIPT_MULTI_TILES   = 0,  /* 8x8 image tiles round-robin over the devices (pixel seeds, cpu.rs:28-29, make the frame
                              * bit-identical to the single-GPU frame); ONE ncclGather of rank-packed f32 slices to
                              * device 0 over xGMI + a de-interleave kernel.  BASELINE config 4. */
    MIPT_MULTI_SAMPLES = 1   /* every device renders all pixels for a disjoint sample range with the per-sample seeds of
                              * rt_compute.wgsl:102 (seed_mode is forced to MIPT_SEED_PER_SAMPLE: the pixel stream cannot be
                              * entered mid-way) into un-normalised sums; ONE ncclReduce(sum, f32) to device 0, then / samples.
                              * The f32 sum order differs from a sequential accumulation.  BASELINE config 5. */
};

typedef struct {
    MiptStats total;              /* counters summed over devices; kernel_ms = the slowest device's trace kernel */
    double    collective_ms;      /* gather/reduce + assemble (+ tonemap) on device 0's stream, HIP events */
    double    wall_ms;            /* the whole call on the host clock; mipt_render_multi: incl. the D2H copy of the outputs */
    double    device_kernel_ms[8];/* trace-kernel time of devices 0..7 */
    uint32_t  n_devices, reserved;
} MiptMultiStats;

/* device_ids: n_devices HIP device ordinals (NULL = 0..n_devices-1; n_devices 0 = every visible device).  The scene crosses PCIe
 * ONCE, to device_ids[0]; the other replicas are device-to-device copies over xGMI, all queued before the first is waited for
 * (pulls from one GPU use one link per destination).  Creates the communicators. */
MIPT_API int  mipt_multi_create(const MiptSceneDesc *desc, const int *device_ids, int n_devices, MiptMulti **out);
/* Same from the triangles alone (mipt_scene_create_from_triangles on device_ids[0], then the xGMI replicas). */
MIPT_API int  mipt_multi_create_from_triangles(const MiptSceneDesc *desc, const int *device_ids, int n_devices, MiptMulti **out);
/* The replica on device `index` (0 .. mipt_multi_device_count()-1), owned by `multi`: for mipt_scene_info / mipt_scene_get_bvh. */
MIPT_API MiptScene *mipt_multi_scene(MiptMulti *multi, int index);
MIPT_API void mipt_multi_destroy(MiptMulti *multi);
MIPT_API int  mipt_multi_device_count(const MiptMulti *multi);

/* Renders one frame on all devices of `multi` into HOST buffers (either may be NULL) and blocks until done; same outputs
 * as mipt_render.  `opt` describes the whole frame: tile_rank / tile_world / sample_begin and the PACKED / SUM / ACCUM
 * flags must be 0 (the call owns the sharding); MIPT_FLAG_COUNT is honoured. */
MIPT_API int  mipt_render_multi(MiptMulti *multi, const MiptCamera *camera, const MiptOptions *opt, uint32_t mode,
                       float *hdr_rgb, uint8_t *rgba8, MiptMultiStats *stats);

/* Same, but the frame stays in HBM: d_hdr_rgb (width*height*3 f32, required) and d_rgba8 (width*height*4 bytes, may be
 * NULL) are buffers in the memory of the ROOT device (mipt_multi_root_device(): device_ids[0]); the assemble kernels write
 * them directly and nothing crosses PCIe.  The multi-GPU counterpart of mipt_render_device. */
MIPT_API int  mipt_render_multi_device(MiptMulti *multi, const MiptCamera *camera, const MiptOptions *opt, uint32_t mode,
                              float *d_hdr_rgb, uint8_t *d_rgba8, MiptMultiStats *stats);
/* HIP ordinal of the device that gathers / reduces and holds the assembled frame, or a negative MiptStatus. */
MIPT_API int  mipt_multi_root_device(const MiptMulti *multi);
/* Trace-kernel stats of device `index` (0 .. mipt_multi_device_count()-1) in the last mipt_render_multi* call:
 * MiptMultiStats.total sums the counters, this is one device's share (what its one launch did). */
MIPT_API int  mipt_multi_device_stats(const MiptMulti *multi, int index, MiptStats *out);

/* ---- host-side restatements of the scene model that feeds the path ------------------- */

/* BVH::build (src/bvh.rs:13-161): binned SAH, 8 bins; reorders `tris` in place exactly as
 * bvh.rs:99-108 does and emits the identical node array.  nodes_cap >= 2*n_tris-1.
 * threads: 0 = hardware concurrency. */
MIPT_API int mipt_bvh_build(MiptTriangle *tris, uint32_t n_tris, MiptNode *nodes_out,
                   uint32_t nodes_cap, uint32_t *n_nodes_out, uint32_t threads);

/* Scene::load for Wavefront OBJ + MTL (src/scene.rs:22-85, src/loader/obj.rs:16-436): parses the
 * file, expands indexed faces into fat triangles and runs BVH::build.  The returned object owns
 * host arrays; mipt_obj_get fills a MiptSceneDesc that borrows them (valid until mipt_obj_free)
 * and, optionally, the material names in material-id order. */
typedef struct MiptObj MiptObj;
MIPT_API int  mipt_obj_load(const char *path, MiptObj **out);
/* The same without BVH::build (scene.rs:80): triangles in file order, no nodes (mipt_obj_get: nodes = NULL, n_nodes = 0) -- the input
 * of mipt_scene_create_from_triangles, which builds the tree on the GPU.  (A 10 M-triangle file: parsing takes seconds on 16 cores,
 * the host BVH build ten times that.) */
MIPT_API int  mipt_obj_load_triangles(const char *path, MiptObj **out);
MIPT_API int  mipt_obj_get(MiptObj *obj, MiptSceneDesc *desc_out, const char ***material_names_out);
MIPT_API void mipt_obj_free(MiptObj *obj);

/* Texture::load (src/texture.rs:13-31): decodes an image file (PNG, JPEG, TGA, BMP or binary PPM, chosen by extension like
 * image::open), flips it vertically and expands to RGBA8; hash_out (may be NULL) receives the djb2 hash the loader
 * de-duplicates textures by (texture.rs:40-48).  desc_out borrows the image's pixels until mipt_texture_free. */
typedef struct MiptImage MiptImage;
MIPT_API int  mipt_texture_load(const char *path, MiptImage **out, MiptTexture *desc_out, uint32_t *hash_out);
MIPT_API void mipt_texture_free(MiptImage *img);

/* The image output of Renderer::render (src/renderer.rs:66-83, image::save_buffer): writes width x height RGBA pixels,
 * top row first, as a PNG with 8 or 16 bits per sample (16-bit samples in host byte order; the reference saves
 * ColorType::Rgba16, and the bytes of its CPU arm are RGBA8 -- SURVEY T12).  Uncompressed deflate blocks. */
MIPT_API int mipt_image_save_png(const char *path, uint32_t width, uint32_t height, uint32_t bits_per_sample, const void *rgba);

/* The same build on the GPU (level-synchronous binned SAH with the partition's closed-form permutation); identical output
 * (sign of zero in a bound aside).  Uploads `tris`, downloads the reordered triangles and the nodes; build_ms_out (may be
 * NULL) receives the device time of the build itself without the transfers. */
MIPT_API int mipt_bvh_build_device(MiptTriangle *tris, uint32_t n_tris, MiptNode *nodes_out, uint32_t nodes_cap,
                          uint32_t *n_nodes_out, int device_id, double *build_ms_out);

/* Camera::update_view + Mat4f::look_at (src/scene.rs:181-194, src/math/mat4.rs:25-44). */
MIPT_API int mipt_camera_from_pose(const float position[3], float pitch_deg, float yaw_deg, MiptCamera *out);

/* Material::default() (src/scene.rs:148-167). */
MIPT_API void mipt_material_default(MiptMaterial *out);

MIPT_API const char *mipt_last_error(void);
MIPT_API int mipt_abi_version(void);
/* number of HIP devices visible, or a negative MiptStatus */
MIPT_API int mipt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */

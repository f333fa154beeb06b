/*
 * mipt_diag.h -- diagnostic probe library (libmipt_diag.so).  Test infrastructure for the GPU box; NOT part of the
 * drop-in boundary (include/mipt.h) and not exported by libmipt.so.
 *
 * Evaluates one device arithmetic primitive of the path-tracing kernel element-wise on the GPU, so tests can pin the
 * kernel's f32/f64 building blocks against the CPU oracle bit for bit.  The functions evaluated are the very ones the
 * kernel inlines (rust_ray_tracing_amd/csrc/pt_device_math.h): the restatement of glibc 2.35's cosf / log10f / powf
 * that stands in for Rust std f32::cos / f32::log10 / f32::powf (reference src/math.rs:15-19, src/math/vec3.rs:80-90).
 */
#ifndef MIPT_DIAG_H
#define MIPT_DIAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MIPT_DIAG_API __attribute__((visibility("default")))
#else
#define MIPT_DIAG_API
#endif

/* host buffers in/out.  op: 0 cosf, 1 log10f, 2 powf(a,b), 3 a/b, 4 sqrt(a), 5 a*b, 6 a+b, 7 min, 8 max,
 *     9 rand_f32(seed=bits(a)), 10 rand_f32_nd(seed), 11 rand_in_unit_sphere(seed)[b], 12 srgb+quantise(a)
 *     (result as integer bits), 13 fract(a), 14 the per-ray-reciprocal division a/b (valid on its checked range),
 *     15 u8 -> f32/255 for the integer whose bits are a, 16 sinf, 17 expf, 18 logf,
 *     19 the kernel's log10f specialised to rand_f32's range {0} u [2^-32, 1], 20 its cosf specialised to [0, 6.2831855].
 * Returns 0, or -1 (bad argument) / -2 (HIP error); device buffers are released on every path. */
MIPT_DIAG_API int mipt_debug_eval(int op, const float *a, const float *b, uint64_t n, float *out);

/* ops 0, 1, 2 (with second argument y), 16, 17, 18, 19, 20 on the n consecutive binary32 bit patterns first_bits, first_bits+1, ...
 * (first_bits + n <= 2^32): the exhaustive sweeps of tests/test_gpu_libm.py. */
MIPT_DIAG_API int mipt_debug_eval_range(int op, uint32_t first_bits, uint64_t n, float y, float *out);

MIPT_DIAG_API const char *mipt_diag_last_error(void);

/* The library-internal device-layout orders of libmipt.so (rust_ray_tracing_amd/csrc/bvh_build.cpp, hidden there), re-exported for
 * tests/test_host_layout.py: order of the 64-byte pair records in HBM (order_out[j] = reference pair index of record j, 0xffffffff = a
 * pad record), the number of breadth-first levels at its top, and the slot of every triangle's record in the intersection stream.
 * nodes: the reference's 32-byte Node array (MiptNode). */
MIPT_DIAG_API int mipt_internal_pair_order(const void *nodes, uint32_t n_nodes, uint32_t *order_out, uint32_t cap, uint32_t *n_records_out);
MIPT_DIAG_API uint32_t mipt_internal_pair_order_top(void);
MIPT_DIAG_API int mipt_internal_tri_slots(const void *nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t *slot_out, uint32_t *n_slots_out);

/* The device layout behind a MiptScene handle of libmipt.so (tests/cpp/scene_hooks.hip), for comparing the layout the GPU kernels
 * build (mipt_scene_create_from_triangles) with the host-built one (mipt_scene_create): sizes in bytes of [pair records | intersection
 * stream] and of the attribute stream; a copy of either (which = 0 / 1) to the host; an order-dependent 64-bit fingerprint of each. */
MIPT_DIAG_API int mipt_diag_scene_sizes(const void *scene, uint64_t out[2]);
MIPT_DIAG_API int mipt_diag_scene_read(const void *scene, int which, void *dst, uint64_t bytes);
MIPT_DIAG_API int mipt_diag_scene_hash(const void *scene, uint64_t out[2]);

/* The HOST layout of a scene's geometry (tests/cpp/host_layout.cpp): byte for byte the buffers rounds 1-3 built on host threads from the
 * reference's arrays -- geom = pair records | intersection stream, attr = attribute stream -- which the product now produces with GPU
 * kernels for both mipt_scene_create and mipt_scene_create_from_triangles; the tests compare mipt_diag_scene_read's bytes with these.
 * `desc` is a MiptSceneDesc with a well-formed tree and triangles in its order.  sizes_out[2] = bytes of geom and attr (call with
 * null buffers first); info_out[4] (may be NULL) = pair records incl. padding, largest leaf, the root's slot and triangle count. */
MIPT_DIAG_API int mipt_diag_host_layout(const void *desc, uint8_t *geom_out, uint64_t geom_cap, uint8_t *attr_out, uint64_t attr_cap,
                                        uint64_t *sizes_out, uint32_t *info_out);

/* mipt_diag_scene_hash's fingerprint of a word stream in HOST memory (for the buffers of mipt_diag_host_layout at sizes where a
 * byte compare is unwieldy). */
MIPT_DIAG_API int mipt_diag_hash_words(const void *words, uint64_t n_words, uint64_t *out);

/* Writes `n_tris` reference Triangles (112 B each) as an OBJ body (tests/cpp/obj_writer.cpp): per triangle 3 v, 3 vt, 3 vn lines and one
 * face line, shortest round-trip decimals; `mtllib` (may be NULL) names the material library, material_names[material_id] go into
 * usemtl lines.  0, -1 (bad argument) or -2 (I/O).  For rust_ray_tracing_amd/synth.py write_obj. */
MIPT_DIAG_API int mipt_diag_write_obj(const char *path, const void *tris, uint64_t n_tris, const char *mtllib, const char *const *material_names,
                                      uint32_t n_materials);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_DIAG_H */

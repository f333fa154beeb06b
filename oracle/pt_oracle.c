/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See pt_oracle.h for scope, provenance and the "parity unpinned" statement.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).  Rust never
 * contracts a*b+c into an FMA and rounds every f32 operation separately
 * (SURVEY.md T11); this file is written the same way: one rounded IEEE-754
 * binary32 operation per source-level operator, evaluated in the reference's
 * order.
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sys/mman.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* FMA contraction is disabled by -ffp-contract=off in oracle/Makefile (gcc ignores the STDC pragma) */

_Static_assert(sizeof(OrcVec3) == 12, "Vec3f");
_Static_assert(sizeof(OrcVertex) == 32, "Vertex");
_Static_assert(sizeof(OrcTriangle) == 112, "Triangle");
_Static_assert(__builtin_offsetof(OrcTriangle, material_id) == 96, "Triangle.material_id");
_Static_assert(sizeof(OrcNode) == 32, "Node");
_Static_assert(__builtin_offsetof(OrcNode, first_tri_or_child) == 12, "Node.first");
_Static_assert(__builtin_offsetof(OrcNode, bounds_max) == 16, "Node.max");
_Static_assert(__builtin_offsetof(OrcNode, num_tris) == 28, "Node.num");
_Static_assert(sizeof(OrcMaterial) == 80, "Material");
_Static_assert(__builtin_offsetof(OrcMaterial, ior) == 28, "Material.ior");
_Static_assert(__builtin_offsetof(OrcMaterial, emission) == 32, "Material.emission");
_Static_assert(__builtin_offsetof(OrcMaterial, base_color_tex_id) == 56, "Material.tex");
_Static_assert(sizeof(OrcCamera) == 80, "UniformCamera");
_Static_assert(__builtin_offsetof(OrcCamera, position) == 64, "UniformCamera.position");

#define ORC_MISS 1e30f                         /* ray.rs:79, :217 */
#define ORC_F32_MAX 3.40282346638528859812e+38f

/* ------------------------------------------------------------------------- */
/* Transcendentals.  The reference calls Rust std f32::{cos,log10,powf} = the */
/* platform libm (math.rs:16-18, vec3.rs:87), i.e. glibc on Linux.            */
/* ORC_LIBM_GLIBC235 (default): the restatement of glibc 2.35's algorithms in */
/* glibc_flt32.h (x86_64 FMA variants) -- the same spec the kernel carries in */
/* csrc/pt_device_math.h; tests/test_libm_pin.py proves it equal to this      */
/* machine's libm on every binary32 argument.  ORC_LIBM_HOST: call the libm   */
/* this process is linked to (what the Rust binary would do); kept so tests   */
/* can assert that whole renders are bit-identical either way.                */
/* ------------------------------------------------------------------------- */
#include "glibc_flt32.h"
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

float orc_glibc_cosf(float x) { return gl_cosf(x); }
float orc_glibc_sinf(float x) { return gl_sinf(x); }
float orc_glibc_logf(float x) { return gl_logf(x); }
float orc_glibc_log10f(float x) { return gl_log10f(x); }
float orc_glibc_expf(float x) { return gl_expf(x); }
float orc_glibc_powf(float x, float y) { return gl_powf(x, y); }
/* array forms for the sweep tests: op 0 cosf, 1 log10f, 2 powf(a, b), 16 sinf, 17 expf, 18 logf (the op numbers of
 * mipt_debug_eval); libm selects the restatement or the host libm */
void orc_eval_array(int op, int libm, const float *a, const float *b, uint64_t n, float *out) {
    for (uint64_t i = 0; i < n; i++) {
        const float x = a[i], y = b ? b[i] : 0.0f;
        float r = 0.0f;
        if (libm == ORC_LIBM_HOST) {
            switch (op) {
            case 0: r = cosf(x); break;
            case 1: r = log10f(x); break;
            case 2: r = powf(x, y); break;
            case 16: r = sinf(x); break;
            case 17: r = expf(x); break;
            case 18: r = logf(x); break;
            default: break;
            }
        } else {
            switch (op) {
            case 0: r = gl_cosf(x); break;
            case 1: r = gl_log10f(x); break;
            case 2: r = gl_powf(x, y); break;
            case 16: r = gl_sinf(x); break;
            case 17: r = gl_expf(x); break;
            case 18: r = gl_logf(x); break;
            default: break;
            }
        }
        out[i] = r;
    }
}

static inline float f_cos(float x, int libm)   { return libm == ORC_LIBM_HOST ? cosf(x) : gl_cosf(x); }
static inline float f_log10(float x, int libm) { return libm == ORC_LIBM_HOST ? log10f(x) : gl_log10f(x); }
static inline float f_pow(float x, float y, int libm) { return libm == ORC_LIBM_HOST ? powf(x, y) : gl_powf(x, y); }
static inline float f_sin(float x, int libm)   { return libm == ORC_LIBM_HOST ? sinf(x) : gl_sinf(x); }
static inline float f_exp(float x, int libm)   { return libm == ORC_LIBM_HOST ? expf(x) : gl_expf(x); }

/* ------------------------------------------------------------------------- */
/* math.rs / vec3.rs / vec2.rs / mat4.rs subset                               */
/* ------------------------------------------------------------------------- */
typedef OrcVec3 v3;
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* vec3.rs:272-278 */
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* vec3.rs:288-294 */
static inline v3 v_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }      /* vec3.rs:304-310 */
static inline v3 v_muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }        /* vec3.rs:312-318 */
static inline v3 v_div(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }      /* vec3.rs:336-342 */
static inline v3 v_divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }        /* vec3.rs:344-350 */
static inline float v_dot(v3 a, v3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); } /* vec3.rs:130-134 */
static inline v3 v_cross(v3 a, v3 b) {                                                  /* vec3.rs:136-144 */
    return V3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
static inline float v_length(v3 a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); } /* vec3.rs:93-97 */
static inline v3 v_normalized(v3 a) { return v_divs(a, v_length(a)); }                  /* vec3.rs:105-109 */
/* f32::min / f32::max are IEEE minNum / maxNum (NaN-ignoring) = C fminf / fmaxf (T5) */
static inline v3 v_min(v3 a, v3 b) { return V3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); } /* vec3.rs:146-154 */
static inline v3 v_max(v3 a, v3 b) { return V3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); } /* vec3.rs:155-163 */

uint32_t orc_xor_shift(uint32_t *state) {              /* math.rs:6-13 */
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return x;
}
float orc_rand_f32(uint32_t *state) {                  /* math.rs:22-24; u32::MAX as f32 == 2^32 (T1) */
    return (float)orc_xor_shift(state) / 4294967296.0f;
}
float orc_rand_f32_nd(uint32_t *state, int libm) {     /* math.rs:15-19 (log10, not ln: T3) */
    float theta = 6.283185f * orc_rand_f32(state);
    float rho = sqrtf(-2.0f * f_log10(orc_rand_f32(state), libm));
    return rho * f_cos(theta, libm);
}
void orc_rand_in_unit_sphere(uint32_t *state, int libm, float out[3]) { /* vec3.rs:66-68 */
    float x = orc_rand_f32_nd(state, libm);
    float y = orc_rand_f32_nd(state, libm);
    float z = orc_rand_f32_nd(state, libm);
    v3 n = v_normalized(V3(x, y, z));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
uint32_t orc_pixel_seed(uint32_t index) {              /* cpu.rs:28-29 */
    return 987612486u * (index + 87636354u);
}
uint32_t orc_sample_seed(uint32_t s, uint32_t x, uint32_t y) { /* rt_compute.wgsl:102 */
    return s * 6023u + (757283u * x + 872653746u * y);
}
static inline v3 mat_mul_v3(const float m[4][4], v3 r) { /* mat4.rs:143-152 (upper-left 3x3, data[col][row]) */
    float x = m[0][0] * r.x + m[1][0] * r.y + m[2][0] * r.z;
    float y = m[0][1] * r.x + m[1][1] * r.y + m[2][1] * r.z;
    float z = m[0][2] * r.x + m[1][2] * r.y + m[2][2] * r.z;
    return V3(x, y, z);
}

void orc_linear_to_srgb(const float in[3], int libm, float out[3]) { /* vec3.rs:80-90, mix :197-205 */
    for (int i = 0; i < 3; i++) {
        float c = in[i];
        float cutoff = (c < 0.0031308f) ? 1.0f : 0.0f;
        float higher = 1.055f * f_pow(c, 1.0f / 2.4f, libm) - 0.055f;
        float lower = c * 12.92f;
        out[i] = (higher * (1.0f - cutoff)) + lower * cutoff;
    }
}
void orc_quantize(const float in[3], uint8_t out[3]) { /* vec3.rs:262-270; NaN -> 0 (Rust `as u8`) */
    for (int i = 0; i < 3; i++) {
        float c = floorf(in[i] * 255.0f);
        if (c < 0.0f) c = 0.0f;                 /* f32::clamp keeps NaN; `as u8` then maps it to 0 */
        if (c > 255.0f) c = 255.0f;
        out[i] = (c != c) ? 0 : (uint8_t)c;
    }
}
/* pp_compute.wgsl:7-34 -- linear_to_srgb THEN aces_filmic on an rgba16unorm accumulator (clamped to [0,1]), stored
 * as unorm16 = floor(x*65535 + 0.5).  One rounded f32 op per WGSL operator. */
void orc_postprocess(const float *hdr, uint64_t n_pixels, float divisor, int libm, uint16_t *out) {
    for (uint64_t i = 0; i < n_pixels; i++) {
        for (int k = 0; k < 3; k++) {
            float v = hdr[i * 3 + k];
            if (divisor != 1.0f) v = v / divisor;
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            float lin[3] = {v, v, v}, srgb[3];
            orc_linear_to_srgb(lin, libm, srgb);
            const float x = srgb[0];
            const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
            float y = (x * (a * x + b)) / (x * (c * x + d) + e);
            y = fminf(fmaxf(y, 0.0f), 1.0f);
            out[i * 4 + k] = (uint16_t)floorf(y * 65535.0f + 0.5f);
        }
        out[i * 4 + 3] = 65535;
    }
}

static inline int32_t f32_as_i32(float f) {           /* Rust `as i32`: saturating, NaN -> 0 */
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static int tex_lookup(const OrcTexture *t, float u, float v, uint8_t out[4]) { /* texture.rs:33-38 */
    float fu = u - truncf(u), fv = v - truncf(v);      /* f32::fract */
    int32_t i = f32_as_i32(fu * (float)t->width);
    int32_t j = f32_as_i32(fv * (float)t->height);
    int64_t index = (int64_t)i + (int64_t)j * (int64_t)t->width;
    int64_t n = (int64_t)t->width * (int64_t)t->height;
    int clamped = 0;
    if (index < 0) { index = 0; clamped = 1; }         /* the reference would panic here (T10) */
    if (index >= n) { index = n - 1; clamped = 1; }
    memcpy(out, t->rgba8 + 4 * index, 4);
    return clamped;
}
void orc_texture_color_at(const OrcTexture *t, float u, float v, uint8_t out[4]) { tex_lookup(t, u, v, out); }

void orc_pixel_screen(uint32_t index, uint32_t w, uint32_t h, float out[2]) { /* cpu.rs:31-35 (T9) */
    uint32_t x = index % w;
    uint32_t y = h - (index / w);
    out[0] = ((((float)x / (float)w) * 2.0f) - 1.0f) * ((float)w / (float)h);
    out[1] = (((float)y / (float)h) * 2.0f) - 1.0f;
}

/* ------------------------------------------------------------------------- */
/* cpu/ray.rs                                                                  */
/* ------------------------------------------------------------------------- */
typedef struct { v3 origin, direction; } Ray;
typedef struct {                                      /* ray.rs:205-227 */
    int has_hit; v3 point, normal; float distance; float uvx, uvy;
    uint32_t material_id; int front_face; float u, v;
} Hit;

typedef struct {
    const OrcTriangle *tris; uint32_t n_tris;
    const OrcNode *nodes; uint32_t n_nodes;
    const OrcMaterial *materials; uint32_t n_materials;
    const OrcTexture *textures; uint32_t n_textures;
} SceneView;

static Hit intersect_tri(const Ray *ray, const OrcTriangle *tri) { /* ray.rs:19-67 */
    v3 v_1 = tri->vertices[0].position, v_2 = tri->vertices[1].position, v_3 = tri->vertices[2].position;
    v3 edge_1 = v_sub(v_2, v_1);
    v3 edge_2 = v_sub(v_3, v_1);
    v3 ray_cross_e2 = v_cross(ray->direction, edge_2);
    float det = v_dot(edge_1, ray_cross_e2);
    float inv_det = 1.0f / det;
    v3 s = v_sub(ray->origin, v_1);
    float u = inv_det * v_dot(s, ray_cross_e2);
    v3 s_cross_e1 = v_cross(s, edge_1);
    float v = inv_det * v_dot(ray->direction, s_cross_e1);
    float t = inv_det * v_dot(edge_2, s_cross_e1);
    int front_face = det > 0.0f;
    float w = 1.0f - u - v;
    v3 normal = v_add(v_add(v_muls(tri->vertices[0].normal, w), v_muls(tri->vertices[1].normal, u)),
                      v_muls(tri->vertices[2].normal, v));
    if (!front_face) normal = V3(-normal.x, -normal.y, -normal.z);
    Hit h;
    h.uvx = ((tri->vertices[0].tex_coord_x * w) + (tri->vertices[1].tex_coord_x * u)) + (tri->vertices[2].tex_coord_x * v);
    h.uvy = ((tri->vertices[0].tex_coord_y * w) + (tri->vertices[1].tex_coord_y * u)) + (tri->vertices[2].tex_coord_y * v);
    /* exact boolean form of ray.rs:56-59 -- NaN u/v pass, NaN t fails (T4) */
    h.has_hit = (t > 0.0f) && !(det < 0.0f && det > -0.0f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f);
    h.point = v_add(ray->origin, v_muls(ray->direction, t));
    h.normal = normal;
    h.distance = t;
    h.material_id = tri->material_id;
    h.front_face = front_face;
    h.u = u; h.v = v;
    return h;
}

static inline float intersect_node(const Ray *ray, const OrcNode *node, int cull, float max_distance) {
    /* ray.rs:69-81; cull=1 adds rt_compute.wgsl:348's `t_near < max_distance`
     * (the caller passes max_distance = best * (1 + cull_margin); margin 0 = the WGSL rule) */
    v3 t_min = v_div(v_sub(node->bounds_min, ray->origin), ray->direction);
    v3 t_max = v_div(v_sub(node->bounds_max, ray->origin), ray->direction);
    v3 t_1 = v_min(t_min, t_max);
    v3 t_2 = v_max(t_min, t_max);
    float t_near = fmaxf(fmaxf(t_1.x, t_1.y), t_1.z);
    float t_far = fminf(fminf(t_2.x, t_2.y), t_2.z);
    if (t_near <= t_far && t_far > 0.0f && (!cull || t_near < max_distance)) return t_near;
    return ORC_MISS;
}

typedef struct { OrcStats s; uint32_t stack_cap; int cull; float cull_scale; int libm; float *rec; uint32_t rec_cap, rec_n; uint32_t cur_tri;
                 uint32_t *vlog; uint64_t vlog_cap, vlog_n; } Ctx;   /* vlog: record-visit trace (orc_visit_log) */
#define VLOG(cx, word) do { if ((cx)->vlog) { if ((cx)->vlog_n < (cx)->vlog_cap) (cx)->vlog[(cx)->vlog_n] = (word); (cx)->vlog_n++; } } while (0)

#define ORC_STACK_MAX 256
static void traverse_bvh(const Ray *ray, const SceneView *sc, Hit *hit, Ctx *cx) { /* ray.rs:84-139 */
    uint32_t stack[ORC_STACK_MAX];                   /* node indices; the reference stacks Node copies */
    uint32_t sp = 0;
    const OrcNode *node = &sc->nodes[0];
    cx->s.rays++;
    VLOG(cx, 0xffffffffu);
    for (;;) {
        if (node->num_tris > 0) {
            for (uint32_t i = 0; i < node->num_tris; i++) {
                Hit th = intersect_tri(ray, &sc->tris[node->first_tri_or_child + i]);
                cx->s.tri_tests++;
                VLOG(cx, 0x80000000u | (node->first_tri_or_child + i));
                if (th.has_hit && th.distance < hit->distance) { *hit = th; cx->cur_tri = node->first_tri_or_child + i; }
            }
            if (sp == 0) break;
            node = &sc->nodes[stack[--sp]];
            continue;
        }
        uint32_t c1 = node->first_tri_or_child, c2 = c1 + 1;
        cx->s.inner_steps++;
        VLOG(cx, (c1 - 1u) / 2u);
        const float max_d = hit->distance * cx->cull_scale;
        float dist_1 = intersect_node(ray, &sc->nodes[c1], cx->cull, max_d);
        float dist_2 = intersect_node(ray, &sc->nodes[c2], cx->cull, max_d);
        if (dist_1 > dist_2) {
            float td = dist_1; dist_1 = dist_2; dist_2 = td;
            uint32_t tc = c1; c1 = c2; c2 = tc;
        }
        if (dist_1 == ORC_MISS) {
            if (sp == 0) break;
            node = &sc->nodes[stack[--sp]];
        } else {
            node = &sc->nodes[c1];
            if (dist_2 < ORC_MISS) {
                if (sp < cx->stack_cap) {
                    stack[sp++] = c2;
                    if (sp > cx->s.max_stack) cx->s.max_stack = sp;
                } else {
                    cx->s.stack_overflows++;         /* the reference panics here (ray.rs:85,134) */
                }
            }
        }
    }
}

static v3 trace(Ray *ray, uint32_t max_bounces, const SceneView *sc, uint32_t *rng, Ctx *cx) { /* ray.rs:141-202 */
    v3 ray_color = V3(1.0f, 1.0f, 1.0f);
    v3 incoming_light = V3(0.0f, 0.0f, 0.0f);
    v3 emitted_light = V3(0.0f, 0.0f, 0.0f);
    uint32_t curr_bounces = 0;
    while (curr_bounces < max_bounces) {
        Hit hit;
        memset(&hit, 0, sizeof hit);
        hit.distance = ORC_MISS;
        cx->cur_tri = UINT32_MAX;
        traverse_bvh(ray, sc, &hit, cx);
        if (cx->rec && cx->rec_n < cx->rec_cap) {     /* debug trace: o, d, tri, t (orc_debug_pixel) */
            float *r = cx->rec + 8 * (size_t)cx->rec_n++;
            r[0] = ray->origin.x; r[1] = ray->origin.y; r[2] = ray->origin.z;
            r[3] = ray->direction.x; r[4] = ray->direction.y; r[5] = ray->direction.z;
            memcpy(&r[6], &cx->cur_tri, 4); r[7] = hit.distance;
        }
        if (hit.has_hit) {
            cx->s.hits++;
            VLOG(cx, 0xc0000000u | cx->cur_tri);
            const OrcMaterial *m = &sc->materials[hit.material_id];   /* ray.rs:153-154 */
            /* ray.rs:155-160: `ior` is computed and never used on the CPU path (T8) */
            if (m->base_color_tex_id != UINT32_MAX) {                  /* ray.rs:162-169 */
                uint8_t px[4];
                cx->s.tex_clamped += tex_lookup(&sc->textures[m->base_color_tex_id], hit.uvx, hit.uvy, px);
                cx->s.texel_fetches++;
                ray_color = v_mul(ray_color, V3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f));
            } else {
                ray_color = v_mul(ray_color, m->base_color);
            }
            if (m->emission_tex_id != UINT32_MAX) {                    /* ray.rs:170-176 */
                uint8_t px[4];
                cx->s.tex_clamped += tex_lookup(&sc->textures[m->emission_tex_id], hit.uvx, hit.uvy, px);
                cx->s.texel_fetches++;
                emitted_light = v_add(emitted_light, V3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f));
            } else {
                emitted_light = v_add(emitted_light, m->emission);
            }
            incoming_light = v_add(incoming_light, v_mul(emitted_light, ray_color)); /* ray.rs:177 */
            float rs[3];
            orc_rand_in_unit_sphere(rng, cx->libm, rs);
            v3 new_dir = v_normalized(v_add(hit.normal, V3(rs[0], rs[1], rs[2])));   /* ray.rs:179-180 (T7) */
            ray->origin = v_add(hit.point, v_muls(new_dir, 0.0001f));                 /* ray.rs:181 */
            ray->direction = new_dir;
            curr_bounces += 1;
        } else {
            ray_color = v_mul(ray_color, V3(1.0f, 1.0f, 1.0f));       /* ray.rs:184-193 */
            emitted_light = v_add(emitted_light, V3(1.0f, 1.0f, 1.0f));
            incoming_light = v_add(incoming_light, v_mul(emitted_light, ray_color));
            break;
        }
    }
    if (curr_bounces == 0) return incoming_light;                      /* ray.rs:197-201 */
    return v_divs(incoming_light, (float)curr_bounces);
}

/* ------------------------------------------------------------------------- */
/* rt_compute.wgsl restatement -- the wgpu backend's material model            */
/* (shading mode 1).  The reference has no CPU oracle for  */
/* it and WGSL leaves much to the implementation (FMA fusion, the precision of */
/* pow/exp/sin/cos/inverseSqrt, the bilinear filter's weights): this is ONE    */
/* deterministic reading -- one rounded f32 op per operator, the glibc restatement for the  */
/* transcendentals, exact f32 bilinear weights -- shared with the kernel.      */
/* "parity unpinned" against a real GPU run.                                   */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z, w; } v4;
static inline v3 w_normalize(v3 a) { return v_divs(a, v_length(a)); }          /* normalize(): v / length(v) */
static inline float w_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

static v4 sample_texture(const OrcTexture *t, float u, float v, Ctx *cx) { /* rt_compute.wgsl:499-501; sampler gpu.rs:393-401: linear, repeat */
    const int64_t W = t->width, H = t->height;
    float uu = u * (float)W - 0.5f, vv = v * (float)H - 0.5f;
    float fu = floorf(uu), fv = floorf(vv);
    float a = uu - fu, b = vv - fv;
    int64_t i0 = (fabsf(fu) < 1e9f) ? (int64_t)fu : 0, j0 = (fabsf(fv) < 1e9f) ? (int64_t)fv : 0;
    if (!(a == a)) a = 0.0f;
    if (!(b == b)) b = 0.0f;
    int64_t i1 = i0 + 1, j1 = j0 + 1;
    i0 = ((i0 % W) + W) % W; i1 = ((i1 % W) + W) % W; j0 = ((j0 % H) + H) % H; j1 = ((j1 % H) + H) % H;   /* AddressMode::Repeat */
    const uint8_t *p00 = t->rgba8 + 4 * (i0 + j0 * W), *p10 = t->rgba8 + 4 * (i1 + j0 * W);
    const uint8_t *p01 = t->rgba8 + 4 * (i0 + j1 * W), *p11 = t->rgba8 + 4 * (i1 + j1 * W);
    float out[4];
    for (int c = 0; c < 4; c++) {
        float t00 = (float)p00[c] / 255.0f, t10 = (float)p10[c] / 255.0f, t01 = (float)p01[c] / 255.0f, t11 = (float)p11[c] / 255.0f;
        float top = t00 * (1.0f - a) + t10 * a;
        float bot = t01 * (1.0f - a) + t11 * a;
        out[c] = top * (1.0f - b) + bot * b;
    }
    cx->s.texel_fetches++;
    v4 r = {out[0], out[1], out[2], out[3]};
    return r;
}

static void build_orthonormal_basis(v3 n, v3 *tangent, v3 *bitangent) {       /* rt_compute.wgsl:565-569 */
    v3 up = (fabsf(n.z) < 0.9999999f) ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
    *tangent = w_normalize(v_cross(up, n));
    *bitangent = v_cross(n, *tangent);
}
static inline v3 to_world(v3 t, v3 b, v3 n, v3 l) {                              /* tbn * local, :561-563 */
    return V3((t.x * l.x + b.x * l.y) + n.x * l.z, (t.y * l.x + b.y * l.y) + n.y * l.z, (t.z * l.x + b.z * l.y) + n.z * l.z);
}
static inline v3 to_local(v3 t, v3 b, v3 n, v3 w) { return V3(v_dot(t, w), v_dot(b, w), v_dot(n, w)); } /* transpose(tbn) * world */

static v3 sample_ggx_vndf(v3 ve, float ax, float ay, uint32_t *rng, int libm) {  /* rt_compute.wgsl:503-525 */
    float u1 = orc_rand_f32(rng), u2 = orc_rand_f32(rng);
    v3 Vh = w_normalize(V3(ax * ve.x, ay * ve.y, ve.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1 = V3(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) { float inv = 1.0f / sqrtf(lensq); T1 = V3(-Vh.y * inv, Vh.x * inv, 0.0f * inv); }
    v3 T2 = v_cross(Vh, T1);
    float r = sqrtf(u1);
    float phi = 2.0f * 3.1415926535f * u2;
    float t1 = r * f_cos(phi, libm);
    float t2 = r * f_sin(phi, libm);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    float k = sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
    v3 Nh = v_add(v_add(v_muls(T1, t1), v_muls(T2, t2)), v_muls(Vh, k));
    return w_normalize(V3(ax * Nh.x, ay * Nh.y, fmaxf(0.0f, Nh.z)));
}
static v3 cosine_sample_hemisphere(uint32_t *rng, int libm) {                    /* rt_compute.wgsl:527-551 */
    float ux = orc_rand_f32(rng), uy = orc_rand_f32(rng);
    float ox = 2.0f * ux - 1.0f, oy = 2.0f * uy - 1.0f;
    float dx, dy;
    if (ox == 0.0f && oy == 0.0f) { dx = 0.0f; dy = 0.0f; }
    else {
        float theta, r;
        if (fabsf(ox) > fabsf(oy)) { r = ox; theta = 0.7853981634f * (oy / ox); }
        else { r = oy; theta = 1.5707963268f - 0.7853981634f * (ox / oy); }
        dx = r * f_cos(theta, libm); dy = r * f_sin(theta, libm);
    }
    float z = sqrtf(fmaxf(0.0f, 1.0f - dx * dx - dy * dy));
    return V3(dx, dy, z);
}

static v3 trace_wgsl(Ray *ray, uint32_t max_depth, const SceneView *sc, uint32_t *rng, Ctx *cx) { /* rt_compute.wgsl:126-229 */
    const float EPSILON = 0.0001f;
    const int libm = cx->libm;
    v3 ray_color = V3(1.0f, 1.0f, 1.0f), incoming = V3(0.0f, 0.0f, 0.0f);
    v3 prev_hit_point = ray->origin;
    uint32_t depth = 0;
    while (depth < max_depth) {
        Hit hit; memset(&hit, 0, sizeof hit); hit.distance = ORC_MISS;
        cx->cur_tri = UINT32_MAX;
        traverse_bvh(ray, sc, &hit, cx);
        if (!hit.has_hit) {                                                      /* :215-223 */
            ray_color = v_mul(ray_color, V3(1.0f, 1.0f, 1.0f));
            incoming = v_add(incoming, v_mul(V3(1.0f, 1.0f, 1.0f), ray_color));
            break;
        }
        cx->s.hits++;
        depth += 1;
        /* intersect_tri differences (rt_compute.wgsl:318,328): point = fma(dir, t, origin); the normal is normalised */
        v3 point = V3(fmaf(ray->direction.x, hit.distance, ray->origin.x), fmaf(ray->direction.y, hit.distance, ray->origin.y),
                      fmaf(ray->direction.z, hit.distance, ray->origin.z));
        v3 normal = w_normalize(hit.normal);
        OrcMaterial m = sc->materials[hit.material_id];
        /* set_surface_properties, :251-294 */
        if (hit.front_face) m.ior = 1.0f / m.ior;
        if (m.base_color_tex_id != UINT32_MAX) {
            v4 t = sample_texture(&sc->textures[m.base_color_tex_id], hit.uvx, hit.uvy, cx);
            m.base_color = V3(f_pow(t.x, 2.2f, libm), f_pow(t.y, 2.2f, libm), f_pow(t.z, 2.2f, libm));
        }
        if (m.transparency_tex_id != UINT32_MAX) m.transparency = sample_texture(&sc->textures[m.transparency_tex_id], hit.uvx, hit.uvy, cx).w;
        if (m.roughness_tex_id != UINT32_MAX) m.roughness = sample_texture(&sc->textures[m.roughness_tex_id], hit.uvx, hit.uvy, cx).y;
        if (m.metallic_tex_id != UINT32_MAX) m.metallic = sample_texture(&sc->textures[m.metallic_tex_id], hit.uvx, hit.uvy, cx).z;
        if (m.emission_tex_id != UINT32_MAX) {
            v4 t = sample_texture(&sc->textures[m.emission_tex_id], hit.uvx, hit.uvy, cx);
            m.emission = V3(f_pow(t.x, 2.2f, libm), f_pow(t.y, 2.2f, libm), f_pow(t.z, 2.2f, libm));
        }
        v3 tangent, bitangent;
        build_orthonormal_basis(normal, &tangent, &bitangent);
        v3 tbn_n = normal;
        if (m.normal_tex_id != UINT32_MAX) {
            v4 t = sample_texture(&sc->textures[m.normal_tex_id], hit.uvx, hit.uvy, cx);
            normal = w_normalize(to_world(tangent, bitangent, tbn_n, V3(t.x * 2.0f - 1.0f, t.y * 2.0f - 1.0f, t.z * 2.0f - 1.0f)));
            build_orthonormal_basis(normal, &tangent, &bitangent);
            tbn_n = normal;
        }
        float transmitted_distance = hit.distance;                               /* :143-148 */
        if (hit.front_face) prev_hit_point = point;
        else transmitted_distance = v_length(v_sub(point, prev_hit_point));
        if (m.transparency < orc_rand_f32(rng)) {                                /* alpha cut-out, :150-153 */
            ray->origin = v_add(point, v_muls(ray->direction, EPSILON));
            continue;
        }
        float alpha = w_clamp(m.roughness * m.roughness, EPSILON, 1.0f);
        v3 neg_dir = V3(-ray->direction.x, -ray->direction.y, -ray->direction.z);
        v3 sampled_normal = to_world(tangent, bitangent, tbn_n, sample_ggx_vndf(to_local(tangent, bitangent, tbn_n, neg_dir), alpha, alpha, rng, libm));
        /* pow(x, 2) and pow(x, 5) have integer literal exponents: read as repeated multiplication (what shader compilers emit;
         * exp2(y*log2(x)) would be NaN for the negative base 1 - ior on back faces, which no render of the reference shows) */
        float f0s = ((1.0f - m.ior) * (1.0f - m.ior)) / ((1.0f + m.ior) * (1.0f + m.ior));
        v3 f0 = V3(f0s * (1.0f - m.metallic) + m.base_color.x * m.metallic, f0s * (1.0f - m.metallic) + m.base_color.y * m.metallic,
                   f0s * (1.0f - m.metallic) + m.base_color.z * m.metallic);                   /* mix(f0, base_color, metallic) */
        float p1 = 1.0f - v_dot(sampled_normal, neg_dir), p2 = p1 * p1;
        float p5 = (p2 * p2) * p1;                                                             /* schlick_fresnel, :553-555 */
        v3 fresnel = V3(f0.x + (1.0f - f0.x) * p5, f0.y + (1.0f - f0.y) * p5, f0.z + (1.0f - f0.z) * p5);
        float two_ndi = 2.0f * v_dot(sampled_normal, ray->direction);                          /* reflect(I, N) = I - 2 dot(N, I) N */
        v3 specular_dir = w_normalize(v_sub(ray->direction, v_muls(sampled_normal, two_ndi)));
        v3 transmitted_dir;                                                                    /* refract(I, N, eta) */
        {
            float ndi = v_dot(sampled_normal, ray->direction);
            float k = 1.0f - m.ior * m.ior * (1.0f - ndi * ndi);
            v3 r = (k < 0.0f) ? V3(0.0f, 0.0f, 0.0f)
                              : v_sub(v_muls(ray->direction, m.ior), v_muls(sampled_normal, m.ior * ndi + sqrtf(k)));
            transmitted_dir = w_normalize(r);
        }
        v3 diffuse_dir = w_normalize(to_world(tangent, bitangent, tbn_n, cosine_sample_hemisphere(rng, libm)));
        /* select_bsdf, :231-248 */
        int specular = 0, transmitted = 0;
        {
            float r = orc_rand_f32(rng);
            if (m.metallic > r) specular = 1;
            else if (m.metallic + m.transmission > r) transmitted = 1;
        }
        v3 new_dir;
        float fl = v_length(fresnel);
        float r2 = orc_rand_f32(rng);
        if (fl < r2 && !specular) {                                                            /* :167-186 */
            ray_color = v_mul(ray_color, m.base_color);
            if (transmitted) {
                new_dir = transmitted_dir;
                if (v_dot(new_dir, normal) > 0.0f) break;
                v3 absorption = V3(1.0f, 1.0f, 1.0f);
                if (!hit.front_face)
                    absorption = V3(f_exp(-(1.0f - m.base_color.x) * transmitted_distance, libm), f_exp(-(1.0f - m.base_color.y) * transmitted_distance, libm),
                                    f_exp(-(1.0f - m.base_color.z) * transmitted_distance, libm));
                ray_color = v_mul(ray_color, absorption);
            } else {
                new_dir = diffuse_dir;
            }
        } else {                                                                               /* :187-196 */
            if (specular) ray_color = v_mul(ray_color, fresnel);
            new_dir = specular_dir;
            if (v_dot(new_dir, normal) < 0.0f) break;
        }
        float rr = 1.0f;                                                                       /* Russian roulette, :198-207 */
        if (depth >= 4) {
            rr = fmaxf(ray_color.x, fmaxf(ray_color.z, ray_color.y));
            if (rr < orc_rand_f32(rng)) break;
        }
        ray_color = v_divs(ray_color, rr);
        incoming = v_add(incoming, v_mul(m.emission, ray_color));                              /* :209 */
        ray->origin = v_add(point, v_muls(new_dir, EPSILON));
        ray->direction = new_dir;
    }
    if (depth == 0) return incoming;
    return v_divs(incoming, (float)depth);
}

float orc_intersect_node(const float o[3], const float d[3], const OrcNode *n) {
    Ray r = {V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    return intersect_node(&r, n, 0, ORC_MISS);
}
void orc_intersect_tri(const float o[3], const float d[3], const OrcTriangle *t, float out[13]) {
    Ray r = {V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    Hit h = intersect_tri(&r, t);
    out[0] = (float)h.has_hit; out[1] = h.distance; out[2] = h.u; out[3] = h.v; out[4] = (float)h.front_face;
    out[5] = h.normal.x; out[6] = h.normal.y; out[7] = h.normal.z; out[8] = h.uvx; out[9] = h.uvy;
    out[10] = h.point.x; out[11] = h.point.y; out[12] = h.point.z;
}
void orc_trace_ray(const OrcTriangle *tris, uint32_t n_tris, const OrcNode *nodes, uint32_t n_nodes,
                   const OrcMaterial *materials, uint32_t n_materials,
                   const OrcTexture *textures, uint32_t n_textures,
                   const float o[3], const float d[3], uint32_t max_depth,
                   uint32_t *rng, int cull, int libm, float out[3]) {
    SceneView sc = {tris, n_tris, nodes, n_nodes, materials, n_materials, textures, n_textures};
    Ctx cx; memset(&cx, 0, sizeof cx);
    cx.stack_cap = 64; cx.cull = cull; cx.cull_scale = 1.0f; cx.libm = libm;
    Ray r = {V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    v3 c = trace(&r, max_depth, &sc, rng, &cx);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/* ------------------------------------------------------------------------- */
/* cpu.rs:13-68 -- the pixel loop                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
    SceneView sc; const OrcCamera *cam; OrcOptions opt;
    float *hdr; uint8_t *rgba8;
    uint64_t n_items, block; volatile uint64_t next_block;
    pthread_mutex_t mu; OrcStats total;
} Job;

static void render_pixel(Job *job, uint64_t index, Ctx *cx) {
    const OrcOptions *o = &job->opt;
    uint32_t w = o->width, h = o->height;
    uint32_t rng = orc_pixel_seed((uint32_t)index);                    /* cpu.rs:28-29 */
    v3 final_color = V3(0.0f, 0.0f, 0.0f);
    uint32_t x = (uint32_t)(index % w);
    uint32_t y = h - (uint32_t)(index / w);                            /* cpu.rs:31-32 (T9) */
    float screen_x = ((((float)x / (float)w) * 2.0f) - 1.0f) * ((float)w / (float)h);
    float screen_y = (((float)y / (float)h) * 2.0f) - 1.0f;
    uint32_t s0 = o->sample_begin ? o->sample_begin : 1u;
    for (uint32_t s = 0; s < o->samples; s++) {                        /* cpu.rs:37 */
        if (o->seed_mode == ORC_SEED_PER_SAMPLE || o->shading == 1)
            rng = orc_sample_seed(s0 + s, x, (uint32_t)(index / w));   /* rt_compute.wgsl:102, global_id.y = row */
        float jx = (orc_rand_f32(&rng) * 2.0f - 1.0f) * 0.0005f;       /* cpu.rs:38-42 */
        float jy = (orc_rand_f32(&rng) * 2.0f - 1.0f) * 0.0005f;
        v3 dir = v_normalized(mat_mul_v3(job->cam->look_at, V3(-screen_x + jx, screen_y + jy, 1.0f))); /* cpu.rs:43-45 */
        Ray ray = {job->cam->position, dir};                           /* cpu.rs:46-50 */
        final_color = v_add(final_color, o->shading == 1 ? trace_wgsl(&ray, o->max_ray_depth, &job->sc, &rng, cx)   /* rt_compute.wgsl:117 */
                                                        : trace(&ray, o->max_ray_depth, &job->sc, &rng, cx)); /* cpu.rs:52-57 */
    }
    if (!o->sum_only) final_color = v_divs(final_color, (float)o->samples); /* cpu.rs:60 */
    if (job->hdr) {
        job->hdr[3 * index + 0] = final_color.x;
        job->hdr[3 * index + 1] = final_color.y;
        job->hdr[3 * index + 2] = final_color.z;
    }
    if (job->rgba8) {
        float lin[3] = {final_color.x, final_color.y, final_color.z}, srgb[3];
        uint8_t q[3];
        orc_linear_to_srgb(lin, cx->libm, srgb);                       /* cpu.rs:61 */
        orc_quantize(srgb, q);                                         /* cpu.rs:63 */
        uint8_t *p = job->rgba8 + 4 * index;
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2]; p[3] = 255;             /* cpu.rs:64 */
    }
}

static void *worker(void *arg) {
    Job *job = (Job *)arg;
    Ctx cx; memset(&cx, 0, sizeof cx);
    cx.stack_cap = job->opt.stack_cap ? job->opt.stack_cap : 64;
    if (cx.stack_cap > ORC_STACK_MAX) cx.stack_cap = ORC_STACK_MAX;
    cx.cull = (int)job->opt.cull; cx.libm = (int)job->opt.libm;
    cx.cull_scale = 1.0f + job->opt.cull_margin;
    uint64_t stride = job->opt.pix_stride ? job->opt.pix_stride : 1;
    uint32_t blk_n = 0; double blk_sum = 0.0, blk_max = 0.0;
    for (;;) {
        /* rayon's by_uniform_blocks (cpu.rs:22-26): contiguous blocks of w*h/T pixel indices */
        uint64_t b = __atomic_fetch_add(&job->next_block, 1, __ATOMIC_RELAXED);
        uint64_t lo = b * job->block;
        if (lo >= job->n_items) break;
        uint64_t hi = lo + job->block; if (hi > job->n_items) hi = job->n_items;
        struct timespec b0, b1;
        clock_gettime(CLOCK_MONOTONIC, &b0);
        for (uint64_t i = lo; i < hi; i++) render_pixel(job, job->opt.pix_begin + i * stride, &cx);
        clock_gettime(CLOCK_MONOTONIC, &b1);
        const double bs = (double)(b1.tv_sec - b0.tv_sec) + 1e-9 * (double)(b1.tv_nsec - b0.tv_nsec);
        blk_n++; blk_sum += bs; if (bs > blk_max) blk_max = bs;
    }
    pthread_mutex_lock(&job->mu);
    job->total.n_blocks += blk_n; job->total.block_sec_mean += blk_sum;      /* sum here, divided in orc_render */
    if (blk_max > job->total.block_sec_max) job->total.block_sec_max = blk_max;
    job->total.rays += cx.s.rays; job->total.inner_steps += cx.s.inner_steps;
    job->total.tri_tests += cx.s.tri_tests; job->total.hits += cx.s.hits;
    job->total.texel_fetches += cx.s.texel_fetches; job->total.stack_overflows += cx.s.stack_overflows;
    job->total.tex_clamped += cx.s.tex_clamped;
    if (cx.s.max_stack > job->total.max_stack) job->total.max_stack = cx.s.max_stack;
    pthread_mutex_unlock(&job->mu);
    return NULL;
}

/* Debug: render ONE pixel (all its samples, in sequence) and record every ray as 8 floats
 * {o.xyz, d.xyz, bits(tri index or 0xffffffff), t}.  Returns the number of rays recorded. */
uint32_t orc_debug_pixel(const OrcTriangle *tris, uint32_t n_tris, const OrcNode *nodes, uint32_t n_nodes,
                         const OrcMaterial *materials, uint32_t n_materials,
                         const OrcTexture *textures, uint32_t n_textures,
                         const OrcCamera *camera, const OrcOptions *opt, uint64_t pixel_index,
                         float *records, uint32_t rec_cap, float out_rgb[3]) {
    Job job; memset(&job, 0, sizeof job);
    job.sc.tris = tris; job.sc.n_tris = n_tris; job.sc.nodes = nodes; job.sc.n_nodes = n_nodes;
    job.sc.materials = materials; job.sc.n_materials = n_materials;
    job.sc.textures = textures; job.sc.n_textures = n_textures;
    job.cam = camera; job.opt = *opt;
    float *hdr = (float *)calloc((size_t)opt->width * opt->height * 3, sizeof(float));
    job.hdr = hdr;
    Ctx cx; memset(&cx, 0, sizeof cx);
    cx.stack_cap = opt->stack_cap ? opt->stack_cap : 64;
    cx.cull = (int)opt->cull; cx.libm = (int)opt->libm; cx.cull_scale = 1.0f + opt->cull_margin;
    cx.rec = records; cx.rec_cap = rec_cap;
    render_pixel(&job, pixel_index, &cx);
    if (out_rgb) { out_rgb[0] = hdr[3 * pixel_index]; out_rgb[1] = hdr[3 * pixel_index + 1]; out_rgb[2] = hdr[3 * pixel_index + 2]; }
    free(hdr);
    return cx.rec_n;
}

/* Analysis aid for the device record layout (tests/tools/layout_model.py): renders pixels pix_begin, pix_begin + stride, ... (n_pixels
 * of them, all samples and bounces, shading mode 0) on one thread and logs which BVH records every ray touches, in order:
 * 0xffffffff = a new ray; k < 2^30 = inner step on child pair k (nodes 2k+1, 2k+2); 0x80000000|i = triangle test i;
 * 0xc0000000|i = attribute fetch of the winning triangle.  Returns the number of words the full log needs (may exceed cap). */
uint64_t orc_visit_log(const OrcTriangle *tris, uint32_t n_tris, const OrcNode *nodes, uint32_t n_nodes,
                       const OrcMaterial *materials, uint32_t n_materials, const OrcTexture *textures, uint32_t n_textures,
                       const OrcCamera *camera, const OrcOptions *opt, uint64_t pix_begin, uint64_t pix_stride, uint64_t n_pixels,
                       uint32_t *log, uint64_t cap) {
    Job job; memset(&job, 0, sizeof job);
    job.sc.tris = tris; job.sc.n_tris = n_tris; job.sc.nodes = nodes; job.sc.n_nodes = n_nodes;
    job.sc.materials = materials; job.sc.n_materials = n_materials;
    job.sc.textures = textures; job.sc.n_textures = n_textures;
    job.cam = camera; job.opt = *opt;
    Ctx cx; memset(&cx, 0, sizeof cx);
    cx.stack_cap = opt->stack_cap ? opt->stack_cap : 64;
    cx.cull = (int)opt->cull; cx.libm = (int)opt->libm; cx.cull_scale = 1.0f + opt->cull_margin;
    cx.vlog = log; cx.vlog_cap = cap;
    const uint64_t n_all = (uint64_t)opt->width * opt->height;
    for (uint64_t i = 0; i < n_pixels; i++) {
        const uint64_t index = pix_begin + i * pix_stride;
        if (index >= n_all) break;
        render_pixel(&job, index, &cx);
    }
    return cx.vlog_n;
}

typedef struct { const char *src_a; char *dst_a; size_t n_a; const char *src_b; char *dst_b; size_t n_b; uint32_t i, T; } SpreadJob;
static void *spread_worker(void *arg) {       /* copies share i of T of both arrays (page-aligned cuts): first touch = this thread */
    const SpreadJob *j = (const SpreadJob *)arg;
    const size_t pg = 4096;
    size_t lo = (j->n_a / j->T * j->i) / pg * pg, hi = (j->i + 1 == j->T) ? j->n_a : (j->n_a / j->T * (j->i + 1)) / pg * pg;
    if (hi > lo) memcpy(j->dst_a + lo, j->src_a + lo, hi - lo);
    lo = (j->n_b / j->T * j->i) / pg * pg; hi = (j->i + 1 == j->T) ? j->n_b : (j->n_b / j->T * (j->i + 1)) / pg * pg;
    if (hi > lo) memcpy(j->dst_b + lo, j->src_b + lo, hi - lo);
    return NULL;
}

int orc_render(const OrcTriangle *tris, uint32_t n_tris, const OrcNode *nodes, uint32_t n_nodes,
               const OrcMaterial *materials, uint32_t n_materials,
               const OrcTexture *textures, uint32_t n_textures,
               const OrcCamera *camera, const OrcOptions *opt,
               float *hdr, uint8_t *rgba8, OrcStats *stats) {
    /* renderer.rs:14-26 invariants */
    if (!opt || opt->width == 0 || opt->height == 0) return -1;
    if (opt->max_ray_depth == 0) return -2;
    if (opt->samples == 0) return -3;
    if (!tris || n_tris == 0 || !nodes || n_nodes == 0 || !materials || !camera) return -4;
    Job job; memset(&job, 0, sizeof job);
    job.sc.tris = tris; job.sc.n_tris = n_tris; job.sc.nodes = nodes; job.sc.n_nodes = n_nodes;
    job.sc.materials = materials; job.sc.n_materials = n_materials;
    job.sc.textures = textures; job.sc.n_textures = n_textures;
    job.cam = camera; job.opt = *opt; job.hdr = hdr; job.rgba8 = rgba8;
    uint64_t total = (uint64_t)opt->width * opt->height;
    if (job.opt.pix_end == 0 || job.opt.pix_end > total) job.opt.pix_end = total;
    if (job.opt.pix_begin > job.opt.pix_end) return -5;
    uint64_t stride = job.opt.pix_stride ? job.opt.pix_stride : 1;
    job.n_items = (job.opt.pix_end - job.opt.pix_begin + stride - 1) / stride;
    long hw = sysconf(_SC_NPROCESSORS_ONLN);
    uint32_t T = opt->threads ? opt->threads : (uint32_t)(hw > 0 ? hw : 1);
    if (T > 1024) T = 1024;
    job.block = job.n_items / T; if (job.block == 0) job.block = 1;      /* cpu.rs:22 */
    pthread_mutex_init(&job.mu, NULL);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * T);
    /* spread_pages: copies of the two big arrays whose pages are first touched by the worker threads, share by share */
    void *cp_tris = NULL, *cp_nodes = NULL;
    const size_t tris_bytes = (size_t)n_tris * sizeof(OrcTriangle), nodes_bytes = (size_t)n_nodes * sizeof(OrcNode);
    if (opt->spread_pages) {
        cp_tris = mmap(NULL, tris_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        cp_nodes = mmap(NULL, nodes_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (cp_tris == MAP_FAILED || cp_nodes == MAP_FAILED) { free(th); return -6; }
        SpreadJob *sj = (SpreadJob *)malloc(sizeof(SpreadJob) * T);
        for (uint32_t i = 0; i < T; i++) {
            sj[i] = (SpreadJob){(const char *)tris, (char *)cp_tris, tris_bytes, (const char *)nodes, (char *)cp_nodes, nodes_bytes, i, T};
            pthread_create(&th[i], NULL, spread_worker, &sj[i]);
        }
        for (uint32_t i = 0; i < T; i++) pthread_join(th[i], NULL);
        free(sj);
        job.sc.tris = (const OrcTriangle *)cp_tris; job.sc.nodes = (const OrcNode *)cp_nodes;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t i = 0; i < T; i++) pthread_create(&th[i], NULL, worker, &job);
    for (uint32_t i = 0; i < T; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    if (cp_tris) munmap(cp_tris, tris_bytes);
    if (cp_nodes) munmap(cp_nodes, nodes_bytes);
    pthread_mutex_destroy(&job.mu);
    if (stats) {
        *stats = job.total;
        if (stats->n_blocks) stats->block_sec_mean /= (double)stats->n_blocks;
        stats->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        stats->threads_used = T;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* scene.rs:181-194, mat4.rs:25-44 -- camera                                  */
/* ------------------------------------------------------------------------- */
void orc_camera_from_pose(const float position[3], float pitch_deg, float yaw_deg, OrcCamera *out) {
    /* f32::to_radians = x * (PI/180) with the f32 constant 0.017453292 */
    const float RADS_PER_DEG = 0.017453292519943295769236907684886f;
    float yaw = yaw_deg * RADS_PER_DEG, pitch = pitch_deg * RADS_PER_DEG;
    v3 direction = V3(cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch));
    v3 world_up = V3(0.0f, 1.0f, 0.0f);
    v3 pos = V3(position[0], position[1], position[2]);
    v3 forward = v_normalized(direction);
    v3 right = v_normalized(v_cross(world_up, forward));
    v3 up = v_cross(forward, right);
    /* Mat4f::look_at(position, position + forward, up) */
    v3 from = pos, to = v_add(pos, forward);
    v3 f = v_normalized(v_sub(from, to));
    v3 r = v_normalized(v_cross(up, f));
    v3 u = v_cross(f, r);
    memset(out, 0, sizeof *out);
    out->look_at[0][0] = r.x; out->look_at[0][1] = r.y; out->look_at[0][2] = r.z; out->look_at[0][3] = 0.0f;
    out->look_at[1][0] = u.x; out->look_at[1][1] = u.y; out->look_at[1][2] = u.z; out->look_at[1][3] = 0.0f;
    out->look_at[2][0] = f.x; out->look_at[2][1] = f.y; out->look_at[2][2] = f.z; out->look_at[2][3] = 0.0f;
    out->look_at[3][0] = from.x; out->look_at[3][1] = from.y; out->look_at[3][2] = from.z; out->look_at[3][3] = 1.0f;
    out->position = pos;
}

/* ------------------------------------------------------------------------- */
/* bvh.rs:13-204 -- binned-SAH builder, restated pass for pass                */
/* ------------------------------------------------------------------------- */
static inline OrcNode node_default(void) {             /* bvh.rs:173-182 */
    OrcNode n;
    n.bounds_min = V3(ORC_F32_MAX, ORC_F32_MAX, ORC_F32_MAX); n.first_tri_or_child = 0;
    n.bounds_max = V3(-ORC_F32_MAX, -ORC_F32_MAX, -ORC_F32_MAX); n.num_tris = 0;
    return n;
}
static inline void grow_by_tri(OrcNode *n, const OrcTriangle *t) { /* bvh.rs:185-194 */
    for (int k = 0; k < 3; k++) {
        const float *p = &t->vertices[k].position.x;
        float *mn = &n->bounds_min.x, *mx = &n->bounds_max.x;
        for (int i = 0; i < 3; i++) { mn[i] = fminf(mn[i], p[i]); mx[i] = fmaxf(mx[i], p[i]); }
    }
}
static inline float surface_area(const OrcNode *n) {   /* bvh.rs:196-203 (half area, T14) */
    v3 e = v_sub(n->bounds_max, n->bounds_min);
    return (e.x * e.z) + (e.x * e.y) + (e.z * e.y);
}
static inline float bounds_mid_axis(const OrcTriangle *t, int axis) { /* scene.rs:114-126 */
    float mn = ORC_F32_MAX, mx = -ORC_F32_MAX;
    for (int k = 0; k < 3; k++) {
        float p = (&t->vertices[k].position.x)[axis];
        mn = fminf(mn, p); mx = fmaxf(mx, p);
    }
    return (mn + mx) / 2.0f;
}
typedef struct { OrcTriangle *tris; OrcNode *nodes; uint32_t n_nodes, cap; int err; } Build;

static float evaluate_sah(const Build *b, const OrcNode *node, int axis, float pos) { /* bvh.rs:138-161 */
    OrcNode left = node_default(), right = node_default();
    for (uint32_t i = 0; i < node->num_tris; i++) {
        const OrcTriangle *t = &b->tris[node->first_tri_or_child + i];
        if (bounds_mid_axis(t, axis) < pos) { grow_by_tri(&left, t); left.num_tris += 1; }
        else { grow_by_tri(&right, t); right.num_tris += 1; }
    }
    float cost = (float)left.num_tris * surface_area(&left) + (float)right.num_tris * surface_area(&right);
    if (cost > 0.0f) return cost;                      /* NaN (empty side: 0*inf) -> f32::MAX */
    return ORC_F32_MAX;
}

static void split_node(Build *b, uint32_t index) {     /* bvh.rs:56-136; explicit stack replaces recursion */
    uint32_t *todo = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(b->cap + 2));
    size_t top = 0;
    todo[top++] = index;
    while (top) {
        uint32_t idx = todo[--top];
        uint32_t used_nodes = b->n_nodes;
        OrcNode *node = &b->nodes[idx];
        float parent_cost = (float)node->num_tris * surface_area(node);
        int best_axis = 0; float best_pos = 0.0f, best_cost = ORC_F32_MAX;
        for (int axis = 0; axis < 3; axis++) {
            float cmin = ORC_F32_MAX, cmax = -ORC_F32_MAX;   /* f32::MIN == -f32::MAX */
            for (uint32_t i = 0; i < node->num_tris; i++) {
                float m = bounds_mid_axis(&b->tris[node->first_tri_or_child + i], axis);
                cmin = fminf(cmin, m); cmax = fmaxf(cmax, m);
            }
            if (cmin == cmax) continue;
            float scale = (cmax - cmin) / 8.0f;
            for (int i = 1; i < 8; i++) {
                float pos = cmin + (float)i * scale;
                float cost = evaluate_sah(b, node, axis, pos);
                if (cost < best_cost) { best_axis = axis; best_pos = pos; best_cost = cost; }
            }
        }
        if (best_cost >= parent_cost) continue;
        uint32_t i = node->first_tri_or_child;
        uint32_t j = i + node->num_tris - 1;
        while (i <= j) {                                /* bvh.rs:99-108 */
            if (bounds_mid_axis(&b->tris[i], best_axis) < best_pos) { i += 1; }
            else {
                OrcTriangle tmp = b->tris[i]; b->tris[i] = b->tris[j]; b->tris[j] = tmp;
                if (j == 0) { b->err = -10; free(todo); return; }   /* Rust: u32 underflow panic */
                j -= 1;
            }
        }
        uint32_t a_count = i - node->first_tri_or_child;
        if (a_count == 0 || a_count == node->num_tris) continue;
        if (b->n_nodes + 2 > b->cap) { b->err = -11; free(todo); return; }
        OrcNode a = node_default(), c = node_default();
        a.first_tri_or_child = node->first_tri_or_child; a.num_tris = a_count;
        c.first_tri_or_child = i; c.num_tris = node->num_tris - a_count;
        node->first_tri_or_child = used_nodes; node->num_tris = 0;
        for (uint32_t k = 0; k < a.num_tris; k++) grow_by_tri(&a, &b->tris[a.first_tri_or_child + k]);
        for (uint32_t k = 0; k < c.num_tris; k++) grow_by_tri(&c, &b->tris[c.first_tri_or_child + k]);
        b->nodes[b->n_nodes++] = a;
        b->nodes[b->n_nodes++] = c;
        /* bvh.rs:134-135 recurses left then right, and node indices are handed out in
         * that depth-first order -- but each call allocates its children at the CURRENT
         * end of the vector, so the left subtree must be finished before the right child
         * is split: push right first, then left (LIFO). */
        todo[top++] = used_nodes + 1;
        todo[top++] = used_nodes;
    }
    free(todo);
}

int orc_bvh_build(OrcTriangle *tris, uint32_t n_tris, OrcNode *nodes_out, uint32_t nodes_cap, uint32_t *n_nodes_out) {
    if (!tris || !nodes_out || nodes_cap == 0) return -1;
    if (n_tris == 0) return -12;                       /* the reference panics on an empty scene (bvh.rs:100 underflow) */
    Build b = {tris, nodes_out, 0, nodes_cap, 0};
    OrcNode root = node_default();                     /* bvh.rs:18-24 */
    for (uint32_t i = 0; i < n_tris; i++) grow_by_tri(&root, &tris[i]);
    root.num_tris = n_tris;
    b.nodes[b.n_nodes++] = root;
    split_node(&b, 0);
    if (n_nodes_out) *n_nodes_out = b.n_nodes;
    return b.err;
}

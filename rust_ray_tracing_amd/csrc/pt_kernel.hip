// pt_kernel.hip -- the gfx950 path-tracing megakernel.
//
// One persistent wave64 = 64 independent pixel streams.  A lane owns one pixel at a time and
// runs that pixel's samples in sequence on one xorshift stream, because the reference's
// per-pixel RNG makes sample k depend on how many numbers samples 0..k-1 drew
// (reference src/renderer/backend/cpu.rs:28-29,37-58).  Lanes that run out of work are
// refilled from a global queue with one wave-aggregated atomic (ballot + mbcnt compaction);
// shading / ray generation is deferred until a ballot says enough lanes need it, so the
// divergent "service" code runs with many lanes active instead of once per finished ray.
// Inside a service pass the three normal draws of every scatter -- the pass's heaviest arithmetic --
// are spread over ALL lanes of the wave through LDS, whatever those lanes' own state.
//
// Traversal restates Ray::traverse_bvh (reference src/renderer/backend/cpu/ray.rs:84-139)
// step for step -- same visit order, same strict-< closest hit -- over re-based 64-byte child
// pairs; the per-lane stack holds 32-bit entries, the first kStackLds of them in LDS
// ([entry][lane] so a wave's access is one conflict-free ds_read/ds_write_b32).
#include "pt_kernel.h"
#include "pt_device_math.h"

namespace mipt {

namespace {

#ifndef MIPT_DIAG_STAMPS
#define MIPT_DIAG_STAMPS 0
#endif
constexpr bool DIAG_STAMPS = MIPT_DIAG_STAMPS != 0;   // counting build only: in-iteration s_memtime stamps
constexpr float kMiss = 1e30f;                     // ray.rs:79,217
constexpr uint32_t kNoTri = 0xffffffffu;
constexpr uint32_t kFrontBit = 0x80000000u;

enum : uint32_t {
    ST_T = 0,   // traversing
    ST_S = 1,   // traversal finished, needs shading
    ST_G = 2,   // needs a camera ray for its next sample
    ST_P = 3,   // needs a pixel
    ST_X = 4    // queue exhausted, lane retired
};

// stack entry: inner -> pair index (bit31 = 0);
// leaf -> bit31 | n << 25 | first_tri  (1 <= n <= 63);
// leaf with n >= 64 -> bit31 | (pair*2 + which)  (n field 0: re-read (a, n) from the pair on pop)
__device__ __forceinline__ uint32_t encode_child(uint32_t a, uint32_t n, uint32_t pair, uint32_t which) {
    const uint32_t leaf = 0x80000000u | ((n < 64u) ? ((n << 25) | a) : (pair * 2u + which));
    return (n == 0u) ? a : leaf;
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// v_min/v_max without the canonicalising v_max(x,x) hipcc puts in front of fminf/fmaxf (it guards against signalling
// NaNs; these operands are FMA results).  Semantics are IEEE minNum/maxNum = Rust f32::min/max.
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float min3_abs(float a, float b, float c) { float r; asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Guard of the exact-division fast path (fdiv_ray, pt_device_math.h), evaluated ONCE PER RAY: every |d_c| in [2^-60, 2], every
// |o_c| <= 2^40 and, per axis, |o_c| >= 2^-70 or o_c == 0 -- the latter only on axes where no bounding plane of the scene has a
// coordinate 0 < |p_c| < 2^-76 (DevScene::tiny_axes, found by mipt_scene_create, which also refuses planes beyond 2^40).
// Then for a = fl(p - o): |a| <= 2^41, so |q| = |a/d| <= 2^101 -- no overflow and no NaN (all operands finite, d != 0).  On the small
// side a is exactly 0 or |a| >= 2^-99:  o = 0 gives a = p, which is 0 or >= 2^-76;  |o| >= 2^-70 with |p| < 2^-76 gives
// |a| > 2^-71;  |o| >= 2^-70 with |p| >= 2^-76 makes both multiples of 2^-99, hence their difference too.  So q is normal and both
// residuals a - q*d (multiples of 2^(e_a - 47) >= 2^-146) are representable: fdiv_ray returns RN(a/d).  For a == 0 every term is a
// zero and the quotient is a zero whose sign may differ from IEEE's; a zero only ever meets min/max and ordered comparisons in
// slab_from_t, which do not see its sign.
__device__ __forceinline__ bool origin_safe(float x, bool zero_ok) {
    const uint32_t m = __float_as_uint(x) & 0x7fffffffu;
    return m >= 0x1c800000u /* 2^-70 */ || (m == 0u && zero_ok);
}
__device__ __forceinline__ bool ray_safe(V3 o, V3 d, uint32_t tiny_axes) {
    const float lo = 8.6736174e-19f /* 2^-60 */, hi = 2.0f, omax = 1.0995116e12f /* 2^40 */;
    return (fabsf(d.x) >= lo) && (fabsf(d.x) <= hi) && (fabsf(d.y) >= lo) && (fabsf(d.y) <= hi) &&
           (fabsf(d.z) >= lo) && (fabsf(d.z) <= hi) && (fabsf(o.x) <= omax) && (fabsf(o.y) <= omax) && (fabsf(o.z) <= omax) &&
           origin_safe(o.x, !(tiny_axes & 1u)) && origin_safe(o.y, !(tiny_axes & 2u)) && origin_safe(o.z, !(tiny_axes & 4u));
}

// MIPT_FLAG_TOUCHED (counting build): set the line's bit.  The plain read first keeps the hot lines -- the top of the tree is
// touched ~10^9 times per frame -- from serialising on one atomic each (measured: 26.8 s per launch with the bare atomicOr);
// a stale 0 from a non-coherent cache only costs a redundant atomic.
__device__ __forceinline__ void mark_line(uint32_t *bitmap, uint32_t line) {
    const uint32_t bit = 1u << (line & 31u);
    if (!(__builtin_nontemporal_load(&bitmap[line >> 5]) & bit)) atomicOr(&bitmap[line >> 5], bit);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <class R>
__device__ __forceinline__ float4 ldg4(R rsrc, uint32_t byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ray.rs:69-81 on quotients already computed (+ rt_compute.wgsl:348's t_near < max_distance when CULL)
template <bool CULL>
__device__ __forceinline__ float slab_from_t(float tminx, float tminy, float tminz, float tmaxx, float tmaxy, float tmaxz, float best) {
    float t1x = min_raw(tminx, tmaxx), t1y = min_raw(tminy, tmaxy), t1z = min_raw(tminz, tmaxz);
    float t2x = max_raw(tminx, tmaxx), t2y = max_raw(tminy, tmaxy), t2z = max_raw(tminz, tmaxz);
    float t_near = max3_raw(t1x, t1y, t1z);      // max(max(x, y), z): maxNum is associative, NaNs dropped either way
    float t_far = min3_raw(t2x, t2y, t2z);
    bool ok = (t_near <= t_far) && (t_far > 0.0f);
    if (CULL) ok = ok && (t_near < best);
    return ok ? t_near : kMiss;
}
template <bool CULL>
__device__ __forceinline__ float slab(V3 o, V3 d, float4 lo, float4 hi, float best) {
    return slab_from_t<CULL>((lo.x - o.x) / d.x, (lo.y - o.y) / d.y, (lo.z - o.z) / d.z,
                             (hi.x - o.x) / d.x, (hi.y - o.y) / d.y, (hi.z - o.z) / d.z, best);
}
// both children of a pair: exact quotients by the per-ray reciprocal, computed for every lane (straight-line code in front of
// the only branch, so the first quotients start while the later loads are still in flight; for a ray that failed ray_safe they
// are finite-or-not garbage that traps nothing) and replaced by IEEE divisions for the lanes of such rays
template <bool CULL>
__device__ __forceinline__ void slab_pair(V3 o, V3 d, V3 rd, bool safe, float4 r0, float4 r1, float4 r2, float4 r3,
                                          float best, float &d1, float &d2) {
    const float a0 = r0.x - o.x, a1 = r0.y - o.y, a2 = r0.z - o.z, a3 = r1.x - o.x, a4 = r1.y - o.y, a5 = r1.z - o.z;
    const float b0 = r2.x - o.x, b1 = r2.y - o.y, b2 = r2.z - o.z, b3 = r3.x - o.x, b4 = r3.y - o.y, b5 = r3.z - o.z;
    d1 = slab_from_t<CULL>(fdiv_ray(a0, d.x, rd.x), fdiv_ray(a1, d.y, rd.y), fdiv_ray(a2, d.z, rd.z),
                           fdiv_ray(a3, d.x, rd.x), fdiv_ray(a4, d.y, rd.y), fdiv_ray(a5, d.z, rd.z), best);
    d2 = slab_from_t<CULL>(fdiv_ray(b0, d.x, rd.x), fdiv_ray(b1, d.y, rd.y), fdiv_ray(b2, d.z, rd.z),
                           fdiv_ray(b3, d.x, rd.x), fdiv_ray(b4, d.y, rd.y), fdiv_ray(b5, d.z, rd.z), best);
    if (!safe) {
        d1 = slab<CULL>(o, d, r0, r1, best);
        d2 = slab<CULL>(o, d, r2, r3, best);
    }
}

// texture.rs:33-38; out-of-range indices (reference: panic) are clamped and counted
__device__ __forceinline__ V3 texel_rgb(const DevScene &sc, uint32_t offset, uint32_t width, uint32_t height, float u, float v, DevStats *st) {
    float fu = u - truncf(u), fv = v - truncf(v);                  // f32::fract
    float fi = fu * (float)width, fj = fv * (float)height;
    // Rust `as i32`: saturating, NaN -> 0
    long long i = (fi != fi) ? 0ll : (fi >= 2147483648.0f ? 2147483647ll : (fi <= -2147483648.0f ? -2147483648ll : (long long)(int)fi));
    long long j = (fj != fj) ? 0ll : (fj >= 2147483648.0f ? 2147483647ll : (fj <= -2147483648.0f ? -2147483648ll : (long long)(int)fj));
    long long index = i + j * (long long)width;
    long long n = (long long)width * (long long)height;
    if (index < 0 || index >= n) {
        index = index < 0 ? 0 : n - 1;
        atomicAdd(&st->tex_clamped, 1ull);
    }
    uint32_t px = sc.texels[(size_t)offset + (size_t)index];
    return mk(u8_over_255(px & 255u), u8_over_255((px >> 8) & 255u), u8_over_255((px >> 16) & 255u)); // vec3.rs:252-260
}

// ---------------------------------------------------------------------------------------------------------------
// Shading mode 1: the wgpu backend's material model (rt_compute.wgsl:126-294, 503-569), restated operator for operator
// exactly as the CPU oracle restates it (one rounded f32 op per WGSL operator, transcendentals through the glibc restatement, exact
// f32 bilinear weights).  SURVEY 8(f) rank 2.  Returns true when the path ends (`break` in the WGSL loop).
// ---------------------------------------------------------------------------------------------------------------
struct V4 { float x, y, z, w; };
__device__ __forceinline__ float w_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ V4 sample_texture_bilinear(const uint32_t *texels, uint32_t offset, uint32_t width, uint32_t height, float u, float v) {
    const uint32_t W = width, H = height;                                    // textureSampleLevel: linear, repeat (gpu.rs:393-401)
    const float uu = u * (float)W - 0.5f, vv = v * (float)H - 0.5f;
    const float fu = floorf(uu), fv = floorf(vv);
    float a = uu - fu, b = vv - fv;
    const int32_t ic = (fabsf(fu) < 1e9f) ? (int32_t)fu : 0, jc = (fabsf(fv) < 1e9f) ? (int32_t)fv : 0;
    if (!(a == a)) a = 0.0f;
    if (!(b == b)) b = 0.0f;
    const uint32_t i0 = floor_mod(ic, W), j0 = floor_mod(jc, H);             // texel (ic, jc) and its +1 neighbours, wrapped (|ic|, |jc| < 1e9)
    const uint32_t i1 = (i0 + 1u == W) ? 0u : i0 + 1u, j1 = (j0 + 1u == H) ? 0u : j0 + 1u;
    const size_t row0 = (size_t)offset + (size_t)j0 * W, row1 = (size_t)offset + (size_t)j1 * W;
    const uint32_t p00 = texels[row0 + i0], p10 = texels[row0 + i1];
    const uint32_t p01 = texels[row1 + i0], p11 = texels[row1 + i1];
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float t00 = u8_over_255((p00 >> (8 * c)) & 255u), t10 = u8_over_255((p10 >> (8 * c)) & 255u);
        const float t01 = u8_over_255((p01 >> (8 * c)) & 255u), t11 = u8_over_255((p11 >> (8 * c)) & 255u);
        const float top = t00 * (1.0f - a) + t10 * a;
        const float bot = t01 * (1.0f - a) + t11 * a;
        out[c] = top * (1.0f - b) + bot * b;
    }
    V4 r; r.x = out[0]; r.y = out[1]; r.z = out[2]; r.w = out[3];
    return r;
}
__device__ __forceinline__ void build_onb(V3 n, V3 &tangent, V3 &bitangent) {                  // rt_compute.wgsl:565-569
    const V3 up = (fabsf(n.z) < 0.9999999f) ? mk(0.0f, 0.0f, 1.0f) : mk(1.0f, 0.0f, 0.0f);
    tangent = normalized(cross(up, n));
    bitangent = cross(n, tangent);
}
__device__ __forceinline__ V3 to_world(V3 t, V3 b, V3 n, V3 l) {
    return mk((t.x * l.x + b.x * l.y) + n.x * l.z, (t.y * l.x + b.y * l.y) + n.y * l.z, (t.z * l.x + b.z * l.y) + n.z * l.z);
}
__device__ __forceinline__ V3 to_local(V3 t, V3 b, V3 n, V3 w) { return mk(dot(t, w), dot(b, w), dot(n, w)); }

__device__ __forceinline__ V3 sample_ggx_vndf(V3 ve, float ax, float ay, uint32_t &rng) {      // rt_compute.wgsl:503-525
    const float u1 = rand_f32(rng), u2 = rand_f32(rng);
    const V3 Vh = normalized(mk(ax * ve.x, ay * ve.y, ve.z));
    const float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = mk(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) { const float inv = 1.0f / __builtin_sqrtf(lensq); T1 = mk(-Vh.y * inv, Vh.x * inv, 0.0f * inv); }
    const V3 T2 = cross(Vh, T1);
    const float r = __builtin_sqrtf(u1);
    const float phi = 2.0f * 3.1415926535f * u2;
    const float t1 = r * gl_cosf(phi);
    float t2 = r * gl_sinf(phi);
    const float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * __builtin_sqrtf(1.0f - t1 * t1) + s * t2;
    const float k = __builtin_sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
    const V3 Nh = (T1 * t1 + T2 * t2) + Vh * k;
    return normalized(mk(ax * Nh.x, ay * Nh.y, fmaxf(0.0f, Nh.z)));
}
// cosine_sample_hemisphere (rt_compute.wgsl:527-551), split: the two RNG draws happen where the shader calls the function;
// the direction itself is a pure function of them and is only evaluated on the branch that uses it.
__device__ __forceinline__ V3 cosine_hemisphere_from(float ux, float uy) {
    const float ox = 2.0f * ux - 1.0f, oy = 2.0f * uy - 1.0f;
    float dx, dy;
    if (ox == 0.0f && oy == 0.0f) { dx = 0.0f; dy = 0.0f; }
    else {
        float theta, r;
        if (fabsf(ox) > fabsf(oy)) { r = ox; theta = 0.7853981634f * (oy / ox); }
        else { r = oy; theta = 1.5707963268f - 0.7853981634f * (ox / oy); }
        dx = r * gl_cosf(theta); dy = r * gl_sinf(theta);
    }
    const float z = __builtin_sqrtf(fmaxf(0.0f, 1.0f - dx * dx - dy * dy));
    return mk(dx, dy, z);
}

__device__ __forceinline__ bool shade_wgsl(const DevScene &sc, V3 &o, V3 &d, V3 &ray_color, V3 &incoming, V3 &prev_hit_point,
                                        uint32_t depth, uint32_t &rng, float t, float u, float v, uint32_t best_tri, uint32_t &n_tex) {
    const float EPSILON = 0.0001f;
    const uint32_t tri = best_tri & ~kFrontBit;
    const bool front_face = (best_tri & kFrontBit) != 0u;
    const float4 a0 = sc.tri_attr[(size_t)tri * 4 + 0], a1 = sc.tri_attr[(size_t)tri * 4 + 1];
    const float4 a2 = sc.tri_attr[(size_t)tri * 4 + 2], a3 = sc.tri_attr[(size_t)tri * 4 + 3];
    const float w = 1.0f - u - v;
    V3 raw_n = mk(a0.x, a0.y, a0.z) * w + mk(a0.w, a1.x, a1.y) * u + mk(a1.z, a1.w, a2.x) * v;
    if (!front_face) raw_n = mk(-raw_n.x, -raw_n.y, -raw_n.z);
    V3 normal = normalized(raw_n);                                                     // rt_compute.wgsl:328
    const float uvx = ((a2.y * w) + (a2.w * u)) + (a3.y * v);
    const float uvy = ((a2.z * w) + (a3.x * u)) + (a3.z * v);
    const V3 point = mk(__builtin_fmaf(d.x, t, o.x), __builtin_fmaf(d.y, t, o.y), __builtin_fmaf(d.z, t, o.z));   // :318 fma
    const DevMaterialFull m = sc.mats_full[__float_as_uint(a3.w)];
    V3 base = mk(m.base[0], m.base[1], m.base[2]), emission = mk(m.emission[0], m.emission[1], m.emission[2]);
    float ior = m.ior, transparency = m.transparency, roughness = m.roughness, metallic = m.metallic;
    if (front_face) ior = 1.0f / ior;                                                  // set_surface_properties, :251-294
    if (m.tex[0][1] != 0u) {
        const V4 tx = sample_texture_bilinear(sc.texels, m.tex[0][0], m.tex[0][1], m.tex[0][2], uvx, uvy); n_tex++;
        base = mk(gl_powf(tx.x, 2.2f), gl_powf(tx.y, 2.2f), gl_powf(tx.z, 2.2f));
    }
    if (m.tex[1][1] != 0u) { transparency = sample_texture_bilinear(sc.texels, m.tex[1][0], m.tex[1][1], m.tex[1][2], uvx, uvy).w; n_tex++; }
    if (m.tex[2][1] != 0u) { roughness = sample_texture_bilinear(sc.texels, m.tex[2][0], m.tex[2][1], m.tex[2][2], uvx, uvy).y; n_tex++; }
    if (m.tex[3][1] != 0u) { metallic = sample_texture_bilinear(sc.texels, m.tex[3][0], m.tex[3][1], m.tex[3][2], uvx, uvy).z; n_tex++; }
    if (m.tex[4][1] != 0u) {
        const V4 tx = sample_texture_bilinear(sc.texels, m.tex[4][0], m.tex[4][1], m.tex[4][2], uvx, uvy); n_tex++;
        emission = mk(gl_powf(tx.x, 2.2f), gl_powf(tx.y, 2.2f), gl_powf(tx.z, 2.2f));
    }
    V3 tangent, bitangent;
    build_onb(normal, tangent, bitangent);
    V3 tbn_n = normal;
    if (m.tex[5][1] != 0u) {
        const V4 tx = sample_texture_bilinear(sc.texels, m.tex[5][0], m.tex[5][1], m.tex[5][2], uvx, uvy); n_tex++;
        normal = normalized(to_world(tangent, bitangent, tbn_n, mk(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f)));
        build_onb(normal, tangent, bitangent);
        tbn_n = normal;
    }
    float transmitted_distance = t;                                                    // :143-148
    if (front_face) prev_hit_point = point;
    else transmitted_distance = length(point - prev_hit_point);
    if (transparency < rand_f32(rng)) {                                                // alpha cut-out, :150-153
        o = point + d * EPSILON;
        return false;
    }
    const float alpha = w_clamp(roughness * roughness, EPSILON, 1.0f);
    const V3 neg_dir = mk(-d.x, -d.y, -d.z);
    const V3 sampled_normal = to_world(tangent, bitangent, tbn_n, sample_ggx_vndf(to_local(tangent, bitangent, tbn_n, neg_dir), alpha, alpha, rng));
    // integer literal exponents = repeated multiplication (same reading as the CPU oracle)
    const float f0s = ((1.0f - ior) * (1.0f - ior)) / ((1.0f + ior) * (1.0f + ior));
    const V3 f0 = mk(f0s * (1.0f - metallic) + base.x * metallic, f0s * (1.0f - metallic) + base.y * metallic, f0s * (1.0f - metallic) + base.z * metallic);
    const float p1 = 1.0f - dot(sampled_normal, neg_dir), p2 = p1 * p1;
    const float p5 = (p2 * p2) * p1;                                                   // schlick_fresnel, :553-555
    const V3 fresnel = mk(f0.x + (1.0f - f0.x) * p5, f0.y + (1.0f - f0.y) * p5, f0.z + (1.0f - f0.z) * p5);
    // The shader evaluates reflect / refract / cosine-hemisphere eagerly (:160-165) and then uses one of them; they are pure
    // functions of values fixed here, so only the RNG draws keep their place and each direction is computed on its branch.
    const float hemi_ux = rand_f32(rng), hemi_uy = rand_f32(rng);                      // cosine_sample_hemisphere's draws
    bool specular = false, transmitted = false;                                        // select_bsdf, :231-248
    {
        const float r = rand_f32(rng);
        if (metallic > r) specular = true;
        else if (metallic + m.transmission > r) transmitted = true;
    }
    V3 new_dir;
    const float fl = length(fresnel);
    const float r2 = rand_f32(rng);
    if (fl < r2 && !specular) {                                                        // :167-186
        ray_color = ray_color * base;
        if (transmitted) {
            const float ndi = dot(sampled_normal, d);                                  // refract
            const float k = 1.0f - ior * ior * (1.0f - ndi * ndi);
            const V3 r = (k < 0.0f) ? mk(0.0f, 0.0f, 0.0f) : (d * ior - sampled_normal * (ior * ndi + __builtin_sqrtf(k)));
            new_dir = normalized(r);
            if (dot(new_dir, normal) > 0.0f) return true;
            V3 absorption = mk(1.0f, 1.0f, 1.0f);
            if (!front_face)
                absorption = mk(gl_expf(-(1.0f - base.x) * transmitted_distance), gl_expf(-(1.0f - base.y) * transmitted_distance),
                                gl_expf(-(1.0f - base.z) * transmitted_distance));
            ray_color = ray_color * absorption;
        } else {
            new_dir = normalized(to_world(tangent, bitangent, tbn_n, cosine_hemisphere_from(hemi_ux, hemi_uy)));
        }
    } else {                                                                           // :187-196
        if (specular) ray_color = ray_color * fresnel;
        const float two_ndi = 2.0f * dot(sampled_normal, d);
        new_dir = normalized(d - sampled_normal * two_ndi);                            // reflect
        if (dot(new_dir, normal) < 0.0f) return true;
    }
    float rr = 1.0f;                                                                   // Russian roulette, :198-207
    if (depth >= 4u) {
        rr = fmaxf(ray_color.x, fmaxf(ray_color.z, ray_color.y));
        if (rr < rand_f32(rng)) return true;
    }
    ray_color = ray_color / rr;
    incoming = incoming + emission * ray_color;                                        // :209
    o = point + new_dir * EPSILON;
    d = new_dir;
    return false;
}

} // namespace

#ifndef MIPT_MIN_WAVES_PER_SIMD
#define MIPT_MIN_WAVES_PER_SIMD 5      // CPU-backend shading: 96 VGPRs, no scratch (97 without the bound = 4 waves)
#endif
#ifndef MIPT_MIN_WAVES_UNCULLED
#define MIPT_MIN_WAVES_UNCULLED 4      // the CPU backend's un-culled traversal: squeezed into 96 VGPRs it runs 169 ms, at its natural 98 (4 waves) 143 ms
#endif
#ifndef MIPT_MIN_WAVES_SHADING1
#define MIPT_MIN_WAVES_SHADING1 4      // wgpu-shader shading, fully inlined: 127 VGPRs + 20 B scratch, 21.1 ms on config M (146 VGPRs at 3 waves: 25.2 ms; a non-inlined
                                       // shade_wgsl: 272 B of call frame, 33.3 ms -- tools/experiments/mode1_noinline_shade_result.txt)
#endif
template <bool COUNT, bool CULL, int SHADING>
__global__ __launch_bounds__(kBlockThreads, SHADING == 0 ? ((CULL && !COUNT) ? MIPT_MIN_WAVES_PER_SIMD : MIPT_MIN_WAVES_UNCULLED) : MIPT_MIN_WAVES_SHADING1) void pt_trace_kernel(DevScene sc, DevParams pr) {
    __shared__ uint32_t s_stack[kWavesPerBlock][kStackLds + 1][64];   // row kStackLds: scratch target of the branch-free push
    __shared__ double s_logtab[32];                                    // __logf_data.tab (16 x {invc, logc}) for gl_log10f
    __shared__ __attribute__((aligned(16))) float s_draw[kWavesPerBlock][64 * 6];                   // scatter draws of a service pass: 3 x {u_theta, u_rho} per hit lane, compacted
    if (threadIdx.x < 32u) s_logtab[threadIdx.x] = gl_d(glibc_logf_tab, (int)threadIdx.x);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wib = threadIdx.x >> 6;
    uint32_t(*stk)[64] = s_stack[wib];
    uint32_t *ovf = pr.ovf + ((size_t)blockIdx.x * kWavesPerBlock + wib) * (size_t)(kStackOvf * 64) + lane;

    const auto geom = __builtin_amdgcn_make_buffer_rsrc((void *)sc.pairs, 0, (int)sc.geom_bytes, 0x00020000);

    // ---- per-lane path state ----
    uint32_t state = ST_P;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), rd = mk(0, 0, 1);   // rd = 1/d per component (exact-division helper)
    bool dir_safe = false;
    V3 ray_color = mk(1, 1, 1), incoming = mk(0, 0, 0), emitted = mk(0, 0, 0), final_color = mk(0, 0, 0);
    uint32_t rng = 0, pix = 0, slot = 0, sample = 0, bounces = 0;
    V3 prev_hit_point = mk(0, 0, 0);                       // SHADING == 1 only (rt_compute.wgsl:130)
    float screen_x = 0, screen_y = 0;
    // ---- per-lane traversal state ----
    float best_t = kMiss, best_u = 0, best_v = 0;
    uint32_t best_tri = kNoTri;
    uint32_t tri_cur = 0, tri_end = 0, pair = 0, sp = 0;
    // ---- counters (COUNT build only) ----
    unsigned long long c_rays = 0, c_inner = 0, c_tris = 0, c_hits = 0, c_tex = 0;
    uint32_t c_maxsp = 0, c_pixels = 0;
    unsigned long long g_iters = 0, g_inner = 0, g_leaf = 0, g_it_inner = 0, g_it_leaf = 0, g_serv = 0, g_serv_lanes = 0;  // lane 0 only
    unsigned long long g_t_serv = 0, g_t_start = COUNT ? clock64() : 0ull, g_t0 = 0, g_t_mem = 0, g_t_first_x = 0;

    uint32_t leaf_wait = 0;                                // wave-uniform: iterations until the next scheduled leaf phase
    for (;;) {
        const unsigned long long m_t = __ballot(state == ST_T);
        const unsigned long long m_need = __ballot(state != ST_T && state != ST_X);
        const uint32_t n_t = (uint32_t)__popcll(m_t), n_need = (uint32_t)__popcll(m_need);
        if ((n_t | n_need) == 0u) break;

        // ---------------- service: shade / finish pixel / fetch pixel / camera ray ------------
        if (n_need != 0u && (n_t == 0u || n_need * pr.service_den >= (n_t + n_need) * pr.service_num)) {
            if (COUNT) { g_serv++; g_serv_lanes += n_need; g_t0 = clock64(); }
            bool start_ray = false;
            bool scatter = false;                                                  // CPU-backend shading: this lane hit and draws a new direction
            V3 sc_normal = mk(0, 0, 0);
            bool path_done = false;
            if (state == ST_S) {
                if (SHADING == 1 && best_tri != kNoTri) {                          // rt_compute.wgsl:137-213
                    bounces += 1;                                                  // curr_ray_depth += 1 (before the cut-out test)
                    uint32_t n_tex = 0;
                    const bool ended = shade_wgsl(sc, o, d, ray_color, incoming, prev_hit_point, bounces, rng, best_t, best_u, best_v, best_tri, n_tex);
                    if (COUNT) { c_hits++; c_tex += n_tex; }
                    path_done = ended || !(bounces < pr.max_depth);
                } else if (SHADING == 1) {                                         // miss, rt_compute.wgsl:215-223
                    ray_color = ray_color * mk(1.0f, 1.0f, 1.0f);
                    incoming = incoming + mk(1.0f, 1.0f, 1.0f) * ray_color;
                    path_done = true;
                } else if (best_tri != kNoTri) {                                   // ray.rs:152-183
                    const uint32_t tri = best_tri & ~kFrontBit;
                    const float4 a0 = sc.tri_attr[(size_t)tri * 4 + 0], a1 = sc.tri_attr[(size_t)tri * 4 + 1];
                    const float4 a2 = sc.tri_attr[(size_t)tri * 4 + 2], a3 = sc.tri_attr[(size_t)tri * 4 + 3];
                    if (COUNT && pr.touched) mark_line(pr.touched, pr.touched_attr_base + (tri >> 1));
                    const float u = best_u, v = best_v;
                    const float w = 1.0f - u - v;                                   // ray.rs:45
                    V3 normal = mk(a0.x, a0.y, a0.z) * w + mk(a0.w, a1.x, a1.y) * u + mk(a1.z, a1.w, a2.x) * v;
                    if (!(best_tri & kFrontBit)) normal = mk(-normal.x, -normal.y, -normal.z); // ray.rs:46-48
                    const float uvx = ((a2.y * w) + (a2.w * u)) + (a3.y * v);       // ray.rs:50-53
                    const float uvy = ((a2.z * w) + (a3.x * u)) + (a3.z * v);
                    const DevMaterial m = sc.mats[__float_as_uint(a3.w)];           // ray.rs:153-154
                    if (m.base_w != 0u) {                                           // ray.rs:162-169
                        ray_color = ray_color * texel_rgb(sc, m.base_off, m.base_w, m.base_h, uvx, uvy, pr.stats);
                        if (COUNT) c_tex++;
                    } else {
                        ray_color = ray_color * mk(m.base[0], m.base[1], m.base[2]);
                    }
                    if (m.emis_w != 0u) {                                           // ray.rs:170-176
                        emitted = emitted + texel_rgb(sc, m.emis_off, m.emis_w, m.emis_h, uvx, uvy, pr.stats);
                        if (COUNT) c_tex++;
                    } else {
                        emitted = emitted + mk(m.emis[0], m.emis[1], m.emis[2]);
                    }
                    incoming = incoming + emitted * ray_color;                       // ray.rs:177
                    scatter = true; sc_normal = normal;                              // ray.rs:179-183 follow below, wave-cooperatively
                    if (COUNT) c_hits++;
                } else {                                                             // ray.rs:184-193
                    ray_color = ray_color * mk(1.0f, 1.0f, 1.0f);
                    emitted = emitted + mk(1.0f, 1.0f, 1.0f);
                    incoming = incoming + emitted * ray_color;
                    path_done = true;
                }
            }
            if (SHADING == 0) {
                // ---- Vec3f::rand_in_unit_sphere (vec3.rs:66-68) for every hit lane of the pass, spread over the whole wave ----
                // A scatter draws three rand_f32_nd (math.rs:15-19): six xorshift steps on the lane's own stream, then three
                // evaluations of sqrt(-2 log10 u_rho) * cos(6.283185 u_theta) that depend on nothing but their two uniforms.  Only
                // ~a quarter of the lanes are in the pass and the rest would sit out ~1 000 instructions of f64 arithmetic: so
                // the 3 h evaluations of the pass's h hit lanes are compacted through LDS (item 3 * rank + k at float2 slot
                // 3 * rank + k) and every lane of the wave -- whatever its own state -- evaluates item `lane` (+64, +128 when
                // 3 h > 64) and writes the result over the item's first word.  Same operations on the same operands: same bits.
                float *dr = s_draw[wib];
                const unsigned long long m_sc = __ballot(scatter);
                const uint32_t n_items = 3u * (uint32_t)__popcll(m_sc);
                if (n_items != 0u) {                                                 // wave-uniform
                    const uint32_t base = 6u * lane_rank(m_sc);
                    if (scatter) {
#pragma unroll
                        for (int k = 0; k < 6; k++) dr[base + k] = rand_f32(rng);  // u_theta, u_rho of x; of y; of z -- the draw order of vec3.rs:67
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t item = lane; item < n_items; item += 64u) {
                        const float2 uu = *reinterpret_cast<const float2 *>(dr + 2u * item);
                        dr[2u * item] = rand_f32_nd_eval(uu.x, uu.y, (const double *)s_logtab);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (scatter) {
                        const V3 rs = normalized(mk(dr[base], dr[base + 2u], dr[base + 4u]));
                        const V3 point = o + d * best_t;                             // ray.rs:60
                        const V3 new_dir = normalized(sc_normal + rs);              // ray.rs:179-180
                        o = point + new_dir * 0.0001f;                               // ray.rs:181
                        d = new_dir;
                        bounces += 1;
                        path_done = !(bounces < pr.max_depth);                       // ray.rs:147
                    }
                }
            }
            if (state == ST_S) {
                if (path_done) {
                    const V3 res = (bounces == 0u) ? incoming : incoming / (float)bounces; // ray.rs:197-201
                    final_color = final_color + res;                                 // cpu.rs:52
                    sample += 1;
                    if (sample < pr.samples) {
                        state = ST_G;
                    } else {
                        if (!pr.sum_only) final_color = final_color / pr.samples_f;  // cpu.rs:60
                        float *dst = pr.hdr + (size_t)slot * 3;
                        if (pr.accumulate) {      // progressive rendering: add this call's samples to the running sum
                            final_color = mk(dst[0] + final_color.x, dst[1] + final_color.y, dst[2] + final_color.z);
                        }
                        dst[0] = final_color.x; dst[1] = final_color.y; dst[2] = final_color.z;
                        c_pixels++;
                        state = ST_P;
                    }
                } else {
                    state = ST_T;                                                    // next bounce
                    start_ray = true;
                }
            }
            // ---- fetch a pixel: one atomic per wave, compacted over the lanes that need one ----
            // (Eight per-XCD band queues with stealing were tried: 91.4 ms vs 89.9 ms with this single queue -- after the
            //  first bounce the rays are incoherent, so L2 affinity buys nothing and the bands are unevenly loaded.)
            {
                const unsigned long long m_p = __ballot(state == ST_P);
                if (m_p != 0ull) {
                    unsigned long long base = 0;
                    const uint32_t leader = (uint32_t)__ffsll((long long)m_p) - 1u;
                    if (lane == leader) base = atomicAdd(&pr.stats->queue, (unsigned long long)__popcll(m_p));
                    base = __shfl(base, (int)leader);
                    if (state == ST_P) {
                        const unsigned long long wi = base + lane_rank(m_p);
                        if (wi >= pr.total_work) {
                            state = ST_X;
                            if (COUNT && g_t_first_x == 0) g_t_first_x = clock64();
                        } else {
                            // 8x8 pixel tiles, round-robin over ranks: global tile = local*world + rank
                            const uint32_t lt = (uint32_t)(wi >> 6), p = (uint32_t)wi & 63u;
                            const uint32_t lt_o = pr.reverse_tiles ? (pr.n_local_tiles - 1u - lt) : lt;
                            const uint32_t gt = lt_o * pr.tile_world + pr.tile_rank;
                            const uint32_t px = (gt % pr.tiles_x) * 8u + (p & 7u);
                            const uint32_t py = (gt / pr.tiles_x) * 8u + (p >> 3);
                            if (px < pr.width && py < pr.height) {            // ragged edge tiles: skip, stay ST_P
                                pix = py * pr.width + px;
                                slot = pr.packed ? (lt_o * 64u + p) : pix;
                                rng = 987612486u * (pix + 87636354u);                 // cpu.rs:28-29
                                const uint32_t y = pr.height - py;                    // cpu.rs:32
                                screen_x = ((((float)px / (float)pr.width) * 2.0f) - 1.0f) * pr.aspect; // cpu.rs:33-34
                                screen_y = (((float)y / (float)pr.height) * 2.0f) - 1.0f;               // cpu.rs:35
                                final_color = mk(0.0f, 0.0f, 0.0f);
                                sample = 0;
                                state = ST_G;
                            }
                        }
                    }
                }
            }
            // ---- camera ray (cpu.rs:37-50) ----
            if (state == ST_G) {
                if (SHADING == 1 || pr.seed_mode != 0u) {                             // rt_compute.wgsl:102
                    const uint32_t px = pix % pr.width, py = pix / pr.width;
                    rng = (pr.sample_begin + sample) * 6023u + (757283u * px + 872653746u * py);
                }
                const float jx = (rand_f32(rng) * 2.0f - 1.0f) * 0.0005f;
                const float jy = (rand_f32(rng) * 2.0f - 1.0f) * 0.0005f;
                const float rx = -screen_x + jx, ry = screen_y + jy, rz = 1.0f;
                // Mat4f * Vec3f, upper-left 3x3, data[col][row] (mat4.rs:143-152)
                const V3 dir = mk(pr.cam[0] * rx + pr.cam[3] * ry + pr.cam[6] * rz,
                                  pr.cam[1] * rx + pr.cam[4] * ry + pr.cam[7] * rz,
                                  pr.cam[2] * rx + pr.cam[5] * ry + pr.cam[8] * rz);
                d = normalized(dir);
                o = mk(pr.cam[9], pr.cam[10], pr.cam[11]);
                ray_color = mk(1.0f, 1.0f, 1.0f);                                     // ray.rs:142-146
                incoming = mk(0.0f, 0.0f, 0.0f);
                emitted = mk(0.0f, 0.0f, 0.0f);
                bounces = 0;
                if (SHADING == 1) prev_hit_point = o;
                state = ST_T;
                start_ray = true;
            }
            // ---- start traverse_bvh (ray.rs:84-88, HitInfo::default :214-226) ----
            if (start_ray) {
                best_t = kMiss; best_u = 0.0f; best_v = 0.0f; best_tri = kNoTri;
                rd = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                dir_safe = ray_safe(o, d, sc.tiny_axes);
                sp = 0; pair = 0u;
                tri_cur = sc.root_a; tri_end = sc.root_a + sc.root_n;   // root leaf (root_n > 0) or inner (empty range)
                if (COUNT) c_rays++;
            }
            if (COUNT) g_t_serv += clock64() - g_t0;
            continue;   // re-evaluate the ballots
        }

        // ---------------- one traversal step per traversing lane ---------------------------------
        // Leaf phases.  On the 10 M-triangle scene ~5 of the wave's 64 lanes are at a triangle in any one iteration, yet the
        // Moeller-Trumbore block costs the wave as much as it would for 64.  So triangle tests run only in a "leaf phase": when
        // pr.leaf_period iterations have passed since the last one, when at least 1/pr.leaf_den of the traversing lanes wait for
        // one, or when no lane has an inner step to do; lanes that reach a leaf in between sit out.  Pure scheduling: a lane's
        // own sequence of steps is unchanged.  Period 4 on the full frame of config M: 82.4 -> 79.4 ms (2: 81.2, 3: 80.2, 5: 79.5,
        // 6: 81.1, 8: 84.1).  The fraction release is what lets scenes with many leaf visits gain too (dragon, 5.5 triangle tests
        // per 18.5 inner steps: a bare period 4 costs it 16 %, with release at 1/4 it gains 4 %; helmet +9 % / -2 %;
        // tools/sweep_service.py).  A tile shard (1/2 ... 1/8 frame) is bound by its longest pixel chain, where waiting costs
        // 2 - 8 %: the host passes period 1 (mipt_api.cpp).
        bool leaf_hold;
        {
            const unsigned long long m_lf = __ballot(state == ST_T && tri_cur < tri_end);
            const unsigned long long m_in = __ballot(state == ST_T && !(tri_cur < tri_end));
            const bool phase = leaf_wait == 0u || m_in == 0ull || (uint32_t)__popcll(m_lf) * pr.leaf_den >= n_t;
            leaf_wait = phase ? pr.leaf_period - 1u : leaf_wait - 1u;
            leaf_hold = !phase;
            if (COUNT) {                                     // wave-occupancy diagnostics (MiptStats.diag[0..4]); lane 0's copy is reported
                g_iters++;
                g_inner += (unsigned long long)__popcll(m_in); g_it_inner += (m_in != 0ull) ? 1u : 0u;
                g_leaf += phase ? (unsigned long long)__popcll(m_lf) : 0ull; g_it_leaf += (phase && m_lf != 0ull) ? 1u : 0u;
            }
        }
        if (state == ST_T && !(leaf_hold && tri_cur < tri_end)) {
            const bool leaf = tri_cur < tri_end;
            // one buffer descriptor over [pairs | tri_pos], 32-bit byte offset per lane (no 64-bit address math)
            const uint32_t voff = leaf ? (sc.tri_off_bytes + tri_cur * kTriPosStride) : (pair * 64u);
            unsigned long long g_ta = 0;
            if (COUNT && DIAG_STAMPS) g_ta = clock64();
            // top of the stack, read now so that its LDS latency hides under the global loads: a step that pops never pushes
            const uint32_t top_e = stk[(sp - 1u) & (uint32_t)(kStackLds - 1)][lane];
            float4 r0, r1, r2, r3;
            r0 = ldg4(geom, voff); r1 = ldg4(geom, voff + 16u); r2 = ldg4(geom, voff + 32u);
            r3 = ldg4(geom, voff + 48u);                                 // tri_pos is padded by one float4
            if (COUNT && pr.touched) mark_line(pr.touched, voff >> 7);                               // 128-B line voff / 128
            // Keep all four 16-B loads in front of the inner/leaf branch: without this barrier LLVM sinks the last
            // two into the inner branch, i.e. a second dependent memory round trip per step (measured: -10 % time).
            asm volatile("" ::: "memory");
            if (COUNT && DIAG_STAMPS) {      // diagnostic only: split an iteration into memory wait and the rest
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                g_t_mem += clock64() - g_ta;
            }
            bool need_pop = false;
            if (leaf) {                                                              // ray.rs:19-67, 90-99
                const V3 v0 = mk(r0.x, r0.y, r0.z), e1 = mk(r0.w, r1.x, r1.y), e2 = mk(r1.z, r1.w, r2.x);
                const V3 rce2 = cross(d, e2);
                const float det = dot(e1, rce2);
                const float inv_det = 1.0f / det;
                const V3 s = o - v0;
                const float u = inv_det * dot(s, rce2);
                const V3 sce1 = cross(s, e1);
                const float v = inv_det * dot(d, sce1);
                const float t = inv_det * dot(e2, sce1);
                // ray.rs:56-59 verbatim boolean form: NaN u/v pass, NaN t fails;
                // `!(det < 0.0 && det > -0.0)` is identically true and is dropped.
                const bool has_hit = (t > 0.0f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f);
                if (has_hit && t < best_t) {                                         // ray.rs:96 (strict <)
                    best_t = t; best_u = u; best_v = v;
                    best_tri = __float_as_uint(r2.y) | ((det > 0.0f) ? kFrontBit : 0u);   // the record carries its triangle's index in the reference order; ray.rs:39
                }
                if (COUNT) c_tris++;
                tri_cur += 1;
                need_pop = (tri_cur == tri_end);
            } else {                                                                 // ray.rs:108-137
                const float max_d = best_t * pr.cull_scale;
                float d1, d2;
                slab_pair<CULL>(o, d, rd, dir_safe, r0, r1, r2, r3, max_d, d1, d2);
                uint32_t a1 = __float_as_uint(r0.w), n1 = __float_as_uint(r1.w);
                uint32_t a2 = __float_as_uint(r2.w), n2 = __float_as_uint(r3.w);
                uint32_t w2 = 1u;
                if (COUNT) c_inner++;
                if (d1 > d2) {                                                       // ray.rs:120-123
                    float td = d1; d1 = d2; d2 = td;
                    uint32_t ta = a1; a1 = a2; a2 = ta;
                    uint32_t tn = n1; n1 = n2; n2 = tn;
                    w2 = 0u;
                }
                if (d1 == kMiss) {                                                   // ray.rs:124-130
                    need_pop = true;
                } else {
                    const bool push = d2 < kMiss;                                    // ray.rs:133-136
                    const uint32_t e = encode_child(a2, n2, pair, w2);
                    const bool in_lds = sp < (uint32_t)kStackLds;
                    stk[(push && in_lds) ? sp : (uint32_t)kStackLds][lane] = e;      // no branch: non-pushing lanes hit the scratch row
                    if (push && !in_lds) {                                           // rare: spill region / overflow
                        if (sp < (uint32_t)(kStackLds + kStackOvf)) ovf[(size_t)(sp - kStackLds) * 64] = e;
                        else atomicAdd(&pr.stats->stack_overflows, 1ull);             // reference: panic (ray.rs:85)
                    }
                    sp += (push && sp < (uint32_t)(kStackLds + kStackOvf)) ? 1u : 0u;
                    if (COUNT) c_maxsp = sp > c_maxsp ? sp : c_maxsp;
                    if (n1 > 0u) { tri_cur = a1; tri_end = a1 + n1; }                // ray.rs:131 node = child_1
                    else { pair = a1; }
                }
            }
            if (need_pop) {                                                          // ray.rs:100-105, 125-129
                if (sp == 0u) {
                    state = ST_S;
                } else {
                    sp -= 1;
                    uint32_t e = top_e;
                    if (sp >= (uint32_t)kStackLds) e = ovf[(size_t)(sp - kStackLds) * 64];  // rare
                    if (e & 0x80000000u) {
                        uint32_t n = (e >> 25) & 63u, a = e & 0x01ffffffu;
                        if (n == 0u) {                                               // big leaf: child-ref form
                            const uint32_t ref = e & 0x7fffffffu;
                            const float4 *q = sc.pairs + (size_t)(ref >> 1) * 4 + (ref & 1u) * 2;
                            a = __float_as_uint(q[0].w); n = __float_as_uint(q[1].w);
                        }
                        tri_cur = a; tri_end = a + n;
                    } else {
                        pair = e; tri_cur = 0; tri_end = 0;
                    }
                }
            }
        }
    }

    if (COUNT) {
        atomicAdd(&pr.stats->rays, c_rays);
        atomicAdd(&pr.stats->inner_steps, c_inner);
        atomicAdd(&pr.stats->tri_tests, c_tris);
        atomicAdd(&pr.stats->hits, c_hits);
        atomicAdd(&pr.stats->texel_fetches, c_tex);
        atomicMax(&pr.stats->max_stack, (unsigned long long)c_maxsp);
        if (lane == 0) {
            atomicAdd(&pr.stats->d_iters, g_iters); atomicAdd(&pr.stats->d_inner_lanes, g_inner);
            atomicAdd(&pr.stats->d_leaf_lanes, g_leaf); atomicAdd(&pr.stats->d_iters_inner, g_it_inner);
            atomicAdd(&pr.stats->d_iters_leaf, g_it_leaf); atomicAdd(&pr.stats->d_services, g_serv);
            atomicAdd(&pr.stats->d_service_lanes, g_serv_lanes);
            atomicAdd(&pr.stats->d_cycles_service, g_t_serv);
            atomicAdd(&pr.stats->d_cycles_total, clock64() - g_t_start);
            atomicAdd(&pr.stats->d_cycles_mem, g_t_mem);
        }
        {   // tail: wave-cycles between the first lane of the wave finding the queue empty and the wave's exit
            const unsigned long long tx = __shfl(g_t_first_x, (int)(__ffsll((long long)__ballot(g_t_first_x != 0)) - 1));
            if (lane == 0 && tx != 0) atomicAdd(&pr.stats->d_cycles_tail, clock64() - tx);
        }
    }
    if (c_pixels) atomicAdd(&pr.stats->pixels, (unsigned long long)c_pixels);
}

// ---- all-gathered rank-packed tile slices -> full frame -----------------------------------
__global__ void unpack_tiles_kernel(const float *__restrict__ packed_all, uint32_t width, uint32_t height,
                                    uint32_t world, uint32_t tiles_x, uint32_t n_local_tiles, float *__restrict__ hdr) {
    const unsigned long long n = (unsigned long long)width * height;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t px = (uint32_t)(i % width), py = (uint32_t)(i / width);
        const uint32_t gt = (py >> 3) * tiles_x + (px >> 3);
        const uint32_t rank = gt % world, lt = gt / world;
        const unsigned long long src = ((unsigned long long)rank * n_local_tiles + lt) * 64ull + ((py & 7u) * 8u + (px & 7u));
        hdr[i * 3 + 0] = packed_all[src * 3 + 0];
        hdr[i * 3 + 1] = packed_all[src * 3 + 1];
        hdr[i * 3 + 2] = packed_all[src * 3 + 2];
    }
}

// ---- cpu.rs:60 on a reduced sum buffer: final_color /= samples (sample-sharded renders divide once, after the reduce) ----
__global__ void popcount_kernel(const uint32_t *__restrict__ bitmap, unsigned long long n_words, unsigned long long *out) {
    unsigned long long c = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
         i += (unsigned long long)gridDim.x * blockDim.x)
        c += (unsigned long long)__popc(bitmap[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63u) == 0u && c) atomicAdd(out, c);
}
hipError_t launch_popcount(const uint32_t *bitmap, unsigned long long n_words, unsigned long long *out, hipStream_t stream) {
    hipLaunchKernelGGL(popcount_kernel, dim3(1024), dim3(256), 0, stream, bitmap, n_words, out);
    return hipGetLastError();
}

__global__ void divide_kernel(float *__restrict__ hdr, unsigned long long n_floats, float divisor) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_floats;
         i += (unsigned long long)gridDim.x * blockDim.x)
        hdr[i] = hdr[i] / divisor;
}
hipError_t launch_divide(float *hdr, unsigned long long n_floats, float divisor, hipStream_t stream) {
    hipLaunchKernelGGL(divide_kernel, dim3(2048), dim3(256), 0, stream, hdr, n_floats, divisor);
    return hipGetLastError();
}

// ---- cpu.rs:60-64 epilogue: (x scale) -> linear_to_srgb -> floor(x*255) clamp -> [r,g,b,255] ----
__global__ void tonemap_kernel(const float *__restrict__ hdr, unsigned long long n, float divisor, uint32_t *__restrict__ rgba8) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        float r = hdr[i * 3 + 0], g = hdr[i * 3 + 1], b = hdr[i * 3 + 2];
        if (divisor != 1.0f) { r = r / divisor; g = g / divisor; b = b / divisor; }   // cpu.rs:60
        rgba8[i] = srgb_quantize(r) | (srgb_quantize(g) << 8) | (srgb_quantize(b) << 16) | 0xff000000u;
    }
}

// ---- pp_compute.wgsl:7-34: linear_to_srgb, THEN aces_filmic, stored as rgba16unorm -------------------------------
// One rounded f32 operation per WGSL operator (WGSL leaves FMA fusion to the implementation: unpinned); pow through
// the same powf as the sRGB epilogue; unorm16 = floor(x * 65535 + 0.5).
__device__ __forceinline__ float pp_channel(float c) {
    const float cutoff = (c < 0.0031308f) ? 1.0f : 0.0f;
    const float higher = 1.055f * gl_powf(c, 1.0f / 2.4f) - 0.055f;
    const float lower = c * 12.92f;
    const float x = (higher * (1.0f - cutoff)) + lower * cutoff;          // mix(higher, lower, cutoff)
    const float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
    float y = (x * (a * x + b)) / (x * (cc * x + d) + e);
    y = fminf(fmaxf(y, 0.0f), 1.0f);                                      // clamp; NaN -> 0 like WGSL clamp(min(max()))
    return y;
}
__global__ void postprocess_kernel(const float *__restrict__ hdr, unsigned long long n, float divisor, uint16_t *__restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        float c[3] = {hdr[i * 3 + 0], hdr[i * 3 + 1], hdr[i * 3 + 2]};
        for (int k = 0; k < 3; k++) {
            float v = c[k];
            if (divisor != 1.0f) v = v / divisor;
            v = fminf(fmaxf(v, 0.0f), 1.0f);       // the rt texture is rgba16unorm: radiance is clamped to [0,1] before pp (gpu.rs:140)
            out[i * 4 + k] = (uint16_t)floorf(pp_channel(v) * 65535.0f + 0.5f);
        }
        out[i * 4 + 3] = 65535;
    }
}

hipError_t launch_postprocess(const float *hdr, unsigned long long n_pixels, float divisor, uint16_t *rgba16, hipStream_t stream) {
    hipLaunchKernelGGL(postprocess_kernel, dim3(2048), dim3(256), 0, stream, hdr, n_pixels, divisor, rgba16);
    return hipGetLastError();
}

template <bool COUNT, bool CULL, int SHADING>
static hipError_t launch_t(const DevScene &sc, const DevParams &pr, int grid, hipStream_t stream) {
    hipLaunchKernelGGL((pt_trace_kernel<COUNT, CULL, SHADING>), dim3(grid), dim3(kBlockThreads), 0, stream, sc, pr);
    return hipGetLastError();
}
template <bool COUNT, bool CULL, int SHADING>
static int occ_t() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pt_trace_kernel<COUNT, CULL, SHADING>, kBlockThreads, 0) != hipSuccess) n = 1;
    return n;
}
// instantiations: {CPU-backend shading, wgpu-shader shading} x {count, cull}
#define MIPT_DISPATCH(FN, ...)                                                                                          \
    do {                                                                                                                \
        if (shading == 1) {                                                                                             \
            if (count) return cull ? FN<true, true, 1>(__VA_ARGS__) : FN<true, false, 1>(__VA_ARGS__);                  \
            return cull ? FN<false, true, 1>(__VA_ARGS__) : FN<false, false, 1>(__VA_ARGS__);                           \
        }                                                                                                               \
        if (count) return cull ? FN<true, true, 0>(__VA_ARGS__) : FN<true, false, 0>(__VA_ARGS__);                      \
        return cull ? FN<false, true, 0>(__VA_ARGS__) : FN<false, false, 0>(__VA_ARGS__);                               \
    } while (0)

hipError_t launch_trace(const DevScene &sc, const DevParams &pr, bool count, bool cull, int shading, int grid, hipStream_t stream) {
    MIPT_DISPATCH(launch_t, sc, pr, grid, stream);
}
static int occ_dispatch(bool count, bool cull, int shading) { MIPT_DISPATCH(occ_t); }
int trace_blocks_per_cu(bool count, bool cull, int shading) {
    int n = occ_dispatch(count, cull, shading);
    if (n < 1) n = 1;
    if (n > 8) n = 8;
    return n;
}

hipError_t launch_unpack_tiles(const float *packed_all, uint32_t width, uint32_t height, uint32_t world,
                               float *hdr, hipStream_t stream) {
    const uint32_t tiles_x = (width + 7u) / 8u, tiles_y = (height + 7u) / 8u;
    const uint32_t n_local = (tiles_x * tiles_y + world - 1u) / world;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3(2048), dim3(256), 0, stream, packed_all, width, height, world,
                       tiles_x, n_local, hdr);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *hdr, unsigned long long n_pixels, float divisor, uint8_t *rgba8, hipStream_t stream) {
    hipLaunchKernelGGL(tonemap_kernel, dim3(2048), dim3(256), 0, stream, hdr, n_pixels, divisor,
                       reinterpret_cast<uint32_t *>(rgba8));
    return hipGetLastError();
}

} // namespace mipt

// pt_device_math.h -- gfx950 device arithmetic for the path tracer.
//
// Every function is one IEEE-754 binary32 (or, in the libm restatement, binary64)
// operation per source operator, evaluated in the order of the reference's Rust source;
// (note: __builtin_sqrtf is the correctly-rounded expansion; __fsqrt_rn lowers to the 1-ulp v_sqrt_f32)
// the translation unit is built with -ffp-contract=off so hipcc never fuses a*b+c (Rust
// does not: SURVEY.md T11).  f32 divide and sqrt are the correctly-rounded expansions
// (-fhip-fp32-correctly-rounded-divide-sqrt), denormals are preserved.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipt {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }  // vec3.rs:272-278
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }  // vec3.rs:288-294
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }  // vec3.rs:304-310
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }     // vec3.rs:312-318
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }     // vec3.rs:344-350
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); } // vec3.rs:130-134
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                    // vec3.rs:136-144
    return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
__device__ __forceinline__ float length(V3 a) { return __builtin_sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); } // vec3.rs:93-97
__device__ __forceinline__ V3 normalized(V3 a) { return a / length(a); }                             // vec3.rs:105-109

// math.rs:6-13
__device__ __forceinline__ uint32_t xor_shift(uint32_t &s) {
    uint32_t x = s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    s = x;
    return x;
}
// math.rs:22-24 -- u32::MAX as f32 == 2^32; v_cvt_f32_u32 rounds to nearest even like `as f32`
__device__ __forceinline__ float rand_f32(uint32_t &s) { return (float)xor_shift(s) / 4294967296.0f; }

// ---- glibc 2.35 binary32 transcendentals, restated for gfx950 ------------------------------------------------------
// The reference's CPU backend calls Rust std f32::cos / f32::log10 (src/math.rs:15-19) and f32::powf
// (src/math/vec3.rs:87); Rust forwards them to the platform libm = glibc.  These are glibc 2.35's algorithms
// (sysdeps/ieee754/flt-32/{s_cosf.c, s_sinf.c, s_sincosf.h, e_logf.c, e_log10f.c, e_powf.c, e_expf.c}) in the form the
// x86_64 build executes on an FMA-capable CPU -- the *_fma IFUNC variants, in which gcc fused every a*b+c of those
// files (read off the disassembly of libm.so.6, Ubuntu GLIBC 2.35-0ubuntu3.11).  Each __builtin_fma below is one
// vfmadd of that binary = one v_fma_f64 here; everything else is one rounded IEEE operation; the translation unit is
// built with -ffp-contract=off so nothing else fuses.  Tables: glibc_flt32_data.h (generated from that libm).
// Pinned: tests/test_libm_pin.py (CPU restatement == the machine's libm on every binary32 of the path's domains) and
// tests/test_gpu_libm.py (these device functions == the CPU restatement on the same sweeps).
#define GLIBC_FLT32_TABLE static __device__ const
#define GLIBC_FLT32_CONST static constexpr            // read at constant indices only: every coefficient folds into an immediate
#include "glibc_flt32_data.h"

__device__ __forceinline__ double gl_d(const uint64_t *t, int i) { return __longlong_as_double((long long)t[i]); }
__device__ __forceinline__ uint32_t gl_abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
__device__ __forceinline__ float gl_invalidf(float x) { return (x - x) / (x - x); }

// s_sincosf.h sinf_poly.  x86 table layout: sign[4] @0, hpi_inv @4, hpi @5, c0 @6, c1 @7, s1 @8, c2 @9, s2 @10,
// c3 @11, s3 @12, c4 @13.  __sincosf_table[1] (used when n & 2) holds the same sine coefficients and the negated
// cosine coefficients, and the cosine polynomial reads x only through x2: so table[1]'s cosine result is the exact
// negation of table[0]'s (negation commutes with every rounding) and its sine result is table[0]'s -- no second table.
__device__ __forceinline__ double gl_sin_poly(double x, double x2) {
    const uint64_t *p = glibc_sincosf_tab;
    const double x3 = x * x2;
    const double s1 = __builtin_fma(x2, gl_d(p, 12), gl_d(p, 10));   // s2 + x2*s3
    const double x7 = x3 * x2;
    const double s = __builtin_fma(x3, gl_d(p, 8), x);               // x + x3*s1
    return __builtin_fma(s1, x7, s);                                 // s + x7*s1
}
__device__ __forceinline__ double gl_cos_poly(double x2) {
    const uint64_t *p = glibc_sincosf_tab;
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, gl_d(p, 13), gl_d(p, 11));   // c3 + x2*c4
    const double c1 = __builtin_fma(x2, gl_d(p, 7), gl_d(p, 6));     // c0 + x2*c1
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, gl_d(p, 9), c1);              // c1 + x4*c2
    return __builtin_fma(c2, x6, c);                                 // c + x6*c2
}
// reduce_fast (|x| < 120) / reduce_large (integer multiplication by 4/pi); returns the reduced argument, n in *np
__device__ __forceinline__ double gl_reduce_fast(double x, int *np) {
    const double r = x * gl_d(glibc_sincosf_tab, 4);
    const int n = ((int)r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double)n, gl_d(glibc_sincosf_tab, 5), x);
}
__device__ __forceinline__ double gl_reduce_large(uint32_t xi, int *np) {
    const uint32_t *arr = &glibc_inv_pio4[(xi >> 26) & 15u];
    const int shift = (int)((xi >> 23) & 7u);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(long long)res0 * gl_d(glibc_pi63, 0);
}
// s_cosf.c / s_sinf.c in one body: COS selects `n ^ 1` (cosf) or `n` (sinf) as sinf_poly's quadrant argument
template <bool COS>
__device__ __forceinline__ float gl_sincosf(float y) {
    double x = (double)y;
    const uint32_t top = gl_abstop12(y);
    int n = 0, q = 0;                                                 // q: index into sign[] and the (n & 2) table select
    if (top < 0x3f4u /* abstop12(pi/4) */) {
        if (top < 0x398u /* abstop12(0x1p-12f) */) return COS ? 1.0f : y;
    } else if (top < 0x42fu /* abstop12(120.0f) */) {
        x = gl_reduce_fast(x, &n);
        q = n;
    } else if (top < 0x7f8u /* abstop12(inf) */) {
        const uint32_t xi = __float_as_uint(y);
        x = gl_reduce_large(xi, &n);
        q = n + (int)(xi >> 31);
    } else {
        return gl_invalidf(y);
    }
    const int sel = COS ? (n ^ 1) : n;                                // sinf_poly(x * s, x * x, p, sel)
    const double x2 = x * x;
    double r;
    if ((sel & 1) == 0) {
        const double s = ((q + 1) & 2) ? -1.0 : 1.0;                  // sign[q & 3] = {1, -1, -1, 1}
        r = gl_sin_poly(x * s, x2);
    } else {
        r = gl_cos_poly(x2);
        if (q & 2) r = -r;                                            // __sincosf_table[1]
    }
    return (float)r;
}
__device__ __forceinline__ float gl_cosf(float y) { return gl_sincosf<true>(y); }
__device__ __forceinline__ float gl_sinf(float y) { return gl_sincosf<false>(y); }

// e_logf.c.  `tab` = __logf_data.tab as 16 x {invc, logc}: the global table, or a copy the kernel staged in LDS.
template <class TabPtr>
__device__ __forceinline__ float gl_logf(float x, TabPtr tab) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_inff();                  // __math_divzerof(1)
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return gl_invalidf(x);
        ix = __float_as_uint(x * 8388608.0f);                         // subnormal: x * 0x1p23f
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int k = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double z = (double)__uint_as_float(iz);
    const double r = __builtin_fma(z, invc, -1.0);                                    // z*invc - 1
    const double y0 = __builtin_fma((double)k, gl_d(glibc_logf_ln2, 0), logc);        // logc + k*Ln2
    const double r2 = r * r;
    double y = __builtin_fma(gl_d(glibc_logf_poly, 1), r, gl_d(glibc_logf_poly, 2));  // A1*r + A2
    y = __builtin_fma(gl_d(glibc_logf_poly, 0), r2, y);                               // A0*r2 + y
    y = __builtin_fma(y, r2, y0 + r);                                                 // y*r2 + (y0 + r)
    return (float)y;
}
// e_log10f.c (__ieee754_log10f): binary32 arithmetic, never fused (the file has no FMA variant)
template <class TabPtr>
__device__ __forceinline__ float gl_log10f(float x, TabPtr tab) {
    const float two25 = 33554432.0f;
    const float log10_2lo = __uint_as_float(glibc_log10f_consts[1]), ivln10 = __uint_as_float(glibc_log10f_consts[2]);
    const float log10_2hi = __uint_as_float(glibc_log10f_consts[3]);
    int hx = (int)__float_as_uint(x), k = 0;
    if (hx < 0x00800000) {
        if ((hx & 0x7fffffff) == 0) return -two25 / fabsf(x);         // log(+-0) = -inf
        if (hx < 0) return (x - x) / (x - x);                         // log(-#) = NaN
        k -= 25;
        x *= two25;
        hx = (int)__float_as_uint(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    const int i = (int)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    x = __uint_as_float((uint32_t)hx);
    const float z = y * log10_2lo + ivln10 * gl_logf(x, tab);
    return z + y * log10_2hi;
}
struct GlTabGlobal {                                                 // the table in global (constant) memory
    __device__ __forceinline__ double operator[](uint32_t i) const { return gl_d(glibc_logf_tab, (int)i); }
};

// e_powf.c
__device__ __forceinline__ double gl_pow_log2_inline(uint32_t ix) {
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;
    const double invc = gl_d(glibc_pow_log2_tab, 2 * (int)i), logc = gl_d(glibc_pow_log2_tab, 2 * (int)i + 1);
    const double z = (double)__uint_as_float(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = __builtin_fma(gl_d(glibc_pow_log2_poly, 0), r, gl_d(glibc_pow_log2_poly, 1));
    const double p = __builtin_fma(gl_d(glibc_pow_log2_poly, 2), r, gl_d(glibc_pow_log2_poly, 3));
    const double r4 = r2 * r2;
    double q = __builtin_fma(gl_d(glibc_pow_log2_poly, 4), r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    return y;
}
__device__ __forceinline__ float gl_exp2_inline(double xd, uint32_t sign_bias) {
    const double SHIFT = gl_d(glibc_exp2f_shift_scaled, 0);
    double kd = xd + SHIFT;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = xd - kd;
    uint64_t t = glibc_exp2f_tab[ki & 31u];
    const uint64_t ski = ki + sign_bias;
    t += ski << 47;
    const double s = __longlong_as_double((long long)t);
    const double z = __builtin_fma(gl_d(glibc_exp2f_poly, 0), r, gl_d(glibc_exp2f_poly, 1));
    const double r2 = r * r;
    double y = __builtin_fma(gl_d(glibc_exp2f_poly, 2), r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}
__device__ __forceinline__ int gl_checkint(uint32_t iy) {             // 0: not an integer, 1: odd, 2: even
    const int e = (int)(iy >> 23 & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
__device__ __forceinline__ bool gl_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
__device__ __forceinline__ bool gl_issignalingf(float x) { return 2u * (__float_as_uint(x) ^ 0x00400000u) > 2u * 0x7fc00000u; }
__device__ __forceinline__ float gl_xflowf(uint32_t sign, float y) { return (sign ? -y : y) * y; }   // math_errf.c
__device__ __forceinline__ float gl_powf(float x, float y) {
    const float INF = __builtin_inff();
    uint32_t sign_bias = 0;
    uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || gl_zeroinfnan(iy)) {
        if (gl_zeroinfnan(iy)) {
            if (2u * iy == 0u) return gl_issignalingf(x) ? x + y : 1.0f;
            if (ix == 0x3f800000u) return gl_issignalingf(y) ? x + y : 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (gl_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && gl_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
            if (2u * ix == 0u && (iy & 0x80000000u)) return sign_bias ? -INF : INF;   // __math_divzerof
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {
            const int yint = gl_checkint(iy);
            if (yint == 0) return gl_invalidf(x);
            if (yint == 1) sign_bias = 1u << 16;                      // SIGN_BIAS
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {
            ix = __float_as_uint(x * 8388608.0f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double logx = gl_pow_log2_inline(ix);
    const double ylogx = (double)y * logx;
    if ((((uint64_t)__double_as_longlong(ylogx) >> 47) & 0xffffu) >= 0x80bfu /* asuint64(126.0) >> 47 */) {
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -INF : INF;              // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                         // __math_uflowf
        if (ylogx < -149.0) return gl_xflowf(sign_bias, 0x1.4p-75f);                  // __math_may_uflowf
    }
    return gl_exp2_inline(ylogx, sign_bias);
}
// e_expf.c
__device__ __forceinline__ float gl_expf(float x) {
    const double xd = (double)x;
    const uint32_t abstop = gl_abstop12(x);
    if (abstop >= 0x42bu /* abstop12(88.0f) */) {
        if (__float_as_uint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();
        if (x < -0x1.9fe368p6f) return 0.0f;
        if (x < -0x1.9d1d9ep6f) return gl_xflowf(0, 0x1.4p-75f);
    }
    const double InvLn2N = gl_d(glibc_exp2f_invln2_scaled, 0), SHIFT = gl_d(glibc_exp2f_shift, 0);
    double kd = __builtin_fma(InvLn2N, xd, SHIFT);                    // z + SHIFT, z = InvLn2N*xd (fused in the binary)
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= SHIFT;
    const double r = __builtin_fma(InvLn2N, xd, -kd);                 // z - kd (vfmsub)
    uint64_t t = glibc_exp2f_tab[ki & 31u];
    t += ki << 47;
    const double s = __longlong_as_double((long long)t);
    const double z = __builtin_fma(gl_d(glibc_exp2f_poly_scaled, 0), r, gl_d(glibc_exp2f_poly_scaled, 1));
    const double r2 = r * r;
    double y = __builtin_fma(gl_d(glibc_exp2f_poly_scaled, 2), r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

// ---- the two libm calls of the scatter step, specialised to the arguments rand_f32 can produce -------------------------
// rand_f32 (math.rs:22-24) returns RN(k) / 2^32 for a 32-bit k: exactly 0, or a value in [2^-32, 1] -- never negative,
// subnormal, infinite or NaN.  On that domain gl_log10f / gl_cosf reduce to straight-line code (the general functions spend
// a third of their instructions on exec-mask scaffolding for cases that cannot occur here, inside the service pass where
// only ~15 lanes of the wave are active).  Same operations in the same order on every reachable argument, so the results
// are the general functions' bits; tests/test_gpu_libm.py sweeps EVERY binary32 of both domains against the CPU restatement.
//
// log10f on {0} u [2^-32, 1]: x normal and positive, so e_log10f.c's k = exponent, i = (k < 0), x' = mantissa * 2^-i in [0.5, 2)
// and e_logf.c runs without its special-case block; logf(1.0f) is the explicit `return 0` of e_logf.c:40.
template <class TabPtr>
__device__ __forceinline__ float gl_log10f_unit(float x, TabPtr tab) {
    const float log10_2lo = __uint_as_float(glibc_log10f_consts[1]), ivln10 = __uint_as_float(glibc_log10f_consts[2]);
    const float log10_2hi = __uint_as_float(glibc_log10f_consts[3]);
    const int hx = (int)__float_as_uint(x);
    const int k = (hx >> 23) - 127;
    const int i = (int)((uint32_t)k >> 31);
    const uint32_t ix = ((uint32_t)hx & 0x007fffffu) | ((uint32_t)(0x7f - i) << 23);
    const float y = (float)(k + i);
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t j = (tmp >> 19) & 15u;
    const int kk = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = tab[2 * j], logc = tab[2 * j + 1];
    const double z = (double)__uint_as_float(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)kk, gl_d(glibc_logf_ln2, 0), logc);
    const double r2 = r * r;
    double p = __builtin_fma(gl_d(glibc_logf_poly, 1), r, gl_d(glibc_logf_poly, 2));
    p = __builtin_fma(gl_d(glibc_logf_poly, 0), r2, p);
    p = __builtin_fma(p, r2, y0 + r);
    const float lg = (ix == 0x3f800000u) ? 0.0f : (float)p;          // e_logf.c: logf(1) = 0 exactly
    const float zf = y * log10_2lo + ivln10 * lg;
    const float res = zf + y * log10_2hi;
    return (hx == 0) ? -__builtin_inff() : res;                       // e_log10f.c: log10(+0) = -two25 / 0 = -inf
}
// cosf on [0, 6.2831855] (theta = 6.283185f * rand_f32): below pi/4 reduce_fast gives n = 0 and leaves x untouched
// (fma(-0, hpi, x) = x), so s_cosf.c's small-argument branch and its reduce_fast branch are the same arithmetic; below
// 2^-12 the cosine polynomial itself rounds to 1.0f, which is what s_cosf.c returns there.  Both polynomials are evaluated
// and one is selected (the quadrant is random: a branch would run both sides anyway).
__device__ __forceinline__ float gl_cosf_2pi(float y) {
    int n;
    const double x = gl_reduce_fast((double)y, &n);
    const double x2 = x * x;
    const double s = ((n + 1) & 2) ? -1.0 : 1.0;                      // sign[n & 3] = {1, -1, -1, 1}
    const double sn = gl_sin_poly(x * s, x2);
    double cs = gl_cos_poly(x2);
    if (n & 2) cs = -cs;                                              // __sincosf_table[1]
    return (float)(((n ^ 1) & 1) ? cs : sn);
}

// math.rs:15-19 (log10, not ln: SURVEY T3)
template <class TabPtr>
__device__ __forceinline__ float rand_f32_nd(uint32_t &s, TabPtr logtab) {
    float theta = 6.283185f * rand_f32(s);
    float rho = __builtin_sqrtf(-2.0f * gl_log10f_unit(rand_f32(s), logtab));
    return rho * gl_cosf_2pi(theta);
}
// The arithmetic of one rand_f32_nd draw given its two uniforms (u_theta drawn first, u_rho second): the same operations in
// the same order, so the same bits -- the kernel's service pass evaluates the three draws of a scatter on three lanes at once.
template <class TabPtr>
__device__ __forceinline__ float rand_f32_nd_eval(float u_theta, float u_rho, TabPtr logtab) {
    const float theta = 6.283185f * u_theta;
    const float rho = __builtin_sqrtf(-2.0f * gl_log10f_unit(u_rho, logtab));
    return rho * gl_cosf_2pi(theta);
}
// vec3.rs:66-68 -- x, y, z drawn in that order
template <class TabPtr>
__device__ __forceinline__ V3 rand_in_unit_sphere(uint32_t &s, TabPtr logtab) {
    float x = 0.0f, y = 0.0f, z = 0.0f;
#pragma nounroll
    for (int k = 0; k < 3; k++) { x = y; y = z; z = rand_f32_nd(s, logtab); }   // one copy of the code, three draws in order
    return normalized(mk(x, y, z));
}

// vec3.rs:80-90 (+ mix :197-205) and :262-270, one channel
__device__ __forceinline__ uint32_t srgb_quantize(float c) {
    float cutoff = (c < 0.0031308f) ? 1.0f : 0.0f;
    float higher = 1.055f * gl_powf(c, 1.0f / 2.4f) - 0.055f;
    float lower = c * 12.92f;
    float s = (higher * (1.0f - cutoff)) + lower * cutoff;
    float q = floorf(s * 255.0f);
    if (q < 0.0f) q = 0.0f;
    if (q > 255.0f) q = 255.0f;
    return (q != q) ? 0u : (uint32_t)q;          // Rust `as u8`: NaN -> 0
}

// ---- exact division with a per-ray reciprocal -------------------------------------------------------------
// The slab test divides by the ray direction 12 times per inner step (ray.rs:70-71) and bit-exactness forbids
// `x * (1/d)`.  hipcc's IEEE expansion is 11 instructions per quotient (2 div_scale, rcp, 2 reciprocal-refinement
// FMAs, mul, 3 residual FMAs, div_fmas, div_fixup).  Everything that depends only on d is hoisted to once per ray:
// r = RN(1/d); per quotient the same quotient refinement the hardware sequence ends with remains --
//   q0 = a*r; q1 = q0 + (a - q0*d)*r; q2 = q1 + (a - q1*d)*r      (residuals exact in FMA)
// q1 is faithful, q2 = RN(a/d) (Markstein's theorem) -- PROVIDED nothing under/overflows.  That is guaranteed by a guard evaluated
// once per ray and once per scene (ray_safe in pt_kernel.hip states it and the argument): |d| in [2^-60, 2], |o| and the plane
// coordinates <= 2^40, and o either >= 2^-70 or 0 (0 only on axes without tiny plane coordinates), so that a = p - o is 0 or
// >= 2^-99.  Rays failing it do the step with IEEE divisions.  The probe op 14 and tests/test_gpu_more.py::test_fast_division_is_ieee pin q2 == a/d bit for bit on 10^7
// quotients incl. hard cases, the a == 0 zeros and the 2^-99 edge.
__device__ __forceinline__ float fdiv_ray(float a, float d, float r) {
    const float q0 = a * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, d, a), r, q0);
    return __builtin_fmaf(__builtin_fmaf(-q1, d, a), r, q1);
}
// floored i mod W for |i| < 2^30, W in [1, 2^32): one 32-bit unsigned division instead of the two 64-bit ones of
// ((i % W) + W) % W -- hipcc expands a 64-bit modulo to ~100 instructions (the bilinear sampler of shading mode 1 wraps four
// coordinates per sample)
__device__ __forceinline__ uint32_t floor_mod(int32_t i, uint32_t W) {
    const uint32_t a = (uint32_t)(i ^ (i >> 31));        // i >= 0: i;  i < 0: -i - 1
    const uint32_t m = a % W;
    return i >= 0 ? m : W - 1u - m;
}
// u8 -> f32 / 255.0 (vec3.rs:252-260) with the same exact two-correction quotient as fdiv_ray: the numerator is an
// integer in [0, 255] and the divisor the constant 255, so no range guard is needed (0 gives 0 exactly);
// tests/test_gpu_more.py checks all 256 values against IEEE division.
__device__ __forceinline__ float u8_over_255(uint32_t k) {
    return fdiv_ray((float)k, 255.0f, 0.0039215688593685627f /* RN(1/255) */);
}

} // namespace mipt

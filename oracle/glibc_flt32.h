/* glibc_flt32.h -- TEST INFRASTRUCTURE (part of the CPU oracle; the product never includes this file).
 *
 * Restatement of the binary32 transcendentals the reference's CPU backend reaches through Rust std:
 *     f32::cos   -> cosf    reference src/math.rs:18
 *     f32::log10 -> log10f  reference src/math.rs:17
 *     f32::powf  -> powf    reference src/math/vec3.rs:87 (linear_to_srgb)
 * plus sinf / expf, which only the wgpu-shader shading mode needs (rt_compute.wgsl; unpinned there anyway).
 * Rust std forwards these to the platform libm, which on x86_64 Linux is glibc.  The third-party dependency is
 * therefore **glibc 2.35** (this image: Ubuntu GLIBC 2.35-0ubuntu3.11; not in the reference's Cargo.lock because the
 * toolchain provides it), and what is restated is its published algorithm:
 *     sysdeps/ieee754/flt-32/s_cosf.c, s_sinf.c, s_sincosf.h   (sincosf "optimized routines", double-precision kernels)
 *     sysdeps/ieee754/flt-32/e_logf.c, e_powf.c, e_expf.c      (table + polynomial in double)
 *     sysdeps/ieee754/flt-32/e_log10f.c                        (fdlibm float code on top of logf)
 * in the form the x86_64 build actually executes on an FMA-capable CPU: cosf, sinf, logf, powf and expf are IFUNCs
 * whose *_fma variants are compiled with -mfma -mavx2, so gcc contracted every `a * b + c` of those files into one
 * fused multiply-add.  Which operations are fused was read off the disassembly of libm.so.6 (all of them, see the
 * comment on each function); log10f has no variant and uses separate binary32 multiplies and adds around its call to
 * the logf IFUNC.  Every `fma_()` below is one vfmadd in the binary; every other operator is one rounded operation.
 * This file must be compiled with -ffp-contract=off so the compiler adds no fusion of its own.
 *
 * Pinned by tests/test_libm_pin.py: for every binary32 argument of the domains the path uses (and, with
 * MIPT_LIBM_SWEEP=full, for all 2^32 arguments) the functions below return the bits this machine's libm returns.
 * Non-FMA CPUs (the *_sse2 variants) round a few arguments differently and are not what is matched. */
#pragma once
#include <stdint.h>
#include <string.h>
#include "glibc_flt32_data.h"

static inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
static inline uint32_t gl_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float gl_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t gl_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double gl_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static inline double gl_d(const uint64_t *t, int i) { return gl_asdouble(t[i]); }

/* math_errf.c: the value each error helper returns in round-to-nearest (errno / exceptions are not modelled) */
static inline float gl_invalidf(float x) { return (x - x) / (x - x); }
static inline float gl_oflowf(uint32_t sign) { return sign ? -__builtin_inff() : __builtin_inff(); }   /* 0x1p97f * 0x1p97f */
static inline float gl_uflowf(uint32_t sign) { return sign ? -0.0f : 0.0f; }                            /* 0x1p-95f * 0x1p-95f */
static inline float gl_may_uflowf(uint32_t sign) {                                                      /* 0x1.4p-75f * 0x1.4p-75f */
    const float y = 0x1.4p-75f;
    return (sign ? -y : y) * y;
}
static inline float gl_divzerof(uint32_t sign) { return sign ? -__builtin_inff() : __builtin_inff(); }

/* ---- s_sincosf.h ------------------------------------------------------------------------------------------------ */
/* x86 table layout: sign[4] @0, hpi_inv @4, hpi @5, c0 @6, c1 @7, s1 @8, c2 @9, s2 @10, c3 @11, s3 @12, c4 @13 */
enum { GL_SC_HPI_INV = 4, GL_SC_HPI = 5, GL_SC_C0 = 6, GL_SC_C1 = 7, GL_SC_S1 = 8, GL_SC_C2 = 9, GL_SC_S2 = 10,
       GL_SC_C3 = 11, GL_SC_S3 = 12, GL_SC_C4 = 13, GL_SC_STRIDE = 14 };

static inline uint32_t gl_abstop12(float x) { return (gl_asuint(x) >> 20) & 0x7ff; }

/* sinf_poly (s_sincosf.h): n even -> sine polynomial, n odd -> cosine polynomial; all mul-adds fused in *_fma */
static inline float gl_sinf_poly(double x, double x2, const uint64_t *p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma_(x2, gl_d(p, GL_SC_S3), gl_d(p, GL_SC_S2));   /* s2 + x2*s3 */
        const double x7 = x3 * x2;
        const double s = fma_(x3, gl_d(p, GL_SC_S1), x);                    /* x + x3*s1 */
        return (float)fma_(s1, x7, s);                                      /* s + x7*s1 */
    } else {
        const double x4 = x2 * x2;
        const double c2 = fma_(x2, gl_d(p, GL_SC_C4), gl_d(p, GL_SC_C3));   /* c3 + x2*c4 */
        const double c1 = fma_(x2, gl_d(p, GL_SC_C1), gl_d(p, GL_SC_C0));   /* c0 + x2*c1 */
        const double x6 = x4 * x2;
        const double c = fma_(x4, gl_d(p, GL_SC_C2), c1);                   /* c1 + x4*c2 */
        return (float)fma_(c2, x6, c);                                      /* c + x6*c2 */
    }
}
/* reduce_fast: |x| < 120; hpi_inv is prescaled by 2^24 */
static inline double gl_reduce_fast(double x, const uint64_t *p, int *np) {
    const double r = x * gl_d(p, GL_SC_HPI_INV);
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma_(-(double)n, gl_d(p, GL_SC_HPI), x);                         /* x - n*hpi (vfnmadd) */
}
/* reduce_large: 120 <= |x| < inf, integer multiplication by 4/pi */
static inline double gl_reduce_large(uint32_t xi, int *np) {
    const uint32_t *arr = &glibc_inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * arr[0]);
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * gl_d(glibc_pi63, 0);
}

/* s_cosf.c */
static inline float gl_cosf(float y) {
    double x = y;
    int n;
    const uint64_t *p = glibc_sincosf_tab;
    if (gl_abstop12(y) < 0x3f4 /* abstop12(pi/4) */) {
        const double x2 = x * x;
        if (gl_abstop12(y) < 0x398 /* abstop12(0x1p-12f) */) return 1.0f;
        return gl_sinf_poly(x, x2, p, 1);
    } else if (gl_abstop12(y) < 0x42f /* abstop12(120.0f) */) {
        x = gl_reduce_fast(x, p, &n);
        const double s = gl_d(p, n & 3);
        if (n & 2) p = glibc_sincosf_tab + GL_SC_STRIDE;
        return gl_sinf_poly(x * s, x * x, p, n ^ 1);
    } else if (gl_abstop12(y) < 0x7f8 /* abstop12(inf) */) {
        const uint32_t xi = gl_asuint(y);
        const int sign = (int)(xi >> 31);
        x = gl_reduce_large(xi, &n);
        const double s = gl_d(p, (n + sign) & 3);
        if ((n + sign) & 2) p = glibc_sincosf_tab + GL_SC_STRIDE;
        return gl_sinf_poly(x * s, x * x, p, n ^ 1);
    }
    return gl_invalidf(y);
}
/* s_sinf.c */
static inline float gl_sinf(float y) {
    double x = y;
    int n;
    const uint64_t *p = glibc_sincosf_tab;
    if (gl_abstop12(y) < 0x3f4) {
        const double x2 = x * x;
        if (gl_abstop12(y) < 0x398) return y;          /* (the source also forces an underflow exception for subnormals) */
        return gl_sinf_poly(x, x2, p, 0);
    } else if (gl_abstop12(y) < 0x42f) {
        x = gl_reduce_fast(x, p, &n);
        const double s = gl_d(p, n & 3);
        if (n & 2) p = glibc_sincosf_tab + GL_SC_STRIDE;
        return gl_sinf_poly(x * s, x * x, p, n);
    } else if (gl_abstop12(y) < 0x7f8) {
        const uint32_t xi = gl_asuint(y);
        const int sign = (int)(xi >> 31);
        x = gl_reduce_large(xi, &n);
        const double s = gl_d(p, (n + sign) & 3);
        if ((n + sign) & 2) p = glibc_sincosf_tab + GL_SC_STRIDE;
        return gl_sinf_poly(x * s, x * x, p, n);
    }
    return gl_invalidf(y);
}

/* ---- e_logf.c --------------------------------------------------------------------------------------------------- */
static inline float gl_logf(float x) {
    uint32_t ix = gl_asuint(x);
    if (ix == 0x3f800000) return 0.0f;
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
        if (ix * 2 == 0) return gl_divzerof(1);                       /* log(+-0) = -inf */
        if (ix == 0x7f800000) return x;                               /* log(inf) = inf */
        if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return gl_invalidf(x);
        ix = gl_asuint(x * 0x1p23f);                                   /* subnormal: normalise */
        ix -= 23 << 23;
    }
    const uint32_t tmp = ix - 0x3f330000;
    const int i = (tmp >> 19) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000);
    const double invc = gl_d(glibc_logf_tab, 2 * i), logc = gl_d(glibc_logf_tab, 2 * i + 1);
    const double z = (double)gl_asfloat(iz);
    const double r = fma_(z, invc, -1.0);                              /* z*invc - 1 */
    const double y0 = fma_((double)k, gl_d(glibc_logf_ln2, 0), logc);  /* logc + k*Ln2 */
    const double r2 = r * r;
    double y = fma_(gl_d(glibc_logf_poly, 1), r, gl_d(glibc_logf_poly, 2));   /* A1*r + A2 */
    y = fma_(gl_d(glibc_logf_poly, 0), r2, y);                         /* A0*r2 + y */
    y = fma_(y, r2, y0 + r);                                           /* y*r2 + (y0 + r) */
    return (float)y;
}

/* ---- e_log10f.c (__ieee754_log10f; binary32 arithmetic, never fused: the file has no FMA variant) -------------- */
static inline float gl_log10f(float x) {
    const float two25 = gl_asfloat(glibc_log10f_consts[0]), log10_2lo = gl_asfloat(glibc_log10f_consts[1]);
    const float ivln10 = gl_asfloat(glibc_log10f_consts[2]), log10_2hi = gl_asfloat(glibc_log10f_consts[3]);
    int32_t hx = (int32_t)gl_asuint(x), k = 0, i;
    if (hx < 0x00800000) {                                             /* x < 2^-126 */
        if ((hx & 0x7fffffff) == 0) return -two25 / __builtin_fabsf(x);   /* log(+-0) = -inf */
        if (hx < 0) return (x - x) / (x - x);                          /* log(-#) = NaN */
        k -= 25;
        x *= two25;                                                    /* subnormal, scale up */
        hx = (int32_t)gl_asuint(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    x = gl_asfloat((uint32_t)hx);
    const float z = y * log10_2lo + ivln10 * gl_logf(x);
    return z + y * log10_2hi;
}

/* ---- e_powf.c --------------------------------------------------------------------------------------------------- */
static inline double gl_pow_log2_inline(uint32_t ix) {
    const uint32_t tmp = ix - 0x3f330000;
    const int i = (tmp >> 19) % 16;
    const uint32_t top = tmp & 0xff800000;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = gl_d(glibc_pow_log2_tab, 2 * i), logc = gl_d(glibc_pow_log2_tab, 2 * i + 1);
    const double z = (double)gl_asfloat(iz);
    const double r = fma_(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = fma_(gl_d(glibc_pow_log2_poly, 0), r, gl_d(glibc_pow_log2_poly, 1));
    const double p = fma_(gl_d(glibc_pow_log2_poly, 2), r, gl_d(glibc_pow_log2_poly, 3));
    const double r4 = r2 * r2;
    double q = fma_(gl_d(glibc_pow_log2_poly, 4), r, y0);
    q = fma_(p, r2, q);
    y = fma_(y, r4, q);
    return y;
}
static inline float gl_exp2_inline(double xd, uint32_t sign_bias) {
    const double SHIFT = gl_d(glibc_exp2f_shift_scaled, 0);
    double kd = xd + SHIFT;
    const uint64_t ki = gl_asuint64(kd);
    kd -= SHIFT;
    const double r = xd - kd;
    uint64_t t = glibc_exp2f_tab[ki % 32];
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = gl_asdouble(t);
    const double z = fma_(gl_d(glibc_exp2f_poly, 0), r, gl_d(glibc_exp2f_poly, 1));
    const double r2 = r * r;
    double y = fma_(gl_d(glibc_exp2f_poly, 2), r, 1.0);
    y = fma_(z, r2, y);
    y = y * s;
    return (float)y;
}
static inline int gl_checkint(uint32_t iy) {       /* 0: not an integer, 1: odd, 2: even */
    const int e = iy >> 23 & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
static inline int gl_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }
static inline int gl_issignalingf(float x) {
    const uint32_t ix = gl_asuint(x);
    return 2 * (ix ^ 0x00400000) > 2u * 0x7fc00000u;
}
static inline float gl_powf(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = gl_asuint(x), iy = gl_asuint(y);
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000 || gl_zeroinfnan(iy)) {
        if (gl_zeroinfnan(iy)) {
            if (2 * iy == 0) return gl_issignalingf(x) ? x + y : 1.0f;
            if (ix == 0x3f800000) return gl_issignalingf(y) ? x + y : 1.0f;
            if (2 * ix > 2u * 0x7f800000 || 2 * iy > 2u * 0x7f800000) return x + y;
            if (2 * ix == 2 * 0x3f800000) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000) == !(iy & 0x80000000)) return 0.0f;   /* |x|<1 && y==inf or |x|>1 && y==-inf */
            return y * y;
        }
        if (gl_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000) && gl_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
            if (2 * ix == 0 && (iy & 0x80000000)) return gl_divzerof(sign_bias);
            return (iy & 0x80000000) ? 1 / x2 : x2;
        }
        if (ix & 0x80000000) {                                         /* finite x < 0 */
            const int yint = gl_checkint(iy);
            if (yint == 0) return gl_invalidf(x);
            if (yint == 1) sign_bias = 1u << 16;                       /* SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11) */
            ix &= 0x7fffffff;
        }
        if (ix < 0x00800000) {                                         /* subnormal x */
            ix = gl_asuint(x * 0x1p23f);
            ix &= 0x7fffffff;
            ix -= 23 << 23;
        }
    }
    const double logx = gl_pow_log2_inline(ix);
    const double ylogx = (double)y * logx;
    if ((gl_asuint64(ylogx) >> 47 & 0xffff) >= 0x80bf /* asuint64(126.0) >> 47 */) {
        if (ylogx > 0x1.fffffffd1d571p+6) return gl_oflowf(sign_bias);
        if (ylogx <= -150.0) return gl_uflowf(sign_bias);
        if (ylogx < -149.0) return gl_may_uflowf(sign_bias);
    }
    return gl_exp2_inline(ylogx, sign_bias);
}

/* ---- e_expf.c --------------------------------------------------------------------------------------------------- */
static inline float gl_expf(float x) {
    const double xd = (double)x;
    const uint32_t abstop = gl_abstop12(x);
    if (abstop >= 0x42b /* abstop12(88.0f) */) {
        if (gl_asuint(x) == 0xff800000u) return 0.0f;                  /* exp(-inf) */
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return gl_oflowf(0);                    /* x > log(0x1p128) */
        if (x < -0x1.9fe368p6f) return gl_uflowf(0);                   /* x < log(0x1p-150) */
        if (x < -0x1.9d1d9ep6f) return gl_may_uflowf(0);               /* x < log(0x1p-149) */
    }
    const double InvLn2N = gl_d(glibc_exp2f_invln2_scaled, 0), SHIFT = gl_d(glibc_exp2f_shift, 0);
    double kd = fma_(InvLn2N, xd, SHIFT);                              /* z + SHIFT with z = InvLn2N*xd, fused */
    const uint64_t ki = gl_asuint64(kd);
    kd -= SHIFT;
    const double r = fma_(InvLn2N, xd, -kd);                           /* z - kd, fused (vfmsub) */
    uint64_t t = glibc_exp2f_tab[ki % 32];
    t += ki << (52 - 5);
    const double s = gl_asdouble(t);
    const double z = fma_(gl_d(glibc_exp2f_poly_scaled, 0), r, gl_d(glibc_exp2f_poly_scaled, 1));
    const double r2 = r * r;
    double y = fma_(gl_d(glibc_exp2f_poly_scaled, 2), r, 1.0);
    y = fma_(z, r2, y);
    y = y * s;
    return (float)y;
}

// pt_device_math.h -- gfx950 device arithmetic for the path tracer.
//
// Every function is one IEEE-754 binary32 (or, in the transcendental shim, binary64)
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

// ---- deterministic transcendental shim (same spec as the CPU oracle's) ----
// IEEE binary64 + - * / only, fixed order, result rounded once to binary32.
__device__ __forceinline__ double shim_ksin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S6;
    p = S5 + z * p; p = S4 + z * p; p = S3 + z * p; p = S2 + z * p; p = S1 + z * p;
    return r + (r * z) * p;
}
__device__ __forceinline__ double shim_kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = C6;
    p = C5 + z * p; p = C4 + z * p; p = C3 + z * p; p = C2 + z * p; p = C1 + z * p;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
__device__ __noinline__ float shim_cosf(float x) {
    if (!(fabsf(x) <= 1048576.0f)) return x - x;
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    double xd = (double)x;
    double kf = floor(xd * INV_PIO2 + 0.5);
    double r = (xd - kf * PIO2_1) - kf * PIO2_1T;
    long long k = (long long)kf;
    double s = shim_ksin(r), c = shim_kcos(r);
    double v;
    switch (k & 3) {
    case 0: v = c; break;
    case 1: v = -s; break;
    case 2: v = -c; break;
    default: v = s; break;
    }
    return (float)v;
}
__device__ __noinline__ float shim_sinf(float x) {           // same reduction as shim_cosf; sin(x) = {s, c, -s, -c}[k & 3]
    if (!(fabsf(x) <= 1048576.0f)) return x - x;
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    double xd = (double)x;
    double kf = floor(xd * INV_PIO2 + 0.5);
    double r = (xd - kf * PIO2_1) - kf * PIO2_1T;
    long long k = (long long)kf;
    double s = shim_ksin(r), c = shim_kcos(r);
    double v;
    switch (k & 3) {
    case 0: v = s; break;
    case 1: v = c; break;
    case 2: v = -s; break;
    default: v = -c; break;
    }
    return (float)v;
}
__device__ __forceinline__ double shim_log_reduce(double xd, double &e_out) {
    const double SQRT2 = 1.41421356237309514547e+00;
    uint64_t b = (uint64_t)__double_as_longlong(xd);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > SQRT2) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 4.34782608695652161754e-02;
    p = 4.76190476190476164085e-02 + z * p;
    p = 5.26315789473684181249e-02 + z * p;
    p = 5.88235294117647050660e-02 + z * p;
    p = 6.66666666666666657415e-02 + z * p;
    p = 7.69230769230769273453e-02 + z * p;
    p = 9.09090909090909116141e-02 + z * p;
    p = 1.11111111111111104943e-01 + z * p;
    p = 1.42857142857142849213e-01 + z * p;
    p = 2.00000000000000011102e-01 + z * p;
    p = 3.33333333333333314830e-01 + z * p;
    p = 1.0 + z * p;
    e_out = (double)e;
    return (2.0 * s) * p;
}
__device__ __noinline__ float shim_log10f(float x) {
    if (x != x) return x;
    if (x < 0.0f) return (x - x) / (x - x);
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    const double LOG10_2 = 3.01029995663981198017e-01;
    const double INV_LN10 = 4.34294481903251816668e-01;
    double e, lm = shim_log_reduce((double)x, e);
    return (float)(e * LOG10_2 + lm * INV_LN10);
}
__device__ __forceinline__ double shim_exp(double z) {
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = floor(z * INV_LN2 + 0.5);
    double r = (z - kf * LN2_HI) - kf * LN2_LO;
    double p = 1.60590438368216133e-10;
    p = 2.08767569878680989792e-09 + r * p;
    p = 2.50521083854417187751e-08 + r * p;
    p = 2.75573192239858906526e-07 + r * p;
    p = 2.75573192239858906526e-06 + r * p;
    p = 2.48015873015873015873e-05 + r * p;
    p = 1.98412698412698412698e-04 + r * p;
    p = 1.38888888888888894189e-03 + r * p;
    p = 8.33333333333333321769e-03 + r * p;
    p = 4.16666666666666643537e-02 + r * p;
    p = 1.66666666666666657415e-01 + r * p;
    p = 0.5 + r * p;
    p = 1.0 + r * p;
    p = 1.0 + r * p;
    long long k = (long long)kf;
    double scale = __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
    return p * scale;
}
__device__ __noinline__ float shim_expf(float x) {
    if (x != x) return x;
    if (x > 100.0f) return __builtin_inff();
    if (x < -110.0f) return 0.0f;
    return (float)shim_exp((double)x);
}
__device__ __noinline__ float shim_powf(float x, float y) {
    const float INF = __builtin_inff();
    if (y == 0.0f || x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return (x - x) / (x - x);
    if (x == 0.0f) return y > 0.0f ? 0.0f : INF;
    if (x == INF) return y > 0.0f ? INF : 0.0f;
    if (y == INF) return x > 1.0f ? INF : 0.0f;
    if (y == -INF) return x > 1.0f ? 0.0f : INF;
    const double LN2 = 6.93147180559945286227e-01;
    double e, lm = shim_log_reduce((double)x, e);
    double z = (double)y * (e * LN2 + lm);
    if (z > 100.0) return INF;
    if (z < -110.0) return 0.0f;
    return (float)shim_exp(z);
}

// math.rs:15-19 (log10, not ln: SURVEY T3)
__device__ __forceinline__ float rand_f32_nd(uint32_t &s) {
    float theta = 6.283185f * rand_f32(s);
    float rho = __builtin_sqrtf(-2.0f * shim_log10f(rand_f32(s)));
    return rho * shim_cosf(theta);
}
// vec3.rs:66-68 -- x, y, z drawn in that order
__device__ __forceinline__ V3 rand_in_unit_sphere(uint32_t &s) {
    float x = rand_f32_nd(s);
    float y = rand_f32_nd(s);
    float z = rand_f32_nd(s);
    return normalized(mk(x, y, z));
}

// vec3.rs:80-90 (+ mix :197-205) and :262-270, one channel
__device__ __forceinline__ uint32_t srgb_quantize(float c) {
    float cutoff = (c < 0.0031308f) ? 1.0f : 0.0f;
    float higher = 1.055f * shim_powf(c, 1.0f / 2.4f) - 0.055f;
    float lower = c * 12.92f;
    float s = (higher * (1.0f - cutoff)) + lower * cutoff;
    float q = floorf(s * 255.0f);
    if (q < 0.0f) q = 0.0f;
    if (q > 255.0f) q = 255.0f;
    return (q != q) ? 0u : (uint32_t)q;          // Rust `as u8`: NaN -> 0
}

} // namespace mipt

// mipt_diag.hip -- libmipt_diag.so: device-arithmetic probe for the GPU known-answer tests (include/mipt_diag.h).
//
// NOT part of the product library: libmipt.so exports nothing from this file.  The probe evaluates the kernel's own
// arithmetic building blocks (pt_device_math.h: the glibc 2.35 restatement, the exact per-ray division, RNG, sRGB
// quantisation) element-wise, so tests can compare them bit for bit with the CPU oracle on millions of arguments.
#include "../../include/mipt_diag.h"
#include "pt_device_math.h"

#include <stdio.h>

namespace {

thread_local char g_err[256] = "";

__global__ void debug_eval_kernel(int op, const float *__restrict__ a, const float *__restrict__ b, float b_scalar,
                                  unsigned long long n, float *__restrict__ out) {
    using namespace mipt;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = a[i], y = b ? b[i] : b_scalar;
        float r = 0.0f;
        switch (op) {
        case 0: r = gl_cosf(x); break;
        case 1: r = gl_log10f(x, GlTabGlobal()); break;
        case 2: r = gl_powf(x, y); break;
        case 3: r = x / y; break;
        case 4: r = __builtin_sqrtf(x); break;
        case 5: r = x * y; break;
        case 6: r = x + y; break;
        case 7: r = fminf(x, y); break;
        case 8: r = fmaxf(x, y); break;
        case 9: { uint32_t s = __float_as_uint(x); r = rand_f32(s); } break;             // xorshift + u32->f32 + /2^32
        case 10: { uint32_t s = __float_as_uint(x); r = rand_f32_nd(s, GlTabGlobal()); } break;
        case 11: { uint32_t s = __float_as_uint(x); V3 v = rand_in_unit_sphere(s, GlTabGlobal()); r = (y == 0.0f) ? v.x : (y == 1.0f ? v.y : v.z); } break;
        case 12: r = __uint_as_float(srgb_quantize(x)); break;
        case 13: r = x - truncf(x); break;
        case 14: r = fdiv_ray(x, y, 1.0f / y); break;
        case 15: r = u8_over_255(__float_as_uint(x)); break;                              // exact-division helper (valid range only)
        case 16: r = gl_sinf(x); break;
        case 17: r = gl_expf(x); break;
        case 18: r = gl_logf(x, GlTabGlobal()); break;
        case 19: r = gl_log10f_unit(x, GlTabGlobal()); break;                              // valid on {0} u [2^-32, 1]
        case 20: r = gl_cosf_2pi(x); break;                                                // valid on [0, 6.2831855]
        case 23: r = __uint_as_float(floor_mod((int32_t)__float_as_uint(x), __float_as_uint(y))); break;   // |i| < 2^30, W >= 1
        default: break;
        }
        out[i] = r;
    }
}

// inputs are the consecutive bit patterns first_bits + i (no input array): the exhaustive sweeps
__global__ void debug_eval_range_kernel(int op, uint32_t first_bits, float y, unsigned long long n, float *__restrict__ out) {
    using namespace mipt;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first_bits + (uint32_t)i);
        float r = 0.0f;
        switch (op) {
        case 0: r = gl_cosf(x); break;
        case 1: r = gl_log10f(x, GlTabGlobal()); break;
        case 2: r = gl_powf(x, y); break;
        case 16: r = gl_sinf(x); break;
        case 17: r = gl_expf(x); break;
        case 18: r = gl_logf(x, GlTabGlobal()); break;
        case 19: r = gl_log10f_unit(x, GlTabGlobal()); break;
        case 20: r = gl_cosf_2pi(x); break;
        default: break;
        }
        out[i] = r;
    }
}

struct DevBuf {                       // frees on every exit path
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

int fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -2;   // MIPT_ERR_HIP
}

} // namespace

extern "C" {

int mipt_debug_eval(int op, const float *a, const float *b, uint64_t n, float *out) {
    if (!a || !out || n == 0) { snprintf(g_err, sizeof g_err, "mipt_debug_eval: bad argument"); return -1; }
    DevBuf da, db, dout;
    hipError_t e;
    if ((e = hipMalloc(&da.p, n * 4)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMalloc(&dout.p, n * 4)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemcpy(da.p, a, n * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    if (b) {
        if ((e = hipMalloc(&db.p, n * 4)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMemcpy(db.p, b, n * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy");
    }
    hipLaunchKernelGGL(debug_eval_kernel, dim3(1024), dim3(256), 0, nullptr, op, (const float *)da.p, (const float *)db.p, 0.0f,
                       (unsigned long long)n, (float *)dout.p);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
    if ((e = hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    return 0;
}

int mipt_debug_eval_range(int op, uint32_t first_bits, uint64_t n, float y, float *out) {
    if (!out || n == 0 || (uint64_t)first_bits + n > (1ull << 32)) { snprintf(g_err, sizeof g_err, "mipt_debug_eval_range: bad argument"); return -1; }
    DevBuf dout;
    hipError_t e;
    if ((e = hipMalloc(&dout.p, n * 4)) != hipSuccess) return fail(e, "hipMalloc");
    hipLaunchKernelGGL(debug_eval_range_kernel, dim3(2048), dim3(256), 0, nullptr, op, first_bits, y, (unsigned long long)n, (float *)dout.p);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
    if ((e = hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    return 0;
}

const char *mipt_diag_last_error(void) { return g_err; }

} // extern "C"

// scene_hooks.hip -- part of libmipt_diag.so (test infrastructure): reads back / fingerprints the device layout a MiptScene holds,
// so tests can compare the layout built by the GPU kernels of scene_device.hip with the one mipt_scene_create's host code builds
// (tests/test_gpu_scene_device.py).  Sees the opaque handle through the library-internal header; nothing of this is in libmipt.so.
#include "../../include/mipt_diag.h"
#include "../../rust_ray_tracing_amd/csrc/mipt_scene.h"

#include <hip/hip_runtime.h>

namespace {
// order-dependent 64-bit fingerprint of a word stream: sum (w_i + c) * (2 i + 1) mod 2^64 -- changes with any word and with any
// swap of two different words
__global__ void hash_words(const uint32_t *w, unsigned long long n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        acc += ((unsigned long long)w[i] + 0x9E3779B97F4A7C15ull) * (2ull * i + 1ull);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63u) == 0u) atomicAdd(out, acc);
}
} // namespace

extern "C" {

int mipt_diag_scene_sizes(const void *scene, uint64_t out[2]) {
    const MiptScene *s = (const MiptScene *)scene;
    if (!s || !out) return -1;
    out[0] = s->dev.geom_bytes; out[1] = s->attr_bytes;
    return 0;
}

int mipt_diag_scene_read(const void *scene, int which, void *dst, uint64_t bytes) {
    const MiptScene *s = (const MiptScene *)scene;
    if (!s || !dst || which < 0 || which > 1) return -1;
    const void *src = which == 0 ? s->d_geom : s->d_tri_attr;
    const uint64_t have = which == 0 ? s->dev.geom_bytes : s->attr_bytes;
    if (bytes > have) return -1;
    if (hipSetDevice(s->device) != hipSuccess || hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return 0;
}

int mipt_diag_scene_hash(const void *scene, uint64_t out[2]) {
    const MiptScene *s = (const MiptScene *)scene;
    if (!s || !out) return -1;
    unsigned long long *d = nullptr;
    if (hipSetDevice(s->device) != hipSuccess || hipMalloc((void **)&d, 16) != hipSuccess) return -2;
    hipError_t e = hipMemset(d, 0, 16);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hash_words, dim3(2048), dim3(256), 0, nullptr, (const uint32_t *)s->d_geom, (unsigned long long)s->dev.geom_bytes / 4, d);
        hipLaunchKernelGGL(hash_words, dim3(2048), dim3(256), 0, nullptr, (const uint32_t *)s->d_tri_attr, (unsigned long long)s->attr_bytes / 4, d + 1);
        e = hipMemcpy(out, d, 16, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    return e == hipSuccess ? 0 : -2;
}

} // extern "C"

// mipt_multi.cpp -- multi-GPU rendering behind the C ABI: mipt_multi_create / mipt_render_multi / mipt_multi_destroy.
//
// The reference's host is ONE process holding one Rc<RefCell<Scene>> (reference src/main.rs:46, src/renderer.rs:50-63),
// so the multi-GPU arm of its `match backend` has to be a single call.  One process drives all devices of the node:
// a scene replica per GPU (<= 1.9 GB for 10 M triangles against 288 GB of HBM), one RCCL communicator per GPU
// (ncclCommInitAll), one host thread per GPU for the render, and exactly ONE collective per frame over xGMI:
//   MIPT_MULTI_TILES    8x8 image tiles dealt round-robin; each GPU writes a rank-packed slice; ncclGather to device 0;
//                       de-interleave kernel.  The seed depends only on the pixel index (cpu.rs:28-29), so the frame is
//                       bit-identical to the single-GPU frame.
//   MIPT_MULTI_SAMPLES  every GPU renders all pixels for a disjoint sample range with the per-sample seeds of
//                       rt_compute.wgsl:102 and writes un-normalised sums; ncclReduce(sum, f32) to device 0; one divide.
// xGMI is point-to-point, so a gather to the root moves each slice over its own link (3.1 MB per GPU for a 1080p f32
// frame); the reduce of a 4096^2 frame is 201 MB per GPU through RCCL's ring/tree.  Either is noise next to the trace.
#include "../../include/mipt.h"
#include "pt_kernel.h"
#include "mipt_internal.h"

#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>


struct MiptMulti {
    int n = 0;
    std::vector<int> devices;
    std::vector<MiptScene *> scenes;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<float *> d_part;          // per device: packed tile slice / full-frame partial sum
    std::vector<size_t> part_floats;
    float *d_all = nullptr;               // root: gathered slices
    size_t all_floats = 0;
    float *d_frame = nullptr;             // root: assembled frame
    size_t frame_floats = 0;
    uint8_t *d_rgba = nullptr;
    size_t rgba_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<MiptStats> last;          // per device: trace-kernel stats of the last mipt_render_multi* call
};

namespace {

// libmipt_multitest.so (make multitest; test infrastructure, never the product) compiles this file against the RCCL test double of
// tests/cpp/rccl_double/, whose header defines MIPT_RCCL_DOUBLE: there a HIP device may carry several logical ranks, so that the
// n > 1 code below runs on the one-GPU test box.  Real RCCL refuses a device listed twice, and so does the product.
#ifdef MIPT_RCCL_DOUBLE
constexpr bool kLogicalRanks = true;
#else
constexpr bool kLogicalRanks = false;
#endif

int fail(int code, const std::string &msg) {
    mipt_internal_set_error(msg.c_str());
    return code;
}
#define M_HIP(expr)                                                                                                    \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess) return fail(MIPT_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e__));     \
    } while (0)
#define M_NCCL(expr)                                                                                                   \
    do {                                                                                                               \
        ncclResult_t r__ = (expr);                                                                                     \
        if (r__ != ncclSuccess) return fail(MIPT_ERR_RCCL, std::string(#expr " failed: ") + ncclGetErrorString(r__)); \
    } while (0)

int grow(void **p, size_t *have, size_t want_bytes) {
    if (*p && *have >= want_bytes) return MIPT_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    M_HIP(hipMalloc(p, want_bytes));
    *have = want_bytes;
    return MIPT_OK;
}

int root_buffer_check(const void *p, int root_device, const char *name) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) (void)hipGetLastError();         // an unregistered host pointer is reported as an error: clear it
    if (e != hipSuccess || a.type != hipMemoryTypeDevice || a.device != root_device)
        return fail(MIPT_ERR_INVALID_ARG, std::string("mipt_render_multi_device: ") + name + " is not device memory of the root device " + std::to_string(root_device));
    return MIPT_OK;
}

void destroy(MiptMulti *m) {
    if (!m) return;
    for (int i = 0; i < (int)m->comms.size(); i++)
        if (m->comms[i]) (void)ncclCommDestroy(m->comms[i]);
    for (int i = 0; i < (int)m->devices.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        if (i < (int)m->streams.size() && m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
        if (i < (int)m->d_part.size() && m->d_part[i]) (void)hipFree(m->d_part[i]);
        if (i == 0) {
            if (m->d_all) (void)hipFree(m->d_all);
            if (m->d_frame) (void)hipFree(m->d_frame);
            if (m->d_rgba) (void)hipFree(m->d_rgba);
            if (m->ev0) (void)hipEventDestroy(m->ev0);
            if (m->ev1) (void)hipEventDestroy(m->ev1);
        }
        if (i < (int)m->scenes.size() && m->scenes[i]) mipt_scene_destroy(m->scenes[i]);
    }
    delete m;
}

int create_impl(const MiptSceneDesc *desc, const int *device_ids, int n_devices, MiptMulti **out, bool from_triangles) {
    if (!desc || !out) return fail(MIPT_ERR_INVALID_ARG, "mipt_multi_create: null argument");
    *out = nullptr;
    const int visible = mipt_device_count();
    if (visible < 0) return visible;
    if (n_devices == 0) n_devices = visible;                     // 0 = every visible device
    if (n_devices < 1 || (!kLogicalRanks && n_devices > visible) || n_devices > 64)
        return fail(MIPT_ERR_INVALID_ARG, "mipt_multi_create: n_devices " + std::to_string(n_devices) + " but " + std::to_string(visible) + " HIP device(s) visible");
    MiptMulti *m = new MiptMulti();
    m->n = n_devices;
    for (int i = 0; i < n_devices; i++) {
        const int d = device_ids ? device_ids[i] : i;
        if (d < 0 || d >= visible) { destroy(m); return fail(MIPT_ERR_INVALID_ARG, "mipt_multi_create: bad device id " + std::to_string(d)); }
        for (int j = 0; j < i && !kLogicalRanks; j++)
            if (m->devices[j] == d) { destroy(m); return fail(MIPT_ERR_INVALID_ARG, "mipt_multi_create: device listed twice"); }
        m->devices.push_back(d);
    }
    m->scenes.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    m->d_part.assign(n_devices, nullptr);
    m->part_floats.assign(n_devices, 0);
    m->last.assign(n_devices, MiptStats{});
    // scene replicas: the scene reaches device 0 once (host layout + upload, or built there from the triangles); the others are
    // device-to-device copies
    {
        const int rc = mipt::scene_create_replicas(desc, m->devices.data(), n_devices, m->scenes.data(), from_triangles);
        if (rc) { const std::string e = mipt_last_error(); destroy(m); return fail(rc, e); }
    }
    for (int i = 0; i < n_devices; i++) {
        hipError_t e = hipSetDevice(m->devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) { destroy(m); return fail(MIPT_ERR_HIP, std::string("stream creation failed: ") + hipGetErrorString(e)); }
    }
    (void)hipSetDevice(m->devices[0]);
    if (hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess) { destroy(m); return fail(MIPT_ERR_HIP, "event creation failed"); }
    m->comms.assign(n_devices, nullptr);
    ncclResult_t r = ncclCommInitAll(m->comms.data(), n_devices, m->devices.data());
    if (r != ncclSuccess) { destroy(m); return fail(MIPT_ERR_RCCL, std::string("ncclCommInitAll failed: ") + ncclGetErrorString(r)); }
    *out = m;
    return MIPT_OK;
}

// Every stream of `m` drained, errors ignored: after a failure between ncclGroupStart and the final synchronisation the other
// ranks' streams may still hold their part of the collective, and the next call reuses d_part / d_all.
void drain(MiptMulti *m) {
    for (int i = 0; i < m->n; i++) {
        if (hipSetDevice(m->devices[i]) == hipSuccess && m->streams[i]) (void)hipStreamSynchronize(m->streams[i]);
    }
    (void)hipSetDevice(m->devices[0]);
}

// device_out: hdr_rgb / rgba8 are buffers in DEVICE 0's memory (mipt_render_multi_device) -- the assemble kernels write them
// directly and nothing is copied to the host; otherwise host buffers (mipt_render_multi).
int render_impl(MiptMulti *m, const MiptCamera *camera, const MiptOptions *opt, uint32_t mode, float *hdr_rgb, uint8_t *rgba8,
                MiptMultiStats *stats, bool device_out) {
    if (!m || !camera || !opt) return fail(MIPT_ERR_INVALID_ARG, "mipt_render_multi: null argument");
    if (device_out && !hdr_rgb) return fail(MIPT_ERR_INVALID_ARG, "mipt_render_multi_device: d_hdr_rgb == NULL");
    if (device_out) {                                     // a host pointer or another device's buffer would fault inside the assemble kernels
        int rc = root_buffer_check(hdr_rgb, m->devices[0], "d_hdr_rgb");
        if (rc == MIPT_OK && rgba8) rc = root_buffer_check(rgba8, m->devices[0], "d_rgba8");
        if (rc) return rc;
    }
    if (mode > MIPT_MULTI_SAMPLES) return fail(MIPT_ERR_INVALID_ARG, "mipt_render_multi: unknown mode");
    if (opt->tile_rank || opt->tile_world > 1 || opt->sample_begin > 1 ||
        (opt->flags & (MIPT_FLAG_PACKED | MIPT_FLAG_SUM | MIPT_FLAG_ACCUM)))
        return fail(MIPT_ERR_INVALID_ARG, "mipt_render_multi owns the sharding: tile_rank/tile_world/sample_begin and the PACKED/SUM/ACCUM flags must be 0");
    if (opt->width == 0 || opt->height == 0 || opt->samples == 0)
        return fail(MIPT_ERR_INVALID_ARG, "Width, height and sample count must be greater than 0");
    const auto t_wall0 = std::chrono::steady_clock::now();
    const int n = m->n;
    const uint64_t n_pix = (uint64_t)opt->width * opt->height;
    const uint64_t slots = mipt_packed_pixels(opt->width, opt->height, (uint32_t)n);
    const size_t part_floats = (size_t)(mode == MIPT_MULTI_TILES ? slots : n_pix) * 3;

    for (int i = 0; i < n; i++) {
        M_HIP(hipSetDevice(m->devices[i]));
        size_t have = m->part_floats[i] * sizeof(float);
        int rc = grow((void **)&m->d_part[i], &have, part_floats * sizeof(float));
        if (rc) return rc;
        m->part_floats[i] = have / sizeof(float);
    }
    M_HIP(hipSetDevice(m->devices[0]));
    {
        size_t have = m->all_floats * sizeof(float);
        int rc = grow((void **)&m->d_all, &have, (mode == MIPT_MULTI_TILES ? (size_t)n * part_floats : 4) * sizeof(float));
        if (rc) return rc;
        m->all_floats = have / sizeof(float);
        if (!device_out) {
            have = m->frame_floats * sizeof(float);
            if ((rc = grow((void **)&m->d_frame, &have, (size_t)n_pix * 3 * sizeof(float)))) return rc;
            m->frame_floats = have / sizeof(float);
            if (rgba8 && (rc = grow((void **)&m->d_rgba, &m->rgba_bytes, (size_t)n_pix * 4))) return rc;
        }
    }
    float *const d_frame = device_out ? hdr_rgb : m->d_frame;       // where the assembled frame is built (device 0)
    uint8_t *const d_rgba = device_out ? rgba8 : m->d_rgba;

    // ---- per-device options ----
    std::vector<MiptOptions> opts(n, *opt);
    std::vector<uint32_t> share(n, 0);
    if (mode == MIPT_MULTI_TILES) {
        for (int i = 0; i < n; i++) {
            opts[i].tile_rank = (uint32_t)i; opts[i].tile_world = (uint32_t)n;
            opts[i].flags |= MIPT_FLAG_PACKED;
            share[i] = opt->samples;
        }
    } else {
        // sample ranges [begin, begin + count): sample numbers start at 1 (gpu.rs:252), the first `rem` devices take one more
        const uint32_t base = opt->samples / (uint32_t)n, rem = opt->samples % (uint32_t)n;
        uint32_t begin = 1;
        for (int i = 0; i < n; i++) {
            share[i] = base + ((uint32_t)i < rem ? 1u : 0u);
            opts[i].seed_mode = MIPT_SEED_PER_SAMPLE;
            opts[i].flags |= MIPT_FLAG_SUM;
            opts[i].sample_begin = begin;
            opts[i].samples = share[i] ? share[i] : 1u;
            begin += share[i];
        }
    }

    // ---- render: one host thread per device, each blocking on its own stream ----
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::vector<MiptStats> st(n);
    bool spawn_failed = false;
    {
        std::vector<std::thread> th;
        th.reserve((size_t)n);
        auto work = [&](int i) {
                memset(&st[i], 0, sizeof(MiptStats));
                if (share[i] == 0) {                                      // more devices than samples: contributes zeros
                    hipError_t e = hipSetDevice(m->devices[i]);
                    if (e == hipSuccess) e = hipMemsetAsync(m->d_part[i], 0, part_floats * sizeof(float), m->streams[i]);
                    if (e == hipSuccess) e = hipStreamSynchronize(m->streams[i]);
                    if (e != hipSuccess) { rcs[i] = MIPT_ERR_HIP; errs[i] = hipGetErrorString(e); }
                    return;
                }
                rcs[i] = mipt::render_device_impl(m->scenes[i], camera, &opts[i], m->d_part[i], nullptr, (void *)m->streams[i], &st[i], true);
                if (rcs[i]) errs[i] = mipt_last_error();
        };
        // a std::thread constructor that throws (EAGAIN) must not unwind past joinable threads: join what was started
        try {
            for (int i = 1; i < n; i++) th.emplace_back(work, i);
        } catch (const std::exception &) { spawn_failed = true; }
        if (!spawn_failed) work(0);                                       // device 0 on the calling thread
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < n; i++) m->last[i] = st[i];
    if (spawn_failed) return fail(MIPT_ERR_HIP, "mipt_render_multi: could not start a host thread per device");
    int soft = MIPT_OK;                                                   // MIPT_ERR_STACK: frame incomplete but delivered, like mipt_render
    for (int i = 0; i < n; i++) {
        if (rcs[i] == MIPT_ERR_STACK) { soft = MIPT_ERR_STACK; continue; }
        if (rcs[i]) return fail(rcs[i], "device " + std::to_string(i) + ": " + errs[i]);
    }

    // ---- the one collective, then assemble on device 0 ----
    // From here on every rank's stream may hold work that touches d_part / d_all: any failure drains all streams before returning.
#define M_HIP_D(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess) { drain(m); return fail(MIPT_ERR_HIP, std::string(#expr " failed: ") + hipGetErrorString(e__)); } \
    } while (0)
    M_HIP(hipSetDevice(m->devices[0]));
    M_HIP(hipEventRecord(m->ev0, m->streams[0]));
    {
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n && r == ncclSuccess; i++) {
            if (mode == MIPT_MULTI_TILES)
                r = ncclGather(m->d_part[i], i == 0 ? m->d_all : nullptr, part_floats, ncclFloat, 0, m->comms[i], m->streams[i]);
            else
                r = ncclReduce(m->d_part[i], i == 0 ? d_frame : nullptr, part_floats, ncclFloat, ncclSum, 0, m->comms[i], m->streams[i]);
        }
        const ncclResult_t r_end = ncclGroupEnd();
        if (r == ncclSuccess) r = r_end;
        if (r != ncclSuccess) { drain(m); return fail(MIPT_ERR_RCCL, std::string("RCCL collective failed: ") + ncclGetErrorString(r)); }
    }
    M_HIP_D(hipSetDevice(m->devices[0]));
    if (mode == MIPT_MULTI_TILES)
        M_HIP_D(mipt::launch_unpack_tiles(m->d_all, opt->width, opt->height, (uint32_t)n, d_frame, m->streams[0]));
    else
        M_HIP_D(mipt::launch_divide(d_frame, (unsigned long long)n_pix * 3, (float)opt->samples, m->streams[0]));   // cpu.rs:60
    if (rgba8) M_HIP_D(mipt::launch_tonemap(d_frame, (unsigned long long)n_pix, 1.0f, d_rgba, m->streams[0]));
    M_HIP_D(hipEventRecord(m->ev1, m->streams[0]));
    if (!device_out) {
        if (hdr_rgb) M_HIP_D(hipMemcpyAsync(hdr_rgb, d_frame, (size_t)n_pix * 3 * sizeof(float), hipMemcpyDeviceToHost, m->streams[0]));
        if (rgba8) M_HIP_D(hipMemcpyAsync(rgba8, d_rgba, (size_t)n_pix * 4, hipMemcpyDeviceToHost, m->streams[0]));
    }
    for (int i = n - 1; i >= 0; i--) {                                    // every rank's part of the collective has drained
        M_HIP_D(hipSetDevice(m->devices[i]));
        M_HIP_D(hipStreamSynchronize(m->streams[i]));
    }
#undef M_HIP_D
    for (int i = 0; i < n; i++) {
        ncclResult_t async_err = ncclSuccess;
        M_NCCL(ncclCommGetAsyncError(m->comms[i], &async_err));
        if (async_err != ncclSuccess) return fail(MIPT_ERR_RCCL, std::string("RCCL asynchronous error: ") + ncclGetErrorString(async_err));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_devices = (uint32_t)n;
        float ms = 0.0f;
        M_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
        stats->collective_ms = ms;
        MiptStats &t = stats->total;
        for (int i = 0; i < n; i++) {
            if (st[i].kernel_ms > t.kernel_ms) t.kernel_ms = st[i].kernel_ms;
            t.rays += st[i].rays; t.inner_steps += st[i].inner_steps; t.tri_tests += st[i].tri_tests; t.hits += st[i].hits;
            t.texel_fetches += st[i].texel_fetches; t.stack_overflows += st[i].stack_overflows; t.tex_clamped += st[i].tex_clamped;
            if (st[i].max_stack > t.max_stack) t.max_stack = st[i].max_stack;
            t.pixels += st[i].pixels;
            if (i < 8) stats->device_kernel_ms[i] = st[i].kernel_ms;
        }
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall0).count();
    }
    if (soft) return fail(soft, "traversal stack overflowed on at least one device (result incomplete)");
    return MIPT_OK;
}

} // namespace

extern "C" {

int mipt_multi_create(const MiptSceneDesc *desc, const int *device_ids, int n_devices, MiptMulti **out) {
    try { return create_impl(desc, device_ids, n_devices, out, false); }
    catch (const std::exception &e) { return fail(MIPT_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); }
}

int mipt_multi_create_from_triangles(const MiptSceneDesc *desc, const int *device_ids, int n_devices, MiptMulti **out) {
    try { return create_impl(desc, device_ids, n_devices, out, true); }
    catch (const std::exception &e) { return fail(MIPT_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); }
}

MiptScene *mipt_multi_scene(MiptMulti *m, int index) { return (m && index >= 0 && index < m->n) ? m->scenes[(size_t)index] : nullptr; }

void mipt_multi_destroy(MiptMulti *m) { destroy(m); }

int mipt_multi_device_count(const MiptMulti *m) { return m ? m->n : 0; }

int mipt_render_multi(MiptMulti *m, const MiptCamera *camera, const MiptOptions *opt, uint32_t mode,
                      float *hdr_rgb, uint8_t *rgba8, MiptMultiStats *stats) {
    try { return render_impl(m, camera, opt, mode, hdr_rgb, rgba8, stats, false); }
    catch (const std::exception &e) { return fail(MIPT_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); }
}

int mipt_render_multi_device(MiptMulti *m, const MiptCamera *camera, const MiptOptions *opt, uint32_t mode,
                             float *d_hdr_rgb, uint8_t *d_rgba8, MiptMultiStats *stats) {
    try { return render_impl(m, camera, opt, mode, d_hdr_rgb, d_rgba8, stats, true); }
    catch (const std::exception &e) { return fail(MIPT_ERR_INVALID_ARG, std::string("internal error: ") + e.what()); }
}

int mipt_multi_device_stats(const MiptMulti *m, int index, MiptStats *out) {
    if (!m || !out || index < 0 || index >= m->n) return fail(MIPT_ERR_INVALID_ARG, "mipt_multi_device_stats: bad argument");
    *out = m->last[(size_t)index];
    return MIPT_OK;
}

int mipt_multi_root_device(const MiptMulti *m) { return m ? m->devices[0] : MIPT_ERR_INVALID_ARG; }

} // extern "C"

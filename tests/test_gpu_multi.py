"""-m gpu: mipt_render_multi -- all GPUs of the node behind ONE C-ABI call, one RCCL collective per frame
(north_star "single RCCL gather of the framebuffer over xGMI").  Runs with
n = mipt_device_count(): on the one-GPU box that is a one-rank communicator, which still drives ncclCommInitAll,
ncclGather / ncclReduce and the assemble kernels; on an 8-GPU node the same test covers eight ranks."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rrt, kind, **kw):
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    return sc


def _renderer(rrt, w, h, spp, depth, **kw):
    return rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h),
                                                output_image_path="/dev/null", **kw))


@pytest.mark.parametrize("w,h", [(128, 72), (61, 37)])
def test_tiles_mode_is_bit_identical_to_the_single_gpu_frame(rrt, orc, w, h):
    from rust_ray_tracing_amd import _lib as L
    n = rrt.load().mipt_device_count()
    sc = _scene(rrt, "atrium", n_target=20000, tex_size=32)
    r = _renderer(rrt, w, h, 4, 12)
    one_hdr, one_rgba, one_st = r.render_buffers(sc, flags=L.FLAG_COUNT)
    hdr, rgba, st = r.render_buffers_multi(sc, mode=L.MULTI_TILES, flags=L.FLAG_COUNT)
    assert st["n_devices"] == n and len(st["device_kernel_ms"]) == n
    assert np.array_equal(hdr.view(np.uint32), one_hdr.view(np.uint32))
    assert np.array_equal(rgba, one_rgba)
    for k in ("rays", "inner_steps", "tri_tests", "hits", "pixels"):
        assert st[k] == one_st[k], k
    assert st["collective_ms"] > 0.0 and st["wall_ms"] >= st["kernel_ms"]
    # and against the oracle (pixel-stream seeds: the CPU backend's own frame)
    ref, ref_rgba, _ = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, 4, 12)
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgba, ref_rgba)
    # a second frame through the same handle (buffers and communicators are reused)
    hdr2, _, _ = r.render_buffers_multi(sc, mode=L.MULTI_TILES)
    assert np.array_equal(hdr2.view(np.uint32), hdr.view(np.uint32))


def test_samples_mode_reduces_per_sample_partials(rrt, orc):
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import sharding
    n = rrt.load().mipt_device_count()
    sc = _scene(rrt, "atrium", n_target=20000, tex_size=32)
    w, h, spp, depth = 64, 36, 7, 8
    r = _renderer(rrt, w, h, spp, depth)
    hdr, rgba, st = r.render_buffers_multi(sc, mode=L.MULTI_SAMPLES, flags=L.FLAG_COUNT)
    # the oracle, summed over the same rank partition in rank order, then divided once (cpu.rs:60)
    m = sc.materials_array()
    total = np.zeros((h, w, 3), dtype=np.float32)
    rays = 0
    for (s0, cnt) in sharding.sample_ranges(spp, n):
        if cnt == 0:
            continue
        part, _, ost = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, cnt, depth, seed_mode=1,
                                  sample_begin=s0, sum_only=1, want_rgba8=False)
        total = total + part
        rays += ost["rays"]
    want = total / np.float32(spp)
    assert st["rays"] == rays and st["pixels"] == w * h * min(n, spp)
    if n <= 2:
        assert np.array_equal(hdr.view(np.uint32), want.view(np.uint32))           # one or two addends: the sum order cannot differ
    else:
        assert np.allclose(hdr, want, rtol=1e-6, atol=1e-6)                        # RCCL's reduction tree vs rank order
    # n = 1: identical to the single-GPU per-sample-seed render
    if n == 1:
        one, _, _ = _renderer(rrt, w, h, spp, depth, seed_mode=L.SEED_PER_SAMPLE).render_buffers(sc)
        assert np.array_equal(hdr.view(np.uint32), one.view(np.uint32))
    assert rgba.shape == (h, w, 4) and np.all(rgba[..., 3] == 255)


def test_explicit_device_list_and_argument_errors(rrt):
    from rust_ray_tracing_amd import _lib as L
    lib = rrt.load()
    sc = _scene(rrt, "cornell")
    multi = sc.upload_multi([0])
    assert lib.mipt_multi_device_count(multi) == 1
    opt = rrt.make_options(32, 32, 1, 2)
    out = np.zeros(32 * 32 * 3, dtype=np.float32)
    st = L.MiptMultiStats()
    cam = L.ptr(sc.camera.uniform)
    assert lib.mipt_render_multi(multi, cam, C.byref(opt), L.MULTI_TILES, L.ptr(out), None, C.byref(st)) == 0
    assert lib.mipt_render_multi(multi, cam, C.byref(opt), 2, L.ptr(out), None, None) == L.ERR_INVALID_ARG
    assert lib.mipt_render_multi(None, cam, C.byref(opt), 0, L.ptr(out), None, None) == L.ERR_INVALID_ARG
    for kw in (dict(tile_rank=1, tile_world=2), dict(flags=L.FLAG_SUM), dict(flags=L.FLAG_PACKED), dict(sample_begin=5)):
        bad = rrt.make_options(32, 32, 1, 2, **kw)
        assert lib.mipt_render_multi(multi, cam, C.byref(bad), 0, L.ptr(out), None, None) == L.ERR_INVALID_ARG, kw
        assert b"owns the sharding" in lib.mipt_last_error()
    h = C.c_void_p()
    d = sc.desc()
    assert lib.mipt_multi_create(C.byref(d), (C.c_int * 2)(0, 0), 2, C.byref(h)) == L.ERR_INVALID_ARG     # same device twice (or > visible)
    assert lib.mipt_multi_create(C.byref(d), None, 65, C.byref(h)) == L.ERR_INVALID_ARG


def test_render_multi_device_leaves_the_frame_on_the_root_device(rrt):
    """mipt_render_multi_device: same call, frame and RGBA8 written into caller-owned buffers of device 0 (no D2H inside the call);
    mipt_multi_device_stats returns each device's own launch."""
    import torch
    from rust_ray_tracing_amd import _lib as L
    lib = rrt.load()
    n = lib.mipt_device_count()
    sc = _scene(rrt, "atrium", n_target=20000, tex_size=32)
    w, h, spp, depth = 128, 72, 4, 12
    r = _renderer(rrt, w, h, spp, depth)
    ref_hdr, ref_rgba, ref_st = r.render_buffers(sc, flags=L.FLAG_COUNT)
    multi = sc.upload_multi(None)
    root = lib.mipt_multi_root_device(multi)
    assert root == 0
    d_hdr = torch.zeros(w * h * 3, dtype=torch.float32, device=f"cuda:{root}")
    d_rgba = torch.zeros(w * h * 4, dtype=torch.uint8, device=f"cuda:{root}")
    torch.cuda.synchronize()
    opt = rrt.make_options(w, h, spp, depth, flags=L.FLAG_COUNT)
    st = L.MiptMultiStats()
    L.check(lib.mipt_render_multi_device(multi, L.ptr(sc.camera.uniform), C.byref(opt), L.MULTI_TILES, C.c_void_p(d_hdr.data_ptr()),
                                         C.c_void_p(d_rgba.data_ptr()), C.byref(st)), "mipt_render_multi_device")
    assert np.array_equal(d_hdr.cpu().numpy().view(np.uint32), ref_hdr.reshape(-1).view(np.uint32))
    assert np.array_equal(d_rgba.cpu().numpy(), ref_rgba.reshape(-1))
    tot = st.as_dict()
    per = []
    for i in range(n):
        s_i = L.MiptStats()
        L.check(lib.mipt_multi_device_stats(multi, i, C.byref(s_i)), "mipt_multi_device_stats")
        per.append(s_i.as_dict())
    for k in ("rays", "inner_steps", "tri_tests", "hits", "pixels"):
        assert sum(p[k] for p in per) == tot[k] == ref_st[k], k
    assert lib.mipt_multi_device_stats(multi, n, C.byref(L.MiptStats())) == L.ERR_INVALID_ARG
    # NULL frame buffer is refused (the device entry has nowhere else to put the frame)
    assert lib.mipt_render_multi_device(multi, L.ptr(sc.camera.uniform), C.byref(opt), L.MULTI_TILES, None, None, None) == L.ERR_INVALID_ARG
    # samples mode into the same buffers
    L.check(lib.mipt_render_multi_device(multi, L.ptr(sc.camera.uniform), C.byref(opt), L.MULTI_SAMPLES, C.c_void_p(d_hdr.data_ptr()),
                                         None, C.byref(st)), "mipt_render_multi_device")
    hdr_s, _, _ = r.render_buffers_multi(sc, mode=L.MULTI_SAMPLES, want_rgba8=False)
    assert np.array_equal(d_hdr.cpu().numpy().view(np.uint32), hdr_s.reshape(-1).view(np.uint32))

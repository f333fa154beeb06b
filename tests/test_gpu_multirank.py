"""-m gpu: mipt_render_multi / mipt_render_multi_device with MORE THAN ONE RANK on the one-GPU box.

The product's multi-GPU arm (csrc/mipt_multi.cpp) is one blocking call from one host thread -- the shape the reference's host needs
(one process, one Rc<RefCell<Scene>>: reference src/main.rs:46, src/renderer.rs:57-63).  RCCL refuses a device listed twice, so on
one GPU the product can only ever run it with n = 1.  libmipt_multitest.so is the product's objects with mipt_multi.cpp compiled
against an RCCL TEST DOUBLE (tests/cpp/rccl_double/: N logical ranks on one device, ncclGather / ncclReduce done with stream-ordered
copies and a rank-ordered sum kernel).  Everything of render_impl that differs for rank i > 0 runs here: the per-device host
threads, the share computation and its remainder, the zero-share branch, the root-only receive offsets, stats aggregation, the
drain-on-failure path.  What the double does NOT cover is RCCL itself (transport, ring / tree reduction order)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mt(rrt):
    from rust_ray_tracing_amd import _lib as L
    return L.load_multitest()


@pytest.fixture(scope="module")
def atrium(rrt):
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.make_scene("atrium", n_target=20000, tex_size=32)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    return sc


class Multi:
    """n logical ranks, all on HIP device 0, through the multitest library's C ABI."""

    def __init__(self, mt, sc, n):
        from rust_ray_tracing_amd import _lib as L
        self.mt, self.sc, self.n, self.L = mt, sc, n, L
        self.h = C.c_void_p()
        d = sc.desc()
        L_ids = (C.c_int * n)(*([0] * n))
        rc = mt.mipt_multi_create(C.byref(d), L_ids, n, C.byref(self.h))
        assert rc == 0, mt.mipt_last_error()
        assert mt.mipt_multi_device_count(self.h) == n and mt.mipt_multi_root_device(self.h) == 0

    def render(self, rrt, w, h, spp, depth, mode, flags=0, want_rgba=True, **kw):
        L = self.L
        opt = rrt.make_options(w, h, spp, depth, flags=flags, **kw)
        hdr = np.zeros((h, w, 3), dtype=np.float32)
        rgba = np.zeros((h, w, 4), dtype=np.uint8) if want_rgba else None
        st = L.MiptMultiStats()
        rc = self.mt.mipt_render_multi(self.h, L.ptr(self.sc.camera.uniform), C.byref(opt), mode, L.ptr(hdr),
                                       L.ptr(rgba) if want_rgba else None, C.byref(st))
        return rc, hdr, rgba, st.as_dict()

    def device_stats(self):
        out = []
        for i in range(self.n):
            s = self.L.MiptStats()
            assert self.mt.mipt_multi_device_stats(self.h, i, C.byref(s)) == 0
            out.append(s.as_dict())
        return out

    def close(self):
        if self.h:
            self.mt.mipt_multi_destroy(self.h)
            self.h = None


COUNTERS = ("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "pixels")


@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("w,h", [(128, 72), (61, 37)])
def test_tiles_over_n_ranks_is_the_single_gpu_frame(rrt, orc, mt, atrium, n, w, h):
    from rust_ray_tracing_amd import _lib as L
    spp, depth = 4, 12
    r = rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h), output_image_path="/dev/null"))
    one_hdr, one_rgba, one_st = r.render_buffers(atrium, flags=L.FLAG_COUNT)           # the PRODUCT library, one device
    m = Multi(mt, atrium, n)
    try:
        rc, hdr, rgba, st = m.render(rrt, w, h, spp, depth, L.MULTI_TILES, flags=L.FLAG_COUNT)
        assert rc == 0, mt.mipt_last_error()
        assert np.array_equal(hdr.view(np.uint32), one_hdr.view(np.uint32)) and np.array_equal(rgba, one_rgba)
        ref, ref_rgba, _ = orc.render(atrium.tris, atrium.bvh_nodes, atrium.materials_array(), atrium.textures, atrium.camera.uniform, w, h, spp, depth)
        assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgba, ref_rgba)
        assert st["n_devices"] == n and len(st["device_kernel_ms"]) == n and all(ms > 0.0 for ms in st["device_kernel_ms"])
        per = m.device_stats()
        for k in COUNTERS:                                                             # MiptMultiStats.total = sum of the devices' launches
            assert sum(p[k] for p in per) == st[k] == one_st[k], k
        assert st["kernel_ms"] == max(p["kernel_ms"] for p in per)
        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        for i, p in enumerate(per):                                                    # rank i owns the tiles t with t % n == i
            own = [t for t in range(i, tiles, n)]
            tx = (w + 7) // 8
            px = sum(min(8, w - (t % tx) * 8) * min(8, h - (t // tx) * 8) for t in own)
            assert p["pixels"] == px, (i, p["pixels"], px)
        # a second frame through the same handle (buffers, streams and communicators reused), without the counters
        rc, hdr2, _, _ = m.render(rrt, w, h, spp, depth, L.MULTI_TILES)
        assert rc == 0 and np.array_equal(hdr2.view(np.uint32), hdr.view(np.uint32))
    finally:
        m.close()


def test_more_ranks_than_tiles(rrt, orc, mt, atrium):
    """8 x 8 pixels = ONE tile over three ranks: ranks 1 and 2 own nothing and contribute empty slices."""
    from rust_ray_tracing_amd import _lib as L
    m = Multi(mt, atrium, 3)
    try:
        rc, hdr, rgba, st = m.render(rrt, 8, 8, 2, 6, L.MULTI_TILES, flags=L.FLAG_COUNT)
        assert rc == 0, mt.mipt_last_error()
        ref, ref_rgba, ost = orc.render(atrium.tris, atrium.bvh_nodes, atrium.materials_array(), atrium.textures, atrium.camera.uniform, 8, 8, 2, 6)
        assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgba, ref_rgba)
        per = m.device_stats()
        assert [p["pixels"] for p in per] == [64, 0, 0] and st["rays"] == ost["rays"] == per[0]["rays"]
    finally:
        m.close()


@pytest.mark.parametrize("n", [3, 8])
def test_samples_over_n_ranks_equals_the_rank_ordered_oracle_sum(rrt, orc, mt, atrium, n):
    """spp = 7: n = 3 -> shares 3, 2, 2 (the remainder goes to the first ranks); n = 8 -> seven ranks with one sample and one
    zero-share rank that contributes zeros.  The double sums in rank order, so the comparison is bit for bit."""
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import sharding
    w, h, spp, depth = 64, 36, 7, 8
    m = Multi(mt, atrium, n)
    try:
        rc, hdr, rgba, st = m.render(rrt, w, h, spp, depth, L.MULTI_SAMPLES, flags=L.FLAG_COUNT)
        assert rc == 0, mt.mipt_last_error()
        mats = atrium.materials_array()
        total, rays, shares = None, 0, sharding.sample_ranges(spp, n)
        assert [c for _, c in shares] == ([3, 2, 2] if n == 3 else [1] * 7 + [0])
        per = m.device_stats()
        for i, (s0, cnt) in enumerate(shares):
            if cnt == 0:
                part = np.zeros((h, w, 3), dtype=np.float32)
                assert per[i]["pixels"] == 0 and per[i]["rays"] == 0
            else:
                part, _, ost = orc.render(atrium.tris, atrium.bvh_nodes, mats, atrium.textures, atrium.camera.uniform, w, h, cnt, depth,
                                          seed_mode=1, sample_begin=s0, sum_only=1, want_rgba8=False)
                rays += ost["rays"]
                assert per[i]["rays"] == ost["rays"] and per[i]["pixels"] == w * h, i
            total = part if total is None else total + part                         # ((p0 + p1) + p2) + ...
        want = total / np.float32(spp)                                               # cpu.rs:60, once, on the root
        assert st["rays"] == rays
        assert np.array_equal(hdr.view(np.uint32), want.view(np.uint32))
        assert rgba.shape == (h, w, 4) and np.all(rgba[..., 3] == 255)
        # and the single-GPU per-sample-seed render of the PRODUCT agrees within the sum-order tolerance
        one, _, _ = rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h), output_image_path="/dev/null",
                                                         seed_mode=L.SEED_PER_SAMPLE)).render_buffers(atrium)
        assert np.allclose(hdr, one, rtol=1e-6, atol=1e-6)
    finally:
        m.close()


def test_device_entry_with_n_ranks_and_pointer_validation(rrt, mt, atrium):
    """mipt_render_multi_device over 3 ranks writes the caller's device buffers; a host pointer is refused, not dereferenced."""
    import torch
    from rust_ray_tracing_amd import _lib as L
    w, h, spp, depth = 128, 72, 4, 12
    r = rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h), output_image_path="/dev/null"))
    ref_hdr, ref_rgba, _ = r.render_buffers(atrium)
    m = Multi(mt, atrium, 3)
    try:
        d_hdr = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
        d_rgba = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        opt = rrt.make_options(w, h, spp, depth)
        cam = L.ptr(atrium.camera.uniform)
        assert mt.mipt_render_multi_device(m.h, cam, C.byref(opt), L.MULTI_TILES, C.c_void_p(d_hdr.data_ptr()), C.c_void_p(d_rgba.data_ptr()), None) == 0, mt.mipt_last_error()
        assert np.array_equal(d_hdr.cpu().numpy().view(np.uint32), ref_hdr.reshape(-1).view(np.uint32))
        assert np.array_equal(d_rgba.cpu().numpy(), ref_rgba.reshape(-1))
        host = np.zeros(w * h * 3, dtype=np.float32)
        assert mt.mipt_render_multi_device(m.h, cam, C.byref(opt), L.MULTI_TILES, L.ptr(host), None, None) == L.ERR_INVALID_ARG
        assert b"not device memory of the root device" in mt.mipt_last_error()
        host8 = np.zeros(w * h * 4, dtype=np.uint8)
        assert mt.mipt_render_multi_device(m.h, cam, C.byref(opt), L.MULTI_TILES, C.c_void_p(d_hdr.data_ptr()), L.ptr(host8), None) == L.ERR_INVALID_ARG
    finally:
        m.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_injected_rank_failure_returns_a_status_and_the_next_call_succeeds(rrt, mt, atrium, mode):
    """Rank 1's collective call fails (kind 1) / rank 1's communicator reports an asynchronous error (kind 2): the call returns
    MIPT_ERR_RCCL with a message, every stream is drained (the handle's buffers are reused right away) and the next frame is right."""
    import torch
    from rust_ray_tracing_amd import _lib as L
    w, h, spp, depth = 64, 36, 3, 8
    m = Multi(mt, atrium, 3)
    try:
        rc, good, _, _ = m.render(rrt, w, h, spp, depth, mode)
        assert rc == 0, mt.mipt_last_error()
        done0 = mt.rccl_double_inject(-1, 0)
        for kind in (1, 2):
            mt.rccl_double_inject(1, kind)
            rc, _, _, _ = m.render(rrt, w, h, spp, depth, mode)
            assert rc == L.ERR_RCCL, (kind, rc)
            msg = mt.mipt_last_error()
            assert (b"RCCL collective failed" in msg) if kind == 1 else (b"asynchronous error" in msg), msg
            torch.cuda.synchronize()                                                 # nothing left in flight on the device
            rc, again, _, _ = m.render(rrt, w, h, spp, depth, mode)
            assert rc == 0, mt.mipt_last_error()
            assert np.array_equal(again.view(np.uint32), good.view(np.uint32)), kind
        # kind 1 moved no data for its frame; every other frame completed exactly one collective
        assert mt.rccl_double_inject(-1, 0) - done0 == 3
    finally:
        mt.rccl_double_inject(1, 0)
        m.close()


def test_argument_errors_with_n_ranks(rrt, mt, atrium):
    from rust_ray_tracing_amd import _lib as L
    m = Multi(mt, atrium, 2)
    try:
        for kw in (dict(tile_rank=1, tile_world=2), dict(flags=L.FLAG_SUM), dict(sample_begin=5)):
            rc, _, _, _ = m.render(rrt, 32, 32, 1, 2, L.MULTI_TILES, **kw)
            assert rc == L.ERR_INVALID_ARG and b"owns the sharding" in mt.mipt_last_error(), kw
        assert mt.mipt_multi_device_stats(m.h, 2, C.byref(L.MiptStats())) == L.ERR_INVALID_ARG
    finally:
        m.close()
    h = C.c_void_p()
    d = atrium.desc()
    assert mt.mipt_multi_create(C.byref(d), (C.c_int * 2)(0, 9), 2, C.byref(h)) == L.ERR_INVALID_ARG        # no such device
    assert mt.mipt_multi_create(C.byref(d), None, 65, C.byref(h)) == L.ERR_INVALID_ARG

"""-m gpu: random camera poses (the round-1 soaks were builder-only scripts under tests/tools/).  For several
poses inside and around three scene families the GPU frame must (i) be identical in both traversal modes -- the culling identity
with margin 2^-7 is empirical, so it is exercised beyond the bench view -- and (ii) equal the CPU oracle
(un-culled traversal, reference cpu/ray.rs:84-139) on a strided pixel sample, bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,kw,n_views,box,seed", [
    ("atrium", dict(n_target=300000, tex_size=128), 9, 14.0, 11),
    ("dragon", dict(n_target=200000), 6, 6.0, 12),
    ("helmet", dict(n_target=15000, tex_size=64), 5, 4.0, 13),
    ("cornell", {}, 4, 0.8, 14),
])
def test_random_views_culled_equals_reference_equals_oracle(rrt, orc, kind, kw, n_views, box, seed):
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import synth
    lib = rrt.load()
    rng = np.random.default_rng(seed)
    w, h, spp, depth, stride = 640, 360, 4, 64, 37
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    m = sc.materials_array()
    bufs = [np.zeros(w * h * 3, dtype=np.float32) for _ in range(2)]
    sel = np.arange(0, w * h, stride)
    checked = 0
    for v in range(n_views):
        if v == 0:
            pos, pitch, yaw = cam
        else:
            pos = tuple(float(x) for x in (np.array(cam[0]) + rng.uniform(-box, box, 3) * np.array([1.0, 0.25, 1.0])))
            pitch, yaw = float(rng.uniform(-60, 60)), float(rng.uniform(-180, 180))
        sc.set_camera(rrt.Camera(position=pos, pitch=pitch, yaw=yaw))
        hnd = sc.upload(0)
        for (trav, margin), buf in zip(((L.TRAVERSAL_REFERENCE, 0.0), (L.TRAVERSAL_CULLED, L.CULL_MARGIN_SAFE)), bufs):
            o = rrt.make_options(w, h, spp, depth, traversal=trav, cull_margin=margin)
            L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, None), "mipt_render")
        a, b = bufs[0].reshape(-1, 3), bufs[1].reshape(-1, 3)
        differ = ((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).any(1)
        assert not differ.any(), (kind, v, pos, pitch, yaw, int(differ.sum()))
        ref, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, spp, depth, cull=0, pix_stride=stride,
                               want_rgba8=False)
        r = ref.reshape(-1, 3)[sel]
        bad = ((b[sel].view(np.uint32) != r.view(np.uint32)) & ~(np.isnan(b[sel]) & np.isnan(r))).any(1)
        assert not bad.any(), (kind, v, pos, pitch, yaw, int(bad.sum()))
        checked += len(sel)
    assert checked == n_views * len(sel)

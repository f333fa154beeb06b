"""-m gpu: goldens, edge cases, device-arithmetic known answers, shards and full-size properties of the HIP path
(all through the C ABI)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _scene(rrt, kind, **kw):
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    return sc


def _render(rrt, sc, w, h, spp, depth, **kw):
    flags = kw.pop("flags", rrt.FLAG_COUNT)
    r = rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h),
                                             output_image_path="/dev/null", **kw))
    return r.render_buffers(sc, flags=flags)


def _raw(rrt, sc, opt, n_floats):
    """mipt_render with explicit MiptOptions into a host buffer of n_floats."""
    from rust_ray_tracing_amd import _lib as L
    out = np.zeros(n_floats, dtype=np.float32)
    st = L.MiptStats()
    L.check(rrt.load().mipt_render(sc.upload(0), L.ptr(sc.camera.uniform), C.byref(opt), L.ptr(out), None, C.byref(st)), "mipt_render")
    return out, st.as_dict()


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cornell_64x64_4spp", "helmet4k_96x54_4spp", "atrium20k_64x36_2spp",
                                  "atrium20k_64x36_4spp_persample", "cornell_256x256_4spp_rgba"])
def test_against_committed_goldens(rrt, name):
    import make_golden
    kind, kw, w, h, spp, depth, seed_mode = make_golden.CASES[name]
    sc = _scene(rrt, kind, **kw)
    hdr, rgba, st = _render(rrt, sc, w, h, spp, depth, seed_mode=seed_mode)
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert st["rays"] == int(g["rays"])
    assert np.array_equal(rgba, g["rgba"])
    if "hdr" in g:
        assert np.array_equal(hdr.view(np.uint32), g["hdr"].view(np.uint32))
        assert st["inner_steps"] == int(g["inner_steps"]) and st["tri_tests"] == int(g["tri_tests"])


def test_config1_obj_to_pixels(rrt, orc, tmp_path):
    """BASELINE.json configs[0]: 12-triangle OBJ, 256x256, 4 spp -- OBJ -> Scene -> BVH -> render, vs the oracle."""
    from rust_ray_tracing_amd import synth
    sc = rrt.Scene.load(synth.write_cornell_obj(str(tmp_path)))
    sc.set_camera(rrt.Camera(position=synth.CORNELL_CAMERA[0], pitch=0.0, yaw=0.0))
    hdr, rgba, st = _render(rrt, sc, 256, 256, 4, 64)
    ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), [], sc.camera.uniform, 256, 256, 4, 64)
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgba, ref_rgba)
    assert rgba[0, 0].tolist() == [254, 254, 254, 255]          # white sky quantises to 254 in f32 (test_oracle_kat)
    raw = rrt.Renderer.new(rrt.RendererOptions(samples=4, max_ray_depth=64, output_image_dimensions=(256, 256),
                                               output_image_path=str(tmp_path / "out.png"))).render(sc)
    assert raw == ref_rgba.tobytes() and len(raw) == 256 * 256 * 4   # the Vec<u8> cpu.rs:63-67 returns
    assert (tmp_path / "out.png").stat().st_size > 100


# ---- edge cases the reference's structure implies -----------------------------------------------------
def test_root_leaf_single_triangle_and_one_pixel(rrt, orc):
    from rust_ray_tracing_amd import TRIANGLE
    t = np.zeros(1, dtype=TRIANGLE)
    t["vertices"]["position"][0] = [(-5, -5, -5), (-5, 5, 5), (-5, 5, -5)]
    t["vertices"]["normal"][0] = [(1, 0, 0)] * 3
    sc = rrt.Scene.from_arrays(t, [rrt.material_default()])
    assert len(sc.bvh_nodes) == 1
    sc.set_camera(rrt.Camera(position=(3, 0, 0), pitch=0.0, yaw=0.0))
    for (w, h, spp) in [(1, 1, 1), (5, 3, 7), (64, 64, 2)]:
        hdr, rgba, st = _render(rrt, sc, w, h, spp, 4)
        ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), [], sc.camera.uniform, w, h, spp, 4)
        assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgba, ref_rgba)
        assert st["rays"] == rst["rays"] and st["pixels"] == w * h


def test_big_leaves_and_deep_stack(rrt, orc):
    """>= 64 triangles in one leaf (child-ref stack entries) and a stack deeper than the 16 LDS entries (HBM spill)."""
    from rust_ray_tracing_amd import TRIANGLE
    n = 150                                             # identical AABBs -> identical centroids -> unsplittable leaf
    big = np.zeros(n, dtype=TRIANGLE)
    tt = np.linspace(-0.95, 0.95, n).astype(np.float32)
    for i in range(n):                                  # coplanar, overlapping: equal-t hits exercise the strict-< order (T6)
        big["vertices"]["position"][i] = [(-4.0, -1.0, -1.0), (-4.0, 1.0, 1.0), (-4.0, tt[i], -tt[i])]
    big["vertices"]["normal"] = (1, 0, 0)
    # a long chain of ever larger nested shells in front forces many pending far children
    from rust_ray_tracing_amd import synth
    shells = []
    for k in range(40):
        r = 0.2 * 1.25 ** k
        q = synth.quad((-2 - 0.02 * k, -r, -r), (-2 - 0.02 * k, r, -r), (-2 - 0.02 * k, r, r), (-2 - 0.02 * k, -r, r), (1, 0, 0), 0)
        shells.append(q)
    tris = np.concatenate([big] + shells)
    sc = rrt.Scene.from_arrays(tris, [rrt.material_default()])
    assert sc.bvh_nodes["num_tris"].max() >= 64
    sc.set_camera(rrt.Camera(position=(3, 0.01, 0.02), pitch=0.0, yaw=0.0))
    hdr, rgba, st = _render(rrt, sc, 96, 96, 2, 6)
    ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), [], sc.camera.uniform, 96, 96, 2, 6)
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))
    assert st["max_stack"] == rst["max_stack"] and st["tri_tests"] == rst["tri_tests"]
    # and a genuinely deep traversal stack: the 1M-triangle atrium needs > 16 entries on some rays
    sc = _scene(rrt, "atrium", n_target=1_000_000, tex_size=64)
    hdr, _, st = _render(rrt, sc, 256, 144, 2, 32)
    ref, _, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, 256, 144, 2, 32, want_rgba8=False)
    assert st["max_stack"] == rst["max_stack"]
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))


def test_stack_overflow_is_reported_not_hidden(rrt):
    """The reference panics when its 32-entry stack overflows (ray.rs:85); the kernel has 64 entries and
    returns MIPT_ERR_STACK beyond that.  A normal scene must report zero overflows."""
    sc = _scene(rrt, "dragon", n_target=30000)
    _, _, st = _render(rrt, sc, 64, 36, 1, 8)
    assert st["stack_overflows"] == 0 and st["tex_clamped"] == 0


def test_option_validation_on_device(rrt):
    from rust_ray_tracing_amd import _lib as L
    sc = _scene(rrt, "cornell")
    lib = rrt.load()
    h = sc.upload(0)
    buf = np.zeros(64 * 64 * 3, dtype=np.float32)
    for bad in (dict(width=0), dict(samples=0), dict(max_ray_depth=0), dict(seed_mode=7), dict(traversal=9), dict(tile_rank=2, tile_world=2)):
        o = rrt.make_options(64, 64, 1, 4)
        for k, v in bad.items():
            setattr(o, k, v)
        assert lib.mipt_render(h, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, None) == L.ERR_INVALID_ARG, bad
    o = rrt.make_options(46341, 46341, 1, 1)            # w*h >= 2^31 - 87636354: 32-bit pixel seed would wrap
    assert lib.mipt_render(h, L.ptr(sc.camera.uniform), C.byref(o), None, None, None) == L.ERR_INVALID_ARG


# ---- the kernel's arithmetic building blocks, bit for bit ----------------------------------------------
def test_device_arithmetic_matches_oracle(rrt, orc):
    lib, O = rrt.load(), orc.load()

    def dev(op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.zeros_like(a)
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
        assert rrt.load_diag().mipt_debug_eval(op, a.ctypes.data, None if bb is None else bb.ctypes.data, a.size, out.ctypes.data) == 0
        return out

    def same(x, y):
        return np.all((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y)))
    rng = np.random.default_rng(1)
    n = 400_000
    a = (rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))).astype(np.float32)
    b = (rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))).astype(np.float32)
    a[:8] = [0, -0.0, np.inf, -np.inf, np.nan, 1, 1e-45, 3e38]
    b[:8] = [0, 1, np.inf, 0, 1, 0, 3, 1e-45]
    with np.errstate(all="ignore"):
        assert same(dev(3, a, b), a / b)                       # IEEE division (slab test, normalize)
        assert same(dev(4, np.abs(a)), np.sqrt(np.abs(a)))     # correctly rounded sqrt (vec3.rs:95)
        assert same(dev(5, a, b), a * b) and same(dev(6, a, b), a + b)   # denormals preserved, no FMA
    x = np.concatenate([(rng.random(50000) * 6.2832).astype(np.float32), np.float32([0, 6.283185, 3.1415927, 1.5707964, np.inf, np.nan])])
    assert same(dev(0, x), np.array([O.orc_glibc_cosf(float(v)) for v in x], dtype=np.float32))
    r = np.concatenate([rng.random(50000).astype(np.float32), np.float32([0, 1, 1e-45, 2.3e-10, np.inf, -1])])
    assert same(dev(1, r), np.array([O.orc_glibc_log10f(float(v)) for v in r], dtype=np.float32))
    y = np.full(len(r), np.float32(1) / np.float32(2.4), dtype=np.float32)
    assert same(dev(2, r, y), np.array([O.orc_glibc_powf(float(v), float(y[0])) for v in r], dtype=np.float32))
    seeds = rng.integers(1, 2**32, 30000, dtype=np.uint32)
    want = np.zeros((len(seeds), 3), dtype=np.float32)
    for i, sd in enumerate(seeds):
        st = C.c_uint32(int(sd))
        o3 = (C.c_float * 3)()
        O.orc_rand_in_unit_sphere(C.byref(st), 0, C.byref(o3))
        want[i] = list(o3)
    for comp in range(3):
        assert same(dev(11, seeds.view(np.float32), np.full(len(seeds), comp, np.float32)), want[:, comp])
    # sRGB + quantise epilogue
    v = np.concatenate([rng.random(20000).astype(np.float32) * 1.2, np.float32([0, 1, 0.0031308, 7.5, np.nan, -0.5])])
    got = dev(12, v).view(np.uint32)
    for i in range(len(v)):
        s3 = (C.c_float * 3)()
        q3 = (C.c_uint8 * 3)()
        O.orc_linear_to_srgb(C.byref((C.c_float * 3)(v[i], v[i], v[i])), 0, C.byref(s3))
        O.orc_quantize(C.byref(s3), C.byref(q3))
        assert got[i] == q3[0], (v[i], got[i], q3[0])


# ---- shards (multi-GPU path on one GPU: loop the virtual ranks) --------------------------------------
@pytest.mark.parametrize("w,h,world", [(64, 40, 2), (61, 37, 3), (128, 72, 8)])
def test_tile_shards_reassemble_bit_exact(rrt, w, h, world):
    import torch
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import sharding
    lib = rrt.load()
    sc = _scene(rrt, "atrium", n_target=20000, tex_size=32)
    full, _ = _raw(rrt, sc, rrt.make_options(w, h, 2, 8), w * h * 3)
    slots = int(lib.mipt_packed_pixels(w, h, world))
    assert slots == sharding.packed_pixels(w, h, world)
    packed_all = np.zeros((world, slots, 3), dtype=np.float32)
    total_px = 0
    for r in range(world):
        out, st = _raw(rrt, sc, rrt.make_options(w, h, 2, 8, flags=L.FLAG_PACKED, tile_rank=r, tile_world=world), slots * 3)
        packed_all[r] = out.reshape(slots, 3)
        total_px += st["pixels"]
        pix = sharding.slot_pixels(w, h, r, world)                 # kernel and Python agree on the slot layout
        assert np.array_equal(out.reshape(slots, 3)[pix >= 0].view(np.uint32), full.reshape(-1, 3)[pix[pix >= 0]].view(np.uint32))
    assert total_px == w * h
    # device-side de-interleave of the all-gathered slices (what bench.py runs after the RCCL all-gather)
    d_all = torch.from_numpy(packed_all.reshape(-1)).cuda()
    d_frame = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    L.check(lib.mipt_unpack_tiles(C.c_void_p(d_all.data_ptr()), w, h, world, C.c_void_p(d_frame.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mipt_unpack_tiles")
    torch.cuda.synchronize()
    assert np.array_equal(d_frame.cpu().numpy().view(np.uint32), full.view(np.uint32))
    assert np.array_equal(sharding.unpack(packed_all, w, h, world).reshape(-1).view(np.uint32), full.view(np.uint32))
    # full-frame (non-packed) shard output writes only the rank's pixels
    part, st = _raw(rrt, sc, rrt.make_options(w, h, 2, 8, tile_rank=1, tile_world=world), w * h * 3)
    own = sharding.slot_pixels(w, h, 1, world)
    own = own[own >= 0]
    assert np.array_equal(part.reshape(-1, 3)[own].view(np.uint32), full.reshape(-1, 3)[own].view(np.uint32))
    mask = np.ones(w * h, bool)
    mask[own] = False
    assert np.all(part.reshape(-1, 3)[mask] == 0)


def test_sample_shards_per_sample_seeds(rrt, orc):
    """Config 5: disjoint sample ranges with the per-sample seed (rt_compute.wgsl:102), un-normalised sums."""
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import sharding
    sc = _scene(rrt, "atrium", n_target=20000, tex_size=32)
    w, h, spp, depth, world = 64, 36, 7, 8, 3
    m = sc.materials_array()
    total = np.zeros(w * h * 3, dtype=np.float32)
    for (s0, n) in sharding.sample_ranges(spp, world):
        part, st = _raw(rrt, sc, rrt.make_options(w, h, n, depth, seed_mode=L.SEED_PER_SAMPLE, flags=L.FLAG_SUM, sample_begin=s0), w * h * 3)
        ref, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, n, depth, seed_mode=1,
                               sample_begin=s0, sum_only=1, want_rgba8=False)
        assert np.array_equal(part.view(np.uint32), ref.reshape(-1).view(np.uint32))   # per-rank partial: bit-exact
        total += part
    whole, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, spp, depth, seed_mode=1, sum_only=1, want_rgba8=False)
    assert np.allclose(total, whole.reshape(-1), rtol=1e-6, atol=1e-6)                   # rank-tree vs sequential f32 sum


def test_tonemap_device_matches_render_path(rrt):
    import torch
    from rust_ray_tracing_amd import _lib as L
    sc = _scene(rrt, "helmet", n_target=4000, tex_size=64)
    hdr, rgba, _ = _render(rrt, sc, 96, 54, 4, 12)
    d = torch.from_numpy(hdr.reshape(-1) * np.float32(4)).cuda()
    out = torch.zeros(96 * 54 * 4, dtype=torch.uint8, device="cuda")
    L.check(rrt.load().mipt_tonemap_device(C.c_void_p(d.data_ptr()), 96 * 54, 4.0, C.c_void_p(out.data_ptr()), None), "tonemap")
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(54, 96, 4)
    assert np.mean(got != rgba) < 0.01            # (x*4)/4 == x except at f32 overflow/denormal edges
    assert np.array_equal(got[..., 3], rgba[..., 3])


# ---- BASELINE.json's full sizes: size-independent properties --------------------------------------------
def test_full_size_properties_config_M(rrt, orc):
    """1920x1080, 8 spp, depth 64 on a 2M-triangle atrium: (i) culled traversal with the safe margin is bit-identical
    to the reference traversal over the WHOLE frame; (ii) a strided sample equals the oracle bit for bit;
    (iii) rendering twice is idempotent; (iv) checksums of the tile shards add up to the frame's."""
    from rust_ray_tracing_amd import _lib as L
    sc = _scene(rrt, "atrium", n_target=2_000_000, tex_size=256)
    w, h, spp, depth = 1920, 1080, 8, 64
    ref_t, st0 = _raw(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_REFERENCE, flags=L.FLAG_COUNT), w * h * 3)
    cul, st1 = _raw(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED, flags=L.FLAG_COUNT), w * h * 3)
    assert st0["rays"] == st1["rays"] and st0["hits"] == st1["hits"] and st1["inner_steps"] < st0["inner_steps"]
    assert np.array_equal(ref_t.view(np.uint32), cul.view(np.uint32))
    again, _ = _raw(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED), w * h * 3)
    assert np.array_equal(again.view(np.uint32), cul.view(np.uint32))
    stride = 997
    o, _, ost = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                           pix_stride=stride, want_rgba8=False)
    idx = np.arange(0, w * h, stride)
    assert np.array_equal(o.reshape(-1, 3)[idx].view(np.uint32), cul.reshape(-1, 3)[idx].view(np.uint32))
    csum = int(cul.view(np.uint32).astype(np.uint64).sum())
    parts = 0
    for r in range(4):
        p, st = _raw(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED, flags=L.FLAG_PACKED, tile_rank=r, tile_world=4),
                     int(rrt.load().mipt_packed_pixels(w, h, 4)) * 3)
        parts += int(p.view(np.uint32).astype(np.uint64).sum())
    assert parts == csum                                         # 1080p has no ragged tiles: padding slots stay 0
    assert float(np.isfinite(cul).mean()) == 1.0 and 0.01 < float(cul.mean()) < 5.0


def test_fast_division_is_ieee(rrt):
    """The slab test's per-ray-reciprocal division (pt_kernel.hip fdiv_ray) must equal IEEE a/d bit for bit on its
    whole guarded range: |d| in [2^-60, 2], a = 0 or |a| in [2^-99, 2^41] (ray_safe + the scene check of mipt_scene_create).
    Includes quotients engineered to sit on rounding boundaries (a = q*d +- a few ulps), exactly representable quotients,
    the hard divisors of reciprocal iterations, the 2^-99 edge, and a == 0 (a zero of either sign: the slab test cannot tell)."""
    def dev(op, a, b):
        out = np.zeros_like(a)
        assert rrt.load_diag().mipt_debug_eval(op, a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data) == 0
        return out
    rng = np.random.default_rng(11)
    n = 4_000_000
    sign = lambda k: rng.choice(np.float32([-1, 1]), k)
    d = (sign(n) * np.exp2(rng.uniform(-60, 1, n))).astype(np.float32)
    q = (sign(n) * np.exp2(rng.uniform(-99, 100, n))).astype(np.float32)
    # 1. random a over the range  2. a = RN(q*d) (quotient near a representable value)  3. +- 1..2 ulps of that
    with np.errstate(all="ignore"):
        a_rand = (sign(n) * np.exp2(rng.uniform(-99, 41, n))).astype(np.float32)
        a_exact = (q.astype(np.float64) * d).astype(np.float32)
        a_near = np.nextafter(a_exact, np.float32(np.inf) * sign(n)).astype(np.float32)
        a_edge = (sign(n) * np.exp2(rng.uniform(-99, -90, n))).astype(np.float32)          # residuals down in the denormals
        a_edge[:1000] = np.float32(2.0 ** -99) * sign(1000)
        # mantissas of all ones / powers of two in the divisor (the classic hard cases for reciprocal iterations)
        d_hard = d.copy()
        d_hard[::2] = (d_hard[::2].view(np.uint32) | np.uint32(0x007FFFFF)).view(np.float32)
        d_hard[1::2] = (d_hard[1::2].view(np.uint32) & np.uint32(0xFF800000)).view(np.float32)
        d_big = (sign(n) * np.exp2(rng.uniform(0, 1, n))).astype(np.float32)                # smallest quotients
        for a, dd in ((a_rand, d), (a_exact, d), (a_near, d), (a_rand, d_hard), (a_near, d_hard), (a_edge, d_big), (a_edge, d_hard)):
            want = (a / dd).astype(np.float32)
            ok = (np.abs(a) >= 2.0 ** -99) & (np.abs(a) <= 2.0 ** 41) & (np.abs(dd) >= 2.0 ** -60) & (np.abs(dd) <= 2)
            got = dev(14, a, dd)
            assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
            assert ok.mean() > 0.6
        zero = np.zeros(n, dtype=np.float32) * sign(n)                                       # +0 and -0
        for dd in (d, d_hard):
            got = dev(14, zero, dd)
            assert not got.any() and not np.isnan(got).any()


def test_exact_division_guard_cases(rrt, orc):
    """The guard of the fast quotient is evaluated per ray and per scene, not per step (pt_kernel.hip ray_safe).  Its corner cases,
    each against the oracle's IEEE divisions, bit for bit in both traversal modes: (i) lattice geometry with the camera ON lattice
    planes (a == 0 in the fast path: the quotient is a zero of possibly the other sign); (ii) the same with the camera at a tiny
    non-zero offset (0 < |o| < 2^-70: those rays must take the IEEE divisions); (iii) a scene with tiny non-zero plane coordinates
    on two axes (DevScene::tiny_axes: rays starting at exactly 0 on those axes take them, a = p would be tiny), seen from a camera
    at 0 on those axes and from one that is not."""
    from rust_ray_tracing_amd import TRIANGLE
    from rust_ray_tracing_amd.synth import material
    rng = np.random.default_rng(77)
    n = 300
    c = np.round(rng.standard_normal((n, 1, 3)) * 3)
    p = c + np.round(rng.standard_normal((n, 3, 3)) * 1.5)
    p[: n // 3, :, 0] = 0.0                                                                  # a third of the triangles lie in planes through 0
    p[n // 3: 2 * n // 3, :, 1] = 0.0
    t = np.zeros(n, dtype=TRIANGLE)
    nrm = rng.standard_normal((n, 3, 3))
    t["vertices"]["normal"] = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
    mats = [material(base=(0.7, 0.6, 0.5), emission=(0.0, 0.0, 0.0)), material(base=(0.9, 0.9, 0.9), emission=(2.0, 1.5, 1.0))]
    t["material_id"] = rng.integers(0, 2, n)
    w, h, spp, depth = 64, 36, 3, 12
    cases = [(p, (0.0, 0.0, -7.0)), (p, (1.0, -2.0, -7.0)), (p, (1e-30, 0.0, -7.0)), (p, (0.0, -1e-38, -7.0)),
             (p + np.array([1e-30, 0.0, 2e-35]), (0.0, 0.0, -7.0)), (p + np.array([1e-30, 0.0, 2e-35]), (0.5, 0.0, -7.0)),
             (p + np.array([0.0, 1.2e-23, 0.0]), (0.0, 1.3234890e-23, -7.0))]
    for k, (pos_arr, cam) in enumerate(cases):
        t["vertices"]["position"] = pos_arr.astype(np.float32)
        sc = rrt.Scene.from_arrays(t, mats, [])
        sc.set_camera(rrt.Camera(position=cam, pitch=0.0, yaw=0.0))
        for trav, margin in ((0, 0.0), (1, 0.0078125)):
            hdr, rgba, st = _render(rrt, sc, w, h, spp, depth, traversal=trav, cull_margin=margin)
            ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                                            cull=trav, cull_margin=margin)
            same = (hdr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(hdr) & np.isnan(ref))
            assert same.all(), (k, trav, int((~same).sum()))
            for key in ("rays", "inner_steps", "tri_tests", "hits"):
                assert st[key] == rst[key], (key, k, trav)
        assert st["hits"] > 1000


def test_u8_over_255_is_ieee(rrt):
    """Texel unpack (vec3.rs:252-260): the kernel's 5-instruction quotient must equal (k as f32) / 255.0 for all 256 bytes."""
    k = np.arange(256, dtype=np.uint32)
    out = np.zeros(256, dtype=np.float32)
    assert rrt.load_diag().mipt_debug_eval(15, k.view(np.float32).ctypes.data, None, 256, out.ctypes.data) == 0
    want = k.astype(np.float32) / np.float32(255.0)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


# ---- BASELINE.json configs[1], [2] and [4] at their named sizes (parity cases, not bench lines) ----------------
@pytest.mark.parametrize("name,kind,kw,w,h,spp,depth,stride", [
    ("config2 helmet-class 15k tris, 1920x1080, 16 spp", "helmet", dict(n_target=15000, tex_size=512), 1920, 1080, 16, 64, 1013),
    ("config3 dragon-class 870k tris, 1920x1080, 64 spp", "dragon", dict(n_target=870000), 1920, 1080, 64, 64, 4999),
])
def test_named_configs_full_size(rrt, orc, name, kind, kw, w, h, spp, depth, stride):
    """Whole frame on the GPU in both traversal modes (must be bit-identical), a strided sample against the oracle."""
    from rust_ray_tracing_amd import _lib as L
    sc = _scene(rrt, kind, **kw)
    cul, st1 = _raw(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED, flags=L.FLAG_COUNT), w * h * 3)
    ref_t, st0 = _raw(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_REFERENCE), w * h * 3)
    assert np.array_equal(ref_t.view(np.uint32), cul.view(np.uint32))
    o, _, ost = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                           pix_stride=stride, want_rgba8=False)
    idx = np.arange(0, w * h, stride)
    assert np.array_equal(o.reshape(-1, 3)[idx].view(np.uint32), cul.reshape(-1, 3)[idx].view(np.uint32))
    print(f"{name}: {st1['rays']} rays, {st1['kernel_ms']:.1f} ms, {st1['rays'] / st1['kernel_ms'] / 1e3:.0f} Mray/s, "
          f"{st1['inner_steps'] / st1['rays']:.1f} inner + {st1['tri_tests'] / st1['rays']:.1f} tri per ray, max stack {st1['max_stack']}")


def test_config5_sample_sharded_4096(rrt, orc):
    """configs[4]: 4096x4096 with samples sharded across ranks (per-sample seeds, sum-reduce) -- at 16 spp over 8 virtual
    ranks on a 1M-triangle atrium; every rank's partial sum is checked on a strided sample against the oracle."""
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import sharding
    sc = _scene(rrt, "atrium", n_target=1_000_000, tex_size=256)
    w = h = 4096
    spp, depth, world, stride = 16, 16, 8, 100003
    idx = np.arange(0, w * h, stride)
    total = np.zeros(w * h * 3, dtype=np.float32)
    m = sc.materials_array()
    for r, (s0, n) in enumerate(sharding.sample_ranges(spp, world)):
        part, st = _raw(rrt, sc, rrt.make_options(w, h, n, depth, seed_mode=L.SEED_PER_SAMPLE, traversal=L.TRAVERSAL_CULLED,
                                                  flags=L.FLAG_SUM, sample_begin=s0), w * h * 3)
        assert st["pixels"] == w * h
        if r in (0, 5):
            o, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, n, depth, seed_mode=1,
                                 sample_begin=s0, sum_only=1, pix_stride=stride, want_rgba8=False)
            assert np.array_equal(o.reshape(-1, 3)[idx].view(np.uint32), part.reshape(-1, 3)[idx].view(np.uint32))
        total += part
    whole, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, spp, depth, seed_mode=1, sum_only=1,
                             pix_stride=stride, want_rgba8=False)
    assert np.allclose(total.reshape(-1, 3)[idx], whole.reshape(-1, 3)[idx], rtol=1e-5, atol=1e-5)


def test_progressive_accumulation_and_postprocess(rrt, orc, tmp_path):
    """resumable per-sample accumulation (gpu.rs:17-77 without the rgba16unorm quantisation), the
    sRGB-then-ACES post-process pass (pp_compute.wgsl:7-34) and RGBA16 output (renderer.rs:67-73)."""
    import torch
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import host
    lib = rrt.load()
    sc = _scene(rrt, "atrium", n_target=20000, tex_size=32)
    w, h, depth = 96, 54, 8
    hnd = sc.upload(0)
    acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    m = sc.materials_array()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    done = 0
    expect = np.zeros((h, w, 3), dtype=np.float32)
    for n in (1, 2, 3):                              # three calls = samples 1, 2..3, 4..6 accumulated in place
        o = rrt.make_options(w, h, n, depth, seed_mode=L.SEED_PER_SAMPLE, flags=L.FLAG_SUM | L.FLAG_ACCUM, sample_begin=done + 1)
        L.check(lib.mipt_render_device(hnd, L.ptr(sc.camera.uniform), C.byref(o), C.c_void_p(acc.data_ptr()), None, stream, None), "render")
        done += n
        # running sum = previous sum + (this call's samples summed in order): the same f32 summation tree on the oracle side
        part, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, n, depth, seed_mode=1,
                                sample_begin=done - n + 1, sum_only=1, want_rgba8=False)
        expect = expect + part
        assert np.array_equal(acc.cpu().numpy().view(np.uint32), expect.reshape(-1).view(np.uint32))
    ref = expect
    whole, _, _ = orc.render(sc.tris, sc.bvh_nodes, m, sc.textures, sc.camera.uniform, w, h, done, depth, seed_mode=1, sum_only=1, want_rgba8=False)
    assert np.allclose(ref, whole, rtol=1e-6, atol=1e-6)     # vs one sequential pass: summation order only
    out16 = torch.zeros(w * h * 4, dtype=torch.int16, device="cuda")
    L.check(lib.mipt_postprocess_device(C.c_void_p(acc.data_ptr()), w * h, float(done), C.c_void_p(out16.data_ptr()), stream), "postprocess")
    torch.cuda.synchronize()
    got = out16.cpu().numpy().view(np.uint16).reshape(h, w, 4)
    want = orc.postprocess(ref, divisor=float(done))
    assert np.array_equal(got, want)
    assert got[..., 3].min() == 65535 and 0 < got[..., :3].mean() < 65535
    host.write_png_rgba16(str(tmp_path / "pp.png"), got)
    sig = open(tmp_path / "pp.png", "rb").read(26)
    assert sig[:8] == b"\x89PNG\r\n\x1a\n" and sig[24] == 16 and sig[25] == 6          # bit depth 16, colour type RGBA
    # ACCUM without SUM, or through the host-buffer entry point, is rejected
    o = rrt.make_options(w, h, 1, depth, flags=L.FLAG_ACCUM)
    assert lib.mipt_render_device(hnd, L.ptr(sc.camera.uniform), C.byref(o), C.c_void_p(acc.data_ptr()), None, stream, None) == L.ERR_INVALID_ARG


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_random_scenes_match_oracle(rrt, orc, seed):
    """Random triangle soups with degenerate / axis-aligned / coincident geometry, random emissive + textured materials and
    random camera poses: the kernel must agree with the oracle bit for bit in both traversal modes.  Exercises the exact
    division's slow path (direction components that are exactly 0, origins exactly on bounding planes), zero-area triangles
    (det = 0 -> inf / NaN), coplanar overlapping triangles (strict-< tie-breaking, T6) and NaN-free clamped texels."""
    from rust_ray_tracing_amd import TRIANGLE
    from rust_ray_tracing_amd.synth import material
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 400))
    scale = float(rng.choice([0.01, 1.0, 50.0]))
    c = rng.standard_normal((n, 1, 3)) * scale * 3
    p = c + rng.standard_normal((n, 3, 3)) * scale * rng.random((n, 1, 1)) * 2
    if seed % 3 == 0:                                    # snap to a grid: axis-aligned, coplanar, coincident, zero-area
        p = np.round(p / scale) * scale
    if seed % 4 == 1:
        p[: n // 4, 2] = p[: n // 4, 1]                  # degenerate (two equal vertices)
    t = np.zeros(n, dtype=TRIANGLE)
    t["vertices"]["position"] = p.astype(np.float32)
    nrm = rng.standard_normal((n, 3, 3))
    t["vertices"]["normal"] = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(np.float32)
    t["vertices"]["tex_coord_x"] = (rng.random((n, 3)) * 7.9).astype(np.float32)
    t["vertices"]["tex_coord_y"] = (rng.random((n, 3)) * 7.9).astype(np.float32)
    n_mat = int(rng.integers(1, 6))
    texs = [rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4), dtype=np.uint8) for _ in range(2)]
    mats = []
    for i in range(n_mat):
        mats.append(material(base=tuple(rng.random(3)), emission=tuple(rng.random(3) * (i % 2) * 3),
                             base_tex=(i % 2) if i % 3 == 0 else 0xFFFFFFFF, emission_tex=1 if i % 4 == 3 else 0xFFFFFFFF))
    t["material_id"] = rng.integers(0, n_mat, n)
    sc = rrt.Scene.from_arrays(t, mats, texs)
    if seed % 2 == 0:                                    # camera on a lattice point looking along an axis: d components exactly 0 / a == 0
        pos = tuple(np.round(rng.standard_normal(3) * 4) * scale)
        pitch, yaw = 0.0, float(rng.choice([0.0, 90.0, 180.0, -90.0]))
    else:
        pos = tuple(rng.standard_normal(3) * scale * 6)
        pitch, yaw = float(rng.uniform(-80, 80)), float(rng.uniform(-180, 180))
    sc.set_camera(rrt.Camera(position=pos, pitch=pitch, yaw=yaw))
    w, h, spp, depth = 65, 33, 3, 12                     # odd sizes: the centre column/row give exactly axis-parallel rays when jitter is 0
    for trav, margin in ((0, 0.0), (1, 0.0078125), (1, 0.0)):
        hdr, rgba, st = _render(rrt, sc, w, h, spp, depth, traversal=trav, cull_margin=margin)
        ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                                        cull=trav, cull_margin=margin)
        a, b = hdr.view(np.uint32), ref.view(np.uint32)
        same = (a == b) | (np.isnan(hdr) & np.isnan(ref))            # NaN radiance (degenerate hits) must be NaN on both sides
        assert same.all(), (seed, trav, int((~same).sum()))
        assert np.array_equal(rgba, ref_rgba)
        for k in ("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "tex_clamped"):
            assert st[k] == rst[k], (k, seed, trav)


def _pbr_scene(rrt, n_target=40000, tex_size=32):
    """Atrium with materials that exercise every branch of the wgpu shader: mirrors, rough metals, glass, alpha cut-out,
    emitters, and textures in all six slots (base, transparency, roughness, metallic, emission, normal)."""
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.make_scene("atrium", n_target=n_target, tex_size=tex_size)
    rng = np.random.default_rng(77)
    texs = list(texs) + [rng.integers(0, 256, (16, 16, 4), dtype=np.uint8) for _ in range(3)]
    nt = len(texs)
    names = list(mats.keys())
    for i, k in enumerate(names):
        m = mats[k]
        m["roughness"] = [1.0, 0.05, 0.3, 0.6][i % 4]
        m["metallic"] = [0.0, 1.0, 0.5, 0.0, 0.0][i % 5]
        m["transmission"] = [0.0, 0.0, 0.0, 1.0, 0.6][i % 5]
        m["transparency"] = 1.0 if i % 6 else 0.5
        m["ior"] = [1.45, 1.33, 2.4][i % 3]
        if i % 7 == 3: m["roughness_tex_id"] = nt - 1
        if i % 7 == 4: m["metallic_tex_id"] = nt - 2
        if i % 7 == 5: m["normal_tex_id"] = nt - 3
        if i % 7 == 6: m["transparency_tex_id"] = nt - 1
        if i % 9 == 2: m["emission_tex_id"] = nt - 2
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    return sc


@pytest.mark.parametrize("traversal,margin", [(0, 0.0), (1, 0.0), (1, 0.0078125)])
def test_wgpu_material_model_matches_its_oracle(rrt, orc, traversal, margin):
    """the wgpu shader's material model (rt_compute.wgsl) as shading mode 1 -- kernel vs the oracle's
    restatement of the same shader, bit for bit (radiance, RGBA8, counters).  Pinned only against that restatement: the
    reference has no CPU implementation of this model and WGSL leaves the transcendental / filtering precision open."""
    sc = _pbr_scene(rrt)
    w, h, spp, depth = 128, 72, 4, 24
    hdr, rgba, st = _render(rrt, sc, w, h, spp, depth, traversal=traversal, cull_margin=margin, shading=rrt.SHADING_WGPU)
    ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                                    cull=traversal, cull_margin=margin, shading=1)
    for k in ("rays", "inner_steps", "tri_tests", "hits", "texel_fetches"):
        assert st[k] == rst[k], k
    a, b = hdr.view(np.uint32), ref.view(np.uint32)
    assert ((a == b) | (np.isnan(hdr) & np.isnan(ref))).all()
    assert np.array_equal(rgba, ref_rgba)
    assert st["rays"] > w * h * spp and np.isfinite(ref).mean() > 0.99
    # the model differs from the CPU backend's (Russian roulette, BSDF lobes): not the same image
    cpu, _, _ = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth, seed_mode=1)
    assert not np.array_equal(cpu, ref)


@pytest.mark.parametrize("kind,kw", [("cornell", {}), ("helmet", dict(n_target=4000, tex_size=16)), ("dragon", dict(n_target=30000)),
                                      ("atrium", dict(n_target=200000, tex_size=16))])
def test_device_bvh_builder_matches_host(rrt, orc, kind, kw):
    """BVH::build on the GPU emits the identical node array and triangle order (sign of zero aside)."""
    from rust_ray_tracing_amd import NODE, synth
    tris = synth.make_scene(kind, **kw)[0]
    host = rrt.Scene.from_arrays(tris, [rrt.material_default()])
    dev = rrt.Scene.from_arrays(tris, [rrt.material_default()], build_bvh=False)
    ms = dev.build_bvh_device(0)

    def canon(n):
        n = n.copy()
        for k in ("bounds_min", "bounds_max"):
            n[k] = n[k] + np.float32(0.0)
        return n.tobytes()
    assert dev.tris.tobytes() == host.tris.tobytes()
    assert canon(dev.bvh_nodes) == canon(host.bvh_nodes)
    assert ms > 0


def test_device_bvh_builder_random_soups(rrt):
    from rust_ray_tracing_amd import TRIANGLE
    rng = np.random.default_rng(5)
    for it in range(30):
        # sizes across every class of the device builder: one thread (<= 16), one wave (<= 2048), one workgroup (<= 8192), many workgroups
        n = int(rng.integers(1, 3000)) if it % 5 != 4 else int(rng.integers(3000, 40000))
        scale = float(rng.choice([1e-3, 1.0, 1e3]))
        p = rng.standard_normal((n, 1, 3)) * scale * 5 + rng.standard_normal((n, 3, 3)) * scale * rng.random((n, 1, 1))
        if it % 2:
            p = np.round(p / scale * 2) * scale / 2          # ties in the < comparisons, identical centroids
        t = np.zeros(n, dtype=TRIANGLE)
        t["vertices"]["position"] = p.astype(np.float32)
        host = rrt.Scene.from_arrays(t, [rrt.material_default()])
        dev = rrt.Scene.from_arrays(t, [rrt.material_default()], build_bvh=False)
        dev.build_bvh_device(0)
        assert dev.tris.tobytes() == host.tris.tobytes(), it
        a, b = host.bvh_nodes.copy(), dev.bvh_nodes.copy()
        for k in ("bounds_min", "bounds_max"):
            a[k] += np.float32(0); b[k] += np.float32(0)
        assert a.tobytes() == b.tobytes(), it


@pytest.mark.parametrize("n", [7, 40, 300, 1500, 9000, 40000])
def test_device_bvh_builder_overflowing_areas(rrt, n):
    """Boxes whose surface area overflows f32: every plane costs inf, `best >= parent` (inf) is false all the same, so bvh.rs:94-108
    splits at position 0.0 on axis 0 and k comes from counting -- the one path that does not take k from the bin counts.  Through
    every size class of the device builder (the chunked one counts in big_choose)."""
    from rust_ray_tracing_amd import TRIANGLE
    rng = np.random.default_rng(n)
    t = np.zeros(n, dtype=TRIANGLE)
    p = rng.standard_normal((n, 1, 3)) * 3e19 + rng.standard_normal((n, 3, 3)) * 1e18
    t["vertices"]["position"] = p.astype(np.float32)
    host = rrt.Scene.from_arrays(t, [rrt.material_default()])
    dev = rrt.Scene.from_arrays(t, [rrt.material_default()], build_bvh=False)
    dev.build_bvh_device(0)
    assert dev.tris.tobytes() == host.tris.tobytes()
    a, b = host.bvh_nodes.copy(), dev.bvh_nodes.copy()
    for k in ("bounds_min", "bounds_max"):
        a[k] += np.float32(0); b[k] += np.float32(0)
    assert a.tobytes() == b.tobytes()
    assert len(a) >= 3                                                    # it did split


def _chain_bvh(rrt, depth):
    """Hand-built BVH (the ABI accepts any well-formed tree): a chain in which every inner node has a FAR leaf child and a
    NEAR inner child for a ray along +x, so each level pushes one stack entry: stack occupancy = depth."""
    from rust_ray_tracing_amd import NODE, TRIANGLE
    n = depth + 1
    tris = np.zeros(n, dtype=TRIANGLE)
    xs = 1000.0 - np.arange(n, dtype=np.float32) * 2.0                    # triangle k sits at x = 1000 - 2k (far side first)
    for k in range(n):
        tris["vertices"]["position"][k] = [(xs[k], -1, -1), (xs[k], 1, -1), (xs[k], 0, 1)]
    tris["vertices"]["normal"] = (-1, 0, 0)
    nodes = np.zeros(2 * n - 1, dtype=NODE)

    def box(lo, hi):
        return (lo, -1.0, -1.0), (hi, 1.0, 1.0)
    nodes[0]["bounds_min"], nodes[0]["bounds_max"] = box(xs[-1], xs[0])
    nodes[0]["first_tri_or_child"] = 1
    for k in range(n - 1):                                                # children of inner node k at 2k+1 (leaf k), 2k+2 (rest)
        leaf, rest = 2 * k + 1, 2 * k + 2
        nodes[leaf]["bounds_min"], nodes[leaf]["bounds_max"] = box(xs[k], xs[k])
        nodes[leaf]["first_tri_or_child"], nodes[leaf]["num_tris"] = k, 1
        nodes[rest]["bounds_min"], nodes[rest]["bounds_max"] = box(xs[-1], xs[k + 1])
        if k == n - 2:
            nodes[rest]["first_tri_or_child"], nodes[rest]["num_tris"] = k + 1, 1
        else:
            nodes[rest]["first_tri_or_child"], nodes[rest]["num_tris"] = 2 * k + 3, 0
    sc = rrt.Scene.from_arrays(tris, [rrt.material_default()], build_bvh=False)
    sc.bvh_nodes = nodes
    sc.set_camera(rrt.Camera(position=(0.0, 0.0, 0.0), pitch=0.0, yaw=180.0))   # rays travel along -forward = +x
    return sc


def test_stack_spill_region_and_overflow_reporting(rrt, orc):
    """Traversal-stack entries 17..64 live in HBM; beyond 64 the kernel reports MIPT_ERR_STACK (the reference panics at 32)."""
    from rust_ray_tracing_amd import _lib as L
    sc = _chain_bvh(rrt, 60)                                               # needs ~60 pending entries: inside the capacity
    hdr, rgba, st = _render(rrt, sc, 32, 32, 2, 4)
    ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), [], sc.camera.uniform, 32, 32, 2, 4)
    assert st["max_stack"] == rst["max_stack"] and st["max_stack"] > 40 and rst["stack_overflows"] == 0
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))
    deep = _chain_bvh(rrt, 120)                                            # overflows both the kernel's 64 and the oracle's 64
    r = rrt.Renderer.new(rrt.RendererOptions(samples=1, max_ray_depth=2, output_image_dimensions=(16, 16), output_image_path="/dev/null"))
    with pytest.raises(rrt.MiptError) as e:
        r.render_buffers(deep)
    assert e.value.code == L.ERR_STACK
    _, _, ost = orc.render(deep.tris, deep.bvh_nodes, deep.materials_array(), [], deep.camera.uniform, 16, 16, 1, 2)
    assert ost["stack_overflows"] > 0


def test_negative_uv_is_clamped_and_counted(rrt, orc):
    """Texture::color_at panics on negative uv (texture.rs:33-38); kernel and oracle clamp the index identically
    and count the event."""
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.make_scene("helmet", n_target=2000, tex_size=16)
    tris["vertices"]["tex_coord_x"] -= 3.3
    tris["vertices"]["tex_coord_y"] -= 1.7
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    r = rrt.Renderer.new(rrt.RendererOptions(samples=2, max_ray_depth=6, output_image_dimensions=(64, 36), output_image_path="/dev/null"))
    hdr, rgba, st = r.render_buffers(sc, flags=rrt.FLAG_COUNT)
    ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, 64, 36, 2, 6)
    assert st["tex_clamped"] == rst["tex_clamped"] and st["tex_clamped"] > 0
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgba, ref_rgba)


def test_touched_lines_equal_the_oracles_visit_log(rrt, orc):
    """MIPT_FLAG_TOUCHED: the distinct 128-B lines a launch reads (the frame's compulsory traffic, bench.py roofline.unique_line_bytes)
    must be exactly the lines the oracle's record-visit log maps to under the device layout (pair order + triangle slots)."""
    from rust_ray_tracing_amd import _lib as L
    sc = _scene(rrt, "atrium", n_target=60000, tex_size=32)
    w, h, spp, depth = 96, 54, 2, 8
    _, _, st = _render(rrt, sc, w, h, spp, depth, flags=L.FLAG_COUNT | L.FLAG_TOUCHED, traversal=1)
    _, _, st_off = _render(rrt, sc, w, h, spp, depth, flags=L.FLAG_COUNT, traversal=1)
    assert st_off["touched_lines"] == [0, 0] and st["rays"] == st_off["rays"]
    # the oracle's log of the same frame
    lib = orc.load()
    lib.orc_visit_log.restype = C.c_uint64
    lib.orc_visit_log.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(orc.OrcTexture), C.c_uint32,
                                  C.c_void_p, C.POINTER(orc.OrcOptions), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
    opt = orc.OrcOptions(w, h, spp, depth, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0.0078125, 0)
    cap = 1 << 26
    log = np.zeros(cap, dtype=np.uint32)
    mats = sc.materials_array()
    n = lib.orc_visit_log(sc.tris.ctypes.data, len(sc.tris), sc.bvh_nodes.ctypes.data, len(sc.bvh_nodes), mats.ctypes.data, len(mats),
                          orc._tex_array(sc.textures), len(sc.textures), sc.camera.uniform.ctypes.data, C.byref(opt), 0, 1, w * h, log.ctypes.data, cap)
    assert n <= cap
    log = log[:n]
    log = log[log != 0xFFFFFFFF]
    kind, idx = log >> 30, (log & 0x3FFFFFFF).astype(np.int64)
    # device layout: pair records, then (line-aligned) the triangle stream
    mlib = rrt.load_diag()                      # the library-internal layout functions, re-exported by libmipt_diag.so
    n_pairs = (len(sc.bvh_nodes) - 1) // 2
    order = np.zeros(2 * n_pairs + 2, dtype=np.uint32)
    n_rec = C.c_uint32(0)
    assert mlib.mipt_internal_pair_order(sc.bvh_nodes.ctypes.data, len(sc.bvh_nodes), order.ctypes.data, order.size, C.byref(n_rec)) == 0
    order = order[: n_rec.value]
    rec = np.zeros(n_pairs, dtype=np.int64)
    rec[order[order != 0xFFFFFFFF]] = np.flatnonzero(order != 0xFFFFFFFF)
    pair_lines = (n_rec.value + 1) // 2
    slot = np.zeros(len(sc.tris), dtype=np.uint32)
    n_slots = C.c_uint32(0)
    assert mlib.mipt_internal_tri_slots(sc.bvh_nodes.ctypes.data, len(sc.bvh_nodes), len(sc.tris), slot.ctypes.data, C.byref(n_slots)) == 0
    geom = set((rec[idx[kind <= 1]] // 2).tolist()) | set((pair_lines + slot[idx[kind == 2]].astype(np.int64) // 2).tolist())
    attr = set((idx[kind == 3] // 2).tolist())
    assert st["touched_lines"] == [len(geom), len(attr)]

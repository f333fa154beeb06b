"""-m gpu: BASELINE.json configs[3] and [4] at their FULL sizes on the 10 M-triangle scene (VERDICT r1 missing-3).

configs[3]: ~10 M tris, 1920x1080, 256 spp, image tiles split over 8 ranks + gather -- here the 8 rank-PACKED tile shards are
            rendered in turn on the one GPU, de-interleaved with mipt_unpack_tiles, and must equal the single-launch frame
            bit for bit; a strided pixel sample must equal the oracle at 256 spp.
configs[4]: ~10 M tris, 4096x4096, 1024 spp sharded by samples over 8 ranks -- here ONE rank's 128-spp share with the
            per-sample seeds of reference src/renderer/backend/gpu/rt_compute.wgsl:102 (sample_begin = 1 + 128*r), whose
            un-normalised partial sum must equal the oracle's on a strided sample, bit for bit.
The scene is the seeded atrium stand-in (no assets ship with the reference)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def atrium10m(rrt):
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.atrium_scene(n_target=10_000_000, tex_size=1024)
    sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
    sc.tris_in_generator_order = tris            # (test_obj_file_to_pixels writes them out; sc.tris is reordered by the fetch below)
    sc.generator_materials = mats
    del tris
    # triangles up once, BVH::build + device layout in HBM (identical tree and layout to the host path: tests/test_gpu_scene_device.py
    # and test_device_resident_setup_at_full_size below); the Scene is left as BVH::build leaves it (nodes + reordered triangles)
    sc.upload_from_triangles(0, fetch_bvh=True)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    yield sc
    sc.release()


def test_device_resident_setup_at_full_size(rrt, atrium10m):
    """10 M triangles: the layout the GPU kernels built -- for mipt_scene_create_from_triangles and for mipt_scene_create given that
    tree -- is byte-identical (order-dependent 64-bit fingerprints of both buffers) to the host restatement's layout of the tree
    (tests/cpp/host_layout.cpp), config M's frame is the same frame, and the whole setup call is a fraction of a second."""
    import zlib
    from rust_ray_tracing_amd import _lib as L
    sc = atrium10m
    info = sc.info()
    print(f"device-resident setup: total {info['total_ms']:.0f} ms (upload {info['upload_ms']:.0f}, build {info['build_ms']:.1f}, layout {info['layout_ms']:.0f}); "
          f"{info['n_nodes']} nodes, {info['n_pair_records']} pair records, {info['geometry_bytes'] / 1e9:.2f} GB")
    assert info["built_on_device"] == 1 and info["n_tris"] == len(sc.tris) > 9_900_000
    assert info["total_ms"] < 600.0                      # bench.py reports the figure; this only guards against a regression to seconds
    diag = rrt.load_diag()
    h_dev = (C.c_uint64 * 2)()
    assert diag.mipt_diag_scene_hash(sc.upload(0), C.byref(h_dev)) == 0
    w, h, spp, depth = 1920, 1080, 8, 64
    opt = rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED, flags=L.FLAG_COUNT)
    f_dev, st_dev = _device_render(rrt, sc, opt, w * h * 3)
    crc_dev = zlib.crc32(f_dev.cpu().numpy().tobytes()) & 0xFFFFFFFF
    # the same tree handed to mipt_scene_create (the reference host's own BVH::build output would arrive like this), and the host
    # restatement of the layout as the reference for both
    host = rrt.Scene.from_arrays(sc.tris, list(sc.materials.values()), sc.textures, build_bvh=False)
    host.bvh_nodes = sc.bvh_nodes
    host.camera = sc.camera
    fp = host.host_layout_fingerprint()
    assert list(h_dev) == list(fp[:2])
    hh = host.upload(0)
    try:
        h_host = (C.c_uint64 * 2)()
        assert diag.mipt_diag_scene_hash(hh, C.byref(h_host)) == 0
        assert list(h_host) == list(fp[:2])
        hi = host.info()
        assert hi["built_on_device"] == 0 and hi["total_ms"] < 2000.0
        for k in ("n_nodes", "n_pair_records", "max_leaf", "geometry_bytes"):
            assert hi[k] == info[k], k
        f_host, st_host = _device_render(rrt, host, opt, w * h * 3)
        assert zlib.crc32(f_host.cpu().numpy().tobytes()) & 0xFFFFFFFF == crc_dev
        for k in ("rays", "inner_steps", "tri_tests", "hits"):
            assert st_dev[k] == st_host[k], k
        print(f"frame crc {crc_dev:08x}; mipt_scene_create with the caller's nodes: {hi['total_ms']:.0f} ms (host checks + layout kernels {hi['layout_ms']:.0f}, upload {hi['upload_ms']:.0f})")
    finally:
        host.release()


def test_obj_file_to_pixels(rrt, atrium10m, tmp_path):
    """BASELINE configs 2-4 name OBJ files (reference src/main.rs:21,36): the 10 M-triangle scene written as .obj + .mtl + PNGs, parsed by
    the chunked loader (mipt_obj_load_triangles, no host BVH build), built and laid out on the GPU, rendered at the metric's
    configuration -- the same frame, bit for bit, as the array path's."""
    import shutil
    import time
    import zlib
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import synth
    sc = atrium10m
    if shutil.disk_usage(str(tmp_path)).free < 8e9:
        pytest.skip("needs ~4 GB of scratch space for the OBJ file")
    t0 = time.time()
    path = synth.write_obj(str(tmp_path), "atrium10m", sc.tris_in_generator_order, sc.generator_materials, sc.textures)
    t1 = time.time()
    size = os.path.getsize(path)
    loaded = rrt.Scene.load(path, build_bvh=False)
    t2 = time.time()
    assert loaded is not None and len(loaded.tris) == len(sc.tris) and len(loaded.bvh_nodes) == 0
    assert loaded.tris.tobytes() == np.ascontiguousarray(sc.tris_in_generator_order).tobytes()
    os.remove(path)
    loaded.upload_from_triangles(0)
    t3 = time.time()
    loaded.camera = sc.camera
    w, h, spp, depth = 1920, 1080, 8, 64
    opt = rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED)
    try:
        f_obj, _ = _device_render(rrt, loaded, opt, w * h * 3)
        f_arr, _ = _device_render(rrt, sc, opt, w * h * 3)
        crc_obj = zlib.crc32(f_obj.cpu().numpy().tobytes()) & 0xFFFFFFFF
        crc_arr = zlib.crc32(f_arr.cpu().numpy().tobytes()) & 0xFFFFFFFF
        print(f"OBJ on-ramp: file {size / 1e9:.2f} GB written in {t1 - t0:.1f} s; Scene.load (parse + expand, incl. the Python copies) {t2 - t1:.1f} s; "
              f"device setup {loaded.info()['total_ms']:.0f} ms; frame crc {crc_obj:08x}")
        assert crc_obj == crc_arr
    finally:
        loaded.release()


def _device_render(rrt, sc, opt, n_floats):
    """mipt_render_device into a torch buffer (stays on the GPU)."""
    import torch
    from rust_ray_tracing_amd import _lib as L
    buf = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    st = L.MiptStats()
    L.check(rrt.load().mipt_render_device(sc.upload(0), L.ptr(sc.camera.uniform), C.byref(opt), C.c_void_p(buf.data_ptr()), None,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(st)), "mipt_render_device")
    return buf, st.as_dict()


def test_config4_1080p_256spp_eight_tile_shards(rrt, orc, atrium10m):
    import torch
    from rust_ray_tracing_amd import _lib as L
    sc = atrium10m
    assert len(sc.tris) > 9_900_000
    w, h, spp, depth, world = 1920, 1080, 256, 64, 8
    lib = rrt.load()
    full, st = _device_render(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED), w * h * 3)
    slots = int(lib.mipt_packed_pixels(w, h, world))
    d_all = torch.zeros(world * slots * 3, dtype=torch.float32, device="cuda")
    pixels = 0
    shard_ms = []
    for r in range(world):
        part, pst = _device_render(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED, flags=L.FLAG_PACKED,
                                                              tile_rank=r, tile_world=world), slots * 3)
        d_all[r * slots * 3:(r + 1) * slots * 3] = part
        pixels += pst["pixels"]
        shard_ms.append(pst["kernel_ms"])
    assert pixels == w * h
    frame = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    L.check(lib.mipt_unpack_tiles(C.c_void_p(d_all.data_ptr()), w, h, world, C.c_void_p(frame.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mipt_unpack_tiles")
    torch.cuda.synchronize()
    assert torch.equal(frame.view(torch.int32), full.view(torch.int32))               # stitched == single launch, every pixel
    assert int(frame.view(torch.int32).to(torch.int64).sum()) == int(full.view(torch.int32).to(torch.int64).sum())
    # oracle at the full 256 spp on a strided sample (>= 300 pixels)
    stride = 6899
    idx = np.arange(0, w * h, stride)
    assert len(idx) >= 300
    o, _, ost = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                           pix_stride=stride, want_rgba8=False)
    got = full.cpu().numpy().reshape(-1, 3)
    assert np.array_equal(o.reshape(-1, 3)[idx].view(np.uint32), got[idx].view(np.uint32))
    print(f"config4: full frame {st['kernel_ms']:.0f} ms, shards {[round(x) for x in shard_ms]} ms, oracle sample {len(idx)} px / {ost['rays']} rays")


def test_config5_4096_one_ranks_128spp_share(rrt, orc, atrium10m):
    from rust_ray_tracing_amd import _lib as L
    sc = atrium10m
    w = h = 4096
    share, depth, rank = 128, 64, 5
    s0 = 1 + share * rank                                                               # rt_compute.wgsl:102 seeds, samples 641..768
    part, st = _device_render(rrt, sc, rrt.make_options(w, h, share, depth, seed_mode=L.SEED_PER_SAMPLE, traversal=L.TRAVERSAL_CULLED,
                                                        flags=L.FLAG_SUM, sample_begin=s0), w * h * 3)
    assert st["pixels"] == w * h
    stride = 55001
    idx = np.arange(0, w * h, stride)
    assert len(idx) >= 300
    o, _, ost = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, share, depth, seed_mode=1,
                           sample_begin=s0, sum_only=1, pix_stride=stride, want_rgba8=False)
    got = part.cpu().numpy().reshape(-1, 3)
    assert np.array_equal(o.reshape(-1, 3)[idx].view(np.uint32), got[idx].view(np.uint32))
    assert float(np.isfinite(got).mean()) == 1.0
    print(f"config5 share: {st['kernel_ms']:.0f} ms for {w}x{h}x{share} spp, oracle sample {len(idx)} px / {ost['rays']} rays")


def test_config_M_culled_traversal_renders_the_reference_traversals_frame(rrt, atrium10m):
    """The metric's configuration (1920x1080, 8 spp, depth 64) on the 10 M-triangle scene: the recommended arm -- best-hit culling with
    the 2^-7 relative margin (rt_compute.wgsl:341-349 + margin) -- must produce the frame of the CPU backend's own un-culled traversal
    (reference src/renderer/backend/cpu/ray.rs:69-81, the Renderer default), all 2 073 600 pixels bit for bit, while visiting fewer
    nodes.  (bench.py re-checks the same identity on every run; INTEGRATION.md recommends the culled arm on this evidence.)"""
    import torch
    from rust_ray_tracing_amd import _lib as L
    sc = atrium10m
    w, h, spp, depth = 1920, 1080, 8, 64
    ref, rst = _device_render(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_REFERENCE, flags=L.FLAG_COUNT), w * h * 3)
    cul, cst = _device_render(rrt, sc, rrt.make_options(w, h, spp, depth, traversal=L.TRAVERSAL_CULLED, cull_margin=L.CULL_MARGIN_SAFE,
                                                        flags=L.FLAG_COUNT), w * h * 3)
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int32), cul.view(torch.int32))
    assert rst["rays"] == cst["rays"] and rst["hits"] == cst["hits"] and rst["pixels"] == w * h
    assert cst["inner_steps"] < 0.7 * rst["inner_steps"] and cst["tri_tests"] <= rst["tri_tests"]

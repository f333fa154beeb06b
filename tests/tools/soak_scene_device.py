"""GPU box, builder-run: soak of the device-resident scene setup on random soups -- for every soup the tree of
mipt_scene_create_from_triangles equals the host builder's, and the device layout built by the GPU kernels (csrc/scene_device.hip),
for that entry and for mipt_scene_create given the tree, is byte-identical (order-dependent 64-bit fingerprints of both buffers,
libmipt_diag.so) to the host restatement's layout of that tree (tests/cpp/host_layout.cpp).
    python tests/tools/soak_scene_device.py [n_soups] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rust_ray_tracing_amd as rrt  # noqa: E402
from rust_ray_tracing_amd import TRIANGLE  # noqa: E402

n_soups = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
diag = rrt.load_diag()


def fingerprint(handle):
    h = (C.c_uint64 * 2)()
    s = (C.c_uint64 * 2)()
    assert diag.mipt_diag_scene_hash(handle, C.byref(h)) == 0 and diag.mipt_diag_scene_sizes(handle, C.byref(s)) == 0
    return (int(h[0]), int(h[1]), int(s[0]), int(s[1]))


total = bad = 0
for it in range(n_soups):
    n = int(np.exp(rng.uniform(0.0, np.log(120_000.0)))) + 1
    t = np.zeros(n, dtype=TRIANGLE)
    kind = it % 5
    c = rng.uniform(-6, 6, (n, 1, 3)).astype(np.float32)
    if kind == 1:
        c = np.round(c * 2) / 2                                   # many equal centroids: ties in the partition
    if kind == 2:
        c[:, :, int(rng.integers(3))] = 1.25                      # a constant axis
    t["vertices"]["position"] = c + rng.normal(0, 0.2 if kind != 3 else 0.0, (n, 3, 3)).astype(np.float32)   # kind 3: zero-area triangles
    if kind == 4 and n > 4:
        t[n // 3:] = t[n // 3]                                    # copies of one triangle
    t["vertices"]["normal"] = rng.normal(0, 1, (n, 3, 3)).astype(np.float32)
    t["vertices"]["tex_coord_x"] = rng.uniform(0, 4, (n, 3)).astype(np.float32)
    dev = rrt.Scene.from_arrays(t, [rrt.material_default()], [], build_bvh=False)
    hd = dev.upload_from_triangles(0, fetch_bvh=True)
    host = rrt.Scene.from_arrays(t, [rrt.material_default()], [])
    same_tree = (len(dev.bvh_nodes) == len(host.bvh_nodes) and np.array_equal(dev.bvh_nodes["first_tri_or_child"], host.bvh_nodes["first_tri_or_child"])
                 and np.array_equal(dev.bvh_nodes["num_tris"], host.bvh_nodes["num_tris"]) and np.array_equal(dev.bvh_nodes["bounds_min"], host.bvh_nodes["bounds_min"])
                 and np.array_equal(dev.bvh_nodes["bounds_max"], host.bvh_nodes["bounds_max"]) and dev.tris.tobytes() == host.tris.tobytes())
    ref = rrt.Scene.from_arrays(dev.tris, [rrt.material_default()], [], build_bvh=False)
    ref.bvh_nodes = dev.bvh_nodes.copy()
    want = ref.host_layout_fingerprint()
    same_layout = fingerprint(hd) == want and fingerprint(ref.upload(0)) == want
    total += n
    if not (same_tree and same_layout):
        bad += 1
        print(f"soup {it}: n={n} kind={kind} tree {same_tree} layout {same_layout}", flush=True)
    dev.release(); ref.release()
print(f"{n_soups} soups, {total} triangles: {bad} differing", flush=True)

"""GPU box: device BVH builder vs host builder on many random triangle soups (sizes across every builder class, with ties,
identical centroids, flat and degenerate triangles, boxes whose area overflows).  Prints the number of soups whose node array or triangle order differs; must be 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import TRIANGLE
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7")))
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
tot = 0
for it in range(n_iter):
    kind = it % 6
    n = int(rng.integers(1, 600)) if kind < 2 else int(rng.integers(600, 9000)) if kind < 4 else int(rng.integers(9000, 120000))
    scale = float(rng.choice([1e-3, 1.0, 1e3]))
    if it % 10 == 9: scale = 1e19                                              # surface areas overflow: no plane is usable, the split falls back to 0.0 on axis 0
    spread = float(rng.choice([0.05, 1.0, 20.0]))
    p = rng.standard_normal((n, 1, 3)) * scale * spread + rng.standard_normal((n, 3, 3)) * scale * rng.random((n, 1, 1))
    mode = it % 5
    if mode == 1: p = np.round(p / scale * 2) * scale / 2                      # ties in the < comparisons, identical centroids
    if mode == 2: p[:, :, int(rng.integers(0, 3))] = 0.25 * scale              # everything in one plane: an unused axis
    if mode == 3: p[rng.random(n) < 0.3] = p[0]                                # many copies of one triangle
    if mode == 4: p[:, 1] = p[:, 0]                                            # degenerate (zero-area) triangles
    t = np.zeros(n, dtype=TRIANGLE)
    t["vertices"]["position"] = p.astype(np.float32)
    host = rrt.Scene.from_arrays(t, [rrt.material_default()])
    dev = rrt.Scene.from_arrays(t, [rrt.material_default()], build_bvh=False)
    dev.build_bvh_device(0)
    a, b = host.bvh_nodes.copy(), dev.bvh_nodes.copy()
    for k in ("bounds_min", "bounds_max"):
        a[k] += np.float32(0); b[k] += np.float32(0)
    same = dev.tris.tobytes() == host.tris.tobytes() and a.tobytes() == b.tobytes()
    tot += n
    if not same:
        bad += 1
        print("MISMATCH soup", it, "n", n, "mode", mode, flush=True)
    if it % 50 == 49: print(it + 1, "soups,", tot, "triangles, mismatching", bad, flush=True)
print("soups", n_iter, "triangles", tot, "mismatching", bad)

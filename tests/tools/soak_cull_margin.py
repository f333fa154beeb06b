"""GPU box: culled (margin 2^-7) vs reference traversal over many random camera poses and scenes -- the identity is empirical
(DESIGN.md section 2), so it is soaked beyond the one bench view.  Prints the number of differing pixels per view."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
lib = rrt.load()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
w, h, spp, depth = 960, 540, 4, 64
buf_a = np.zeros(w * h * 3, dtype=np.float32)
buf_b = np.zeros(w * h * 3, dtype=np.float32)
total_bad = 0
views = 0
for kind, kw, n_views, box in [("atrium", dict(n_target=1000000, tex_size=256), 24, 14.0), ("dragon", dict(n_target=400000), 10, 6.0),
                               ("helmet", dict(n_target=15000, tex_size=64), 6, 4.0)]:
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    for v in range(n_views):
        if v == 0:
            pos, pitch, yaw = cam
        else:
            pos = tuple(float(x) for x in (np.array(cam[0]) + rng.uniform(-box, box, 3) * np.array([1.0, 0.25, 1.0])))
            pitch, yaw = float(rng.uniform(-60, 60)), float(rng.uniform(-180, 180))
        sc.set_camera(rrt.Camera(position=pos, pitch=pitch, yaw=yaw))
        hnd = sc.upload(0)
        for trav, margin, buf in ((0, 0.0, buf_a), (1, 0.0078125, buf_b)):
            o = rrt.make_options(w, h, spp, depth, traversal=trav, cull_margin=margin)
            st = L.MiptStats()
            L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        a, b = buf_a.view(np.uint32).reshape(-1, 3), buf_b.view(np.uint32).reshape(-1, 3)
        bad = int(((a != b) & ~(np.isnan(buf_a.reshape(-1, 3)) & np.isnan(buf_b.reshape(-1, 3)))).any(1).sum())
        total_bad += bad
        views += 1
        print(kind, "view", v, "pos", tuple(round(x, 2) for x in pos), "pitch", round(pitch, 1), "yaw", round(yaw, 1), "differing pixels", bad, flush=True)
print("views", views, "pixels", views * w * h, "differing", total_bad)

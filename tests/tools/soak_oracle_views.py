"""GPU box: GPU frame vs the CPU oracle on a strided pixel sample, over many random camera poses and scenes (beyond the fixed
views of the parity tests).  Prints mismatching pixels per view; all must be 0."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
from oracle import orc
lib = rrt.load()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "2")))
w, h, spp, depth, stride = 960, 540, 4, 64, 53
buf = np.zeros(w * h * 3, dtype=np.float32)
bad_total = 0
n_px = 0
for kind, kw, n_views, box in [("atrium", dict(n_target=300000, tex_size=128), 12, 14.0), ("dragon", dict(n_target=200000), 8, 6.0),
                               ("helmet", dict(n_target=15000, tex_size=64), 6, 4.0), ("cornell", {}, 4, 0.8)]:
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
        o = rrt.make_options(w, h, spp, depth, traversal=1, cull_margin=0.0078125)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ref, _, _ = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth,
                               cull=0, pix_begin=0, pix_stride=stride, want_rgba8=False)
        sel = np.arange(0, w * h, stride)
        a, b = buf.reshape(-1, 3)[sel], ref.reshape(-1, 3)[sel]
        bad = int((((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).any(1)).sum())
        bad_total += bad
        n_px += len(sel)
        print(kind, "view", v, "pitch", round(pitch, 1), "yaw", round(yaw, 1), "sampled", len(sel), "mismatching", bad, flush=True)
print("sampled pixels", n_px, "mismatching", bad_total)

"""GPU box: run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU` -- renders several configs twice
(counting build for wave-level iteration counts, then the production build whose dispatch the PMC row describes)."""
import sys, os, ctypes as C, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
cfgs = [(10_000_000, 1920, 1080, 8, 64), (10_000_000, 1920, 1080, 8, 2), (10_000_000, 1920, 1080, 2, 64), (1_000_000, 1920, 1080, 8, 64), (100_000, 1920, 1080, 8, 6), (1_000_000, 1280, 720, 4, 16)]
out = []
scenes = {}
for (n, w, h, spp, depth) in cfgs:
    if n not in scenes:
        tris, mats, texs, cam = synth.atrium_scene(n_target=n, tex_size=512)
        sc = rrt.Scene.from_arrays(tris, mats, texs)
        sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
        scenes[n] = sc
    sc = scenes[n]
    hnd = sc.upload(0)
    buf = np.zeros(w * h * 3, dtype=np.float32)
    res = {}
    for flags in (L.FLAG_COUNT, 0):
        o = rrt.make_options(w, h, spp, depth, traversal=1, flags=flags)
        st = L.MiptStats()
        L.check(rrt.load().mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        d = st.as_dict()
        if flags:
            res.update(iters=d["diag"][0], iters_inner=d["diag"][3], iters_leaf=d["diag"][4], services=d["diag"][5], rays=d["rays"])
        else:
            res.update(ms=d["kernel_ms"])
    out.append(res)
    print(json.dumps(res), flush=True)

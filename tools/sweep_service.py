"""GPU box: kernel time vs the service-pass threshold (MIPT_SERVICE_NUM/DEN) for the three named 1080p configs."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
lib = rrt.load()
w, h = 1920, 1080
buf = np.zeros(w * h * 3, dtype=np.float32)
cases = [("config2 helmet", "helmet", dict(n_target=15000, tex_size=1024), 16), ("config3 dragon", "dragon", dict(n_target=870000), 64),
         ("configM atrium", "atrium", dict(n_target=int(os.environ.get("SWEEP_TRIS", "10000000")), tex_size=1024), 8)]
for name, kind, kw, spp in cases:
    if os.environ.get("SWEEP_ONLY") and os.environ["SWEEP_ONLY"] not in name: continue
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    hnd = sc.upload(0)
    res = {}
    for num, den in [(1, 8), (3, 16), (1, 4), (5, 16), (3, 8), (1, 2), (5, 8), (3, 4)]:
        os.environ["MIPT_SERVICE_NUM"], os.environ["MIPT_SERVICE_DEN"] = str(num), str(den)
        ts = []
        for rep in range(int(os.environ.get("SWEEP_REPS", "3"))):
            o = rrt.make_options(w, h, spp, 64, traversal=1)
            st = L.MiptStats()
            L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
            ts.append(st.kernel_ms)
        res[f"{num}/{den}"] = round(min(ts), 2)
    print(name, res, flush=True)
    if os.environ.get("SWEEP_LEAF"):                       # leaf-phase period (MIPT_LEAF_PERIOD) at the default service threshold
        del os.environ["MIPT_SERVICE_NUM"], os.environ["MIPT_SERVICE_DEN"]
        res = {}
        for per, lanes in [tuple(int(v) for v in x.split("/")) for x in os.environ["SWEEP_LEAF"].split(",")]:   # "4/16,4/65,..."
            os.environ["MIPT_LEAF_PERIOD"], os.environ["MIPT_LEAF_DEN"] = str(per), str(lanes)
            ts = []
            for rep in range(int(os.environ.get("SWEEP_REPS", "3"))):
                o = rrt.make_options(w, h, spp, 64, traversal=1)
                st = L.MiptStats()
                L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
                ts.append(st.kernel_ms)
            res[f"period {per} release 1/{lanes}"] = round(min(ts), 2)
        del os.environ["MIPT_LEAF_PERIOD"], os.environ["MIPT_LEAF_DEN"]
        print(name, res, flush=True)
    del sc

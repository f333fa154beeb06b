"""GPU box: kernel time of ONE rank's tile share (world = 2, 4, 8) with lane regrouping off / on, vs blocks per CU and park threshold."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
n_tris = int(os.environ.get("SWEEP_TRIS", "10000000"))
tris, mats, texs, cam = synth.atrium_scene(n_target=n_tris, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
hnd = sc.upload(0)
lib = rrt.load()
w, h = 1920, 1080
buf = np.zeros(w * h * 3, dtype=np.float32)


def run(world, env):
    for k in ("MIPT_REGROUP", "MIPT_PARK_LANES", "MIPT_BLOCKS_PER_CU", "MIPT_REGROUP_FRAC"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for rep in range(3):
        o = rrt.make_options(w, h, 8, 64, traversal=1, flags=L.FLAG_PACKED if world > 1 else 0, tile_rank=0, tile_world=world)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ts.append(st.kernel_ms)
    return round(min(ts), 2), buf.copy()


for world in (8, 4, 2, 1):
    base, ref = run(world, {"MIPT_REGROUP": "0"})
    print("world", world, "off(auto bpc)", base, flush=True)
    for bpc in ("3", "5"):
        for pl in ("32",):
            for fr in ("0.25", "0.5", "0.75"):
                t, out = run(world, {"MIPT_REGROUP": "1", "MIPT_PARK_LANES": pl, "MIPT_BLOCKS_PER_CU": bpc, "MIPT_REGROUP_FRAC": fr})
                same = bool(np.array_equal(out.view(np.uint32), ref.view(np.uint32)))
                print("world", world, "regroup bpc", bpc, "park_lanes", pl, "frac", fr, t, "identical" if same else "DIFFERENT", flush=True)
    t, out = run(world, {})
    print("world", world, "auto", t, "identical" if np.array_equal(out.view(np.uint32), ref.view(np.uint32)) else "DIFFERENT", flush=True)

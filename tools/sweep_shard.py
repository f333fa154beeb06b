"""GPU box: kernel time of ONE rank's tile share (world = 1, 2, 4, 8) vs blocks per CU -- the strong-scaling regime."""
import sys, os, ctypes as C, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
tris, mats, texs, cam = synth.atrium_scene(n_target=10_000_000, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
hnd = sc.upload(0)
lib = rrt.load()
w, h = 1920, 1080
buf = np.zeros(w * h * 3, dtype=np.float32)
for world in (1, 2, 4, 8):
    res = {}
    for bpc in ("1", "2", "3", "4", "5", "auto"):
        if bpc == "auto": os.environ.pop("MIPT_BLOCKS_PER_CU", None)
        else: os.environ["MIPT_BLOCKS_PER_CU"] = bpc
        ts = []
        for rep in range(3):
            o = rrt.make_options(w, h, 8, 64, traversal=1, flags=L.FLAG_PACKED if world > 1 else 0, tile_rank=0, tile_world=world)
            st = L.MiptStats()
            L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
            ts.append(st.kernel_ms)
        res[bpc] = round(min(ts), 2)
    print("world", world, res, flush=True)

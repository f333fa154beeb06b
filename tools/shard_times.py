"""GPU box: kernel ms of rank 0's tile share of config M for world = 1, 2, 4, 8 (auto grid) -- run once per MIPT_LIB build variant."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
tris, mats, texs, cam = synth.atrium_scene(n_target=int(os.environ.get("SWEEP_TRIS", "10000000")), tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
hnd = sc.upload(0)
lib = rrt.load()
w, h = 1920, 1080
buf = np.zeros(w * h * 3, dtype=np.float32)
out = {}
for world in (1, 2, 4, 8):
    ts = []
    for rep in range(4):
        o = rrt.make_options(w, h, 8, 64, traversal=1, flags=L.FLAG_PACKED if world > 1 else 0, tile_rank=0, tile_world=world)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ts.append(st.kernel_ms)
    out[world] = round(min(ts), 2)
    if world == 1:
        import zlib
        print("frame crc", zlib.crc32(buf.tobytes()))
print(os.environ.get("MIPT_LIB", "base").split("/")[-1], out, flush=True)

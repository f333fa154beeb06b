import ctypes as C, os, sys, zlib
sys.path.insert(0, os.getcwd())
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
tris, mats, texs, cam = synth.atrium_scene(n_target=int(os.environ.get("TRIS", "10000000")), tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False); del tris
sc.build_bvh_device(0)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
lib = rrt.load(); hnd = sc.upload(0)
w, h = 1920, 1080
buf = np.zeros(w * h * 3, dtype=np.float32)
def run(flags, reps, trav=1, spp=8):
    ts = []
    for _ in range(reps):
        o = rrt.make_options(w, h, spp, 64, traversal=trav, flags=flags)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ts.append(round(st.kernel_ms, 2))
    return ts, zlib.crc32(buf.tobytes())
print("plain      ", run(0, 4))
print("hint       ", run(L.FLAG_ORDER_HINT, 6))
print("plain      ", run(0, 3))
print("hint       ", run(L.FLAG_ORDER_HINT, 4))
print("unculled plain", run(0, 2, trav=0)); print("unculled hint ", run(L.FLAG_ORDER_HINT, 4, trav=0))
o = rrt.make_options(w, h, 8, 64, traversal=1, flags=L.FLAG_ORDER_HINT | L.FLAG_COUNT); st = L.MiptStats()
L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
dg = list(st.diag); print("hint count build: tail", dg[10] / max(dg[8], 1), "kernel", st.kernel_ms)
o = rrt.make_options(w, h, 8, 64, traversal=1, flags=L.FLAG_COUNT); st = L.MiptStats()
L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
dg = list(st.diag); print("plain count build: tail", dg[10] / max(dg[8], 1), "kernel", st.kernel_ms)

"""GPU box: MIPT_QUAD=1 (128-B two-level records) vs the pair kernel -- identical frames on several scenes, then config M timing."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
lib = rrt.load()


def scene(kind, **kw):
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    return sc


def render(sc, quad, w, h, spp, depth, trav, reps=1, world=0, flags=0):
    os.environ["MIPT_QUAD"] = "1" if quad else "0"
    sc.release()                                 # a fresh device scene per mode: the quad records are built at scene creation
    hnd = sc.upload(0)
    buf = np.zeros(w * h * 3, dtype=np.float32)
    ts = []
    for _ in range(reps):
        o = rrt.make_options(w, h, spp, depth, traversal=trav, flags=flags | (L.FLAG_PACKED if world > 1 else 0), tile_rank=0, tile_world=world)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ts.append(st.kernel_ms)
    return buf.copy(), min(ts), sc


ok = True
for name, kind, kw, w, h, spp, depth in [("cornell", "cornell", {}, 256, 256, 4, 64), ("helmet", "helmet", dict(n_target=15000, tex_size=64), 640, 360, 4, 32),
                                         ("dragon", "dragon", dict(n_target=100000), 640, 360, 4, 32), ("atrium1M", "atrium", dict(n_target=1000000, tex_size=256), 640, 360, 4, 64)]:
    sc = scene(kind, **kw)
    for trav in (0, 1):
        a, ta, _ = render(sc, False, w, h, spp, depth, trav)
        b, tb, _ = render(sc, True, w, h, spp, depth, trav)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        ok &= same
        print(name, "traversal", trav, "pair", round(ta, 2), "ms  quad", round(tb, 2), "ms", "identical" if same else f"DIFFERENT ({int((a != b).sum())} floats)", flush=True)
sc = scene("atrium", n_target=int(os.environ.get("SWEEP_TRIS", "10000000")), tex_size=1024)
for world in (1, 4, 8):
    a, ta, _ = render(sc, False, 1920, 1080, 8, 64, 1, reps=3, world=world)
    b, tb, _ = render(sc, True, 1920, 1080, 8, 64, 1, reps=3, world=world)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ok &= same
    print("config M world", world, "pair", round(ta, 2), "ms  quad", round(tb, 2), "ms", "identical" if same else "DIFFERENT", flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")

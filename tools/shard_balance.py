"""GPU box: how evenly does the tile split load the ranks?  One MI355X renders EVERY rank's share of config M in turn
(world 2 / 4 / 8) and reports kernel ms per rank, the slowest share over the mean share (what the frame pays), and rays per rank."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L

tris, mats, texs, cam = synth.atrium_scene(n_target=int(os.environ.get("TRIS", "10000000")), tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
del tris
sc.build_bvh_device(0)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
lib = rrt.load()
hnd = sc.upload(0)
w, h, spp, depth = 1920, 1080, 8, 64
for world in (2, 4, 8):
    n_out = int(lib.mipt_packed_pixels(w, h, world))
    buf = np.zeros(n_out * 3, dtype=np.float32)
    ms, rays = [], []
    for rank in range(world):
        best = 1e9
        for rep in range(3):
            o = rrt.make_options(w, h, spp, depth, traversal=1, flags=L.FLAG_PACKED, tile_rank=rank, tile_world=world)
            st = L.MiptStats()
            L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
            best = min(best, st.kernel_ms)
        o = rrt.make_options(w, h, spp, depth, traversal=1, flags=L.FLAG_PACKED | L.FLAG_COUNT, tile_rank=rank, tile_world=world)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ms.append(best); rays.append(st.rays)
    print(f"world {world}: kernel ms per rank {[round(x, 2) for x in ms]}  slowest/mean {max(ms) / np.mean(ms):.3f}  "
          f"rays max/mean {max(rays) / np.mean(rays):.3f}", flush=True)

"""GPU box: wave-occupancy diagnostics of the trace kernel on config M."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tris, mats, texs, cam = synth.atrium_scene(n_target=n_tris, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
h = sc.upload(0)
buf = np.zeros(1920 * 1080 * 3, dtype=np.float32)
for flags in (L.FLAG_COUNT, 0, 0):
    o = rrt.make_options(1920, 1080, 8, 64, traversal=1, flags=flags)
    st = L.MiptStats()
    L.check(rrt.load().mipt_render(h, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
    d = st.as_dict()
    print(d)
    if flags:
        it, il, ll, iti, itl, sv, svl, cs, ct, cm, ctail = d["diag"]
        print(f"tail: {ctail/ct:.1%} of wave-cycles are after the wave first saw an empty queue")
        print(f"memory wait per traversal iter {cm/it:.0f} cycles")
        print(f"wave-cycles: service {cs:.3e} of total {ct:.3e} = {cs/ct:.1%}; per service pass {cs/max(sv,1):.0f} cycles; per traversal iter {(ct-cs)/it:.0f} cycles")
        print(f"iters {it:.3e}  inner lanes/iter {il/it:.1f}  leaf lanes/iter {ll/it:.1f}  inner-branch iters {iti/it:.2%}  leaf-branch iters {itl/it:.2%}")
        print(f"lanes per inner-branch exec {il/max(iti,1):.1f}, per leaf-branch exec {ll/max(itl,1):.1f}; services {sv:.3e}, lanes/service {svl/max(sv,1):.1f}")

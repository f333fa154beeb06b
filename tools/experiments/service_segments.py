import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
tris, mats, texs, cam = synth.atrium_scene(n_target=10_000_000, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False); sc.build_bvh_device(0)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
h = sc.upload(0)
buf = np.zeros(1920*1080*3, dtype=np.float32)
o = rrt.make_options(1920, 1080, 8, 64, traversal=1, flags=L.FLAG_COUNT)
st = L.MiptStats()
L.check(rrt.load().mipt_render(h, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
d = st.as_dict()["diag"]
A,B,Cc,D,E = d[0],d[1],d[2],d[3],d[4]; serv=d[5]; total_serv=d[7]; total=d[8]; F=d[9]
print("kernel ms", st.kernel_ms, "services", serv)
for n,v in (("A attr wait",A),("B interp+material wait",B),("C texels",Cc),("D rng/scatter/end",D),("E pixel fetch",E),("F camera+start",F)):
    print(f"{n:28s} {v/serv:9.0f} cycles/pass  {v/total_serv:6.1%} of service")
print("service total per pass", total_serv/serv, "share of wave cycles", total_serv/total)

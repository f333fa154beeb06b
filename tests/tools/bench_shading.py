"""GPU box: config M with the wgpu shader's material model (shading mode 1) -- measurement row for DESIGN.md."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
from oracle import orc
tris, mats, texs, cam = synth.atrium_scene(n_target=10_000_000, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
h = sc.upload(0)
w, hh, spp, depth = 1920, 1080, 8, 64
buf = np.zeros(w * hh * 3, dtype=np.float32)
for shading in (0, 1):
    for flags in (L.FLAG_COUNT, 0, 0):
        o = rrt.make_options(w, hh, spp, depth, seed_mode=1, traversal=1, flags=flags, shading=shading)
        st = L.MiptStats()
        L.check(rrt.load().mipt_render(h, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        d = st.as_dict()
        if flags: cnt = d
        else: print(f"shading {shading}: {cnt['rays']} rays, {d['kernel_ms']:.1f} ms, {cnt['rays'] / d['kernel_ms'] / 1e3:.0f} Mray/s, "
                    f"{cnt['inner_steps'] / cnt['rays']:.1f} inner + {cnt['tri_tests'] / cnt['rays']:.1f} tri per ray, {cnt['rays'] / (w * hh * spp):.2f} rays/path", flush=True)
    stride = 9973
    ref, _, _ = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, hh, spp, depth, seed_mode=1,
                           cull=1, cull_margin=0.0078125, shading=shading, pix_stride=stride, want_rgba8=False)
    idx = np.arange(0, w * hh, stride)
    a = buf.reshape(-1, 3)[idx]; b = ref.reshape(-1, 3)[idx]
    print("   sample vs oracle bit-exact:", bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()), len(idx), "pixels")

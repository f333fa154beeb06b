"""Debug helper (GPU box): render one case with the HIP path and the oracle, report where they differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth
from oracle import orc

kind = sys.argv[1]; n_target = int(sys.argv[2]); w, h, spp, depth, trav = map(int, sys.argv[3:8])
kw = {} if kind == "cornell" else dict(n_target=n_target)
if kind in ("helmet", "atrium"): kw["tex_size"] = 64
tris, mats, texs, cam = synth.make_scene(kind, **kw)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
r = rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h), output_image_path="/dev/null", traversal=trav))
hdr, rgba, st = r.render_buffers(sc, flags=rrt.FLAG_COUNT)
ref, ref_rgba, rst = orc.render(sc.tris, sc.bvh_nodes, sc.materials_array(), sc.textures, sc.camera.uniform, w, h, spp, depth, cull=trav)
print("gpu", st); print("cpu", rst)
d = (hdr.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
print("differing pixels:", int(d.sum()), "of", w * h)
ys, xs = np.nonzero(d)
for y, x in list(zip(ys, xs))[:10]:
    print((x, y), hdr[y, x], ref[y, x])

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth
tris = synth.make_scene("atrium", n_target=10_000_000, tex_size=16)[0]
b = rrt.Scene.from_arrays(tris, [rrt.material_default()], build_bvh=False)
print("device build ms", b.build_bvh_device(0))

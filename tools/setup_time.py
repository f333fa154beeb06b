"""GPU box: mipt_scene_create_from_triangles on the 10 M-triangle scene, three times in one process (first call = cold HIP runtime);
prints MiptSceneInfo of each; then mipt_scene_create with the tree the last one built (the reference host's own BVH::build output
would arrive like this), twice.  Under `rocprofv3 --hip-trace --stats` the API table shows what the first call's extra time is made of."""
import os, sys, time
if os.environ.get("SETUP_TIME_TORCH"):            # like bench.py: torch has initialised the HIP runtime before the first call
    import torch
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth
tris, mats, texs, cam = synth.atrium_scene(n_target=int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, tex_size=1024)
for rep in range(3):
    sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
    t0 = time.time()
    sc.upload_from_triangles(0)
    dt = time.time() - t0
    i = sc.info()
    print(f"call {rep}: {dt * 1e3:.0f} ms  (upload {i['upload_ms']:.0f}, build {i['build_ms']:.1f}, layout {i['layout_ms']:.1f}, total {i['total_ms']:.0f})", flush=True)
    sc.release()
sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
sc.upload_from_triangles(0, fetch_bvh=True)               # leaves the Scene as BVH::build would: nodes + reordered triangles
sc.release()
for rep in range(2):
    t0 = time.time()
    sc.upload(0)
    dt = time.time() - t0
    i = sc.info()
    print(f"mipt_scene_create, caller's nodes, call {rep}: {dt * 1e3:.0f} ms  (host checks + layout kernels {i['layout_ms']:.0f}, upload {i['upload_ms']:.0f}, total {i['total_ms']:.0f})", flush=True)
    sc.release()

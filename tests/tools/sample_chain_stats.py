"""CPU: how predictable is the number of RNG draws a sample consumes?  (a pixel's samples form a chain only through
the RNG state, which advances by 2 + 6 x (scatter events) draws per sample; a lane that knew the count could start the next sample
early with a jumped-ahead state.)  For a strided set of pixels of the config-M view on a smaller atrium, the oracle's per-ray debug
records give every sample's ray count; printed: the cost distribution over pixels and, for the heaviest pixels, how many of their
samples run to the depth limit."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth
from oracle import orc
n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 997
tris, mats, texs, cam = synth.atrium_scene(n_target=n_tris, tex_size=64)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
w, h, spp, depth = 1920, 1080, 8, 64
lib = orc.load()
lib.orc_debug_pixel.restype = C.c_uint32
m = np.ascontiguousarray(sc.materials_array())
opt = orc.OrcOptions(w, h, spp, depth, 0, 1, orc.LIBM_GLIBC235, 1, 0, 0, 0, 0, 0, 0, 0.0078125, 0, 0)
rec = np.zeros((spp * (depth + 1) + 8, 8), dtype=np.float32)
cam_pos = np.array(cam[0], dtype=np.float32)
per_pixel = []
texarr = orc._tex_array([np.ascontiguousarray(t) for t in sc.textures])
for pix in range(0, w * h, stride):
    n = lib.orc_debug_pixel(C.c_void_p(sc.tris.ctypes.data), C.c_uint32(len(sc.tris)), C.c_void_p(sc.bvh_nodes.ctypes.data), C.c_uint32(len(sc.bvh_nodes)),
                            C.c_void_p(m.ctypes.data), C.c_uint32(len(m)), texarr, C.c_uint32(len(sc.textures)),
                            C.c_void_p(sc.camera.uniform.ctypes.data), C.byref(opt), C.c_uint64(pix), C.c_void_p(rec.ctypes.data), C.c_uint32(len(rec)), None)
    r = rec[:n]
    starts = np.flatnonzero((r[:, :3] == cam_pos).all(1))
    counts = np.diff(np.append(starts, n))
    per_pixel.append(counts)
tot = np.array([c.sum() for c in per_pixel])
order = np.argsort(-tot)
print(f"{len(tot)} pixels; rays per pixel: mean {tot.mean():.1f}, median {np.median(tot):.0f}, max {tot.max()} ({tot.max() / tot.mean():.1f}x mean); depth limit {depth} rays per sample")
for frac in (0.01, 0.05, 0.2):
    k = max(1, int(len(tot) * frac))
    sel = [per_pixel[i] for i in order[:k]]
    allc = np.concatenate(sel)
    same = np.mean([np.all(c == c[0]) for c in sel])
    print(f"heaviest {frac:.0%} of the pixels ({k}): {np.mean(allc == depth):.0%} of their samples run to the depth limit; all 8 samples equal in {same:.0%} of these pixels; "
          f"mean |count_s - count_(s-1)| = {np.mean([np.abs(np.diff(c)).mean() for c in sel]):.1f}")
allc = np.concatenate(per_pixel)
print(f"all pixels: {np.mean(allc == depth):.0%} of the samples at the limit, {np.mean(allc == 1):.0%} are a single ray (miss); all 8 equal in {np.mean([np.all(c == c[0]) for c in per_pixel]):.0%} of the pixels")

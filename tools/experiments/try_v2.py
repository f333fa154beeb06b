"""GPU box: kernel variant 2 (two paths per lane) vs variant 1 on config M -- identical frame + counters, kernel ms, thresholds."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L
n_tris = int(os.environ.get("SWEEP_TRIS", "10000000"))
tris, mats, texs, cam = synth.atrium_scene(n_target=n_tris, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
hnd = sc.upload(0)
lib = rrt.load()
w, h = 1920, 1080
buf = np.zeros(w * h * 3, dtype=np.float32)
KEYS = ("MIPT_KERNEL", "MIPT_LEAF_NUM", "MIPT_LEAF_DEN", "MIPT_SERVICE_NUM", "MIPT_SERVICE_DEN", "MIPT_BLOCKS_PER_CU")


def run(env, flags=0, reps=3):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ts, st = [], None
    for rep in range(reps):
        o = rrt.make_options(w, h, 8, 64, traversal=1, flags=flags)
        st = L.MiptStats()
        L.check(lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)), "render")
        ts.append(st.kernel_ms)
    return round(min(ts), 2), buf.copy(), st.as_dict()


t1, ref, st1 = run({})
print("v1", t1, flush=True)
_, _, c1 = run({}, flags=L.FLAG_COUNT, reps=1)
_, out, c2 = run({"MIPT_KERNEL": "2"}, flags=L.FLAG_COUNT, reps=1)
print("v2 count build: frame identical", bool(np.array_equal(out.view(np.uint32), ref.view(np.uint32))),
      {k: (c1[k], c2[k]) for k in ("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "pixels", "max_stack")}, flush=True)
d = c2["diag"]
print("v2 diag: iters", d[0], "inner lanes/iter", round(d[1] / max(d[3], 1), 1), "leaf lanes/iter", round(d[2] / max(d[4], 1), 1),
      "inner iters", d[3], "leaf iters", d[4], "services", d[5], "lanes/service", round(d[6] / max(d[5], 1), 1), flush=True)
d = c1["diag"]
print("v1 diag: iters", d[0], "inner lanes/iter", round(d[1] / max(d[3], 1), 1), "leaf lanes/iter", round(d[2] / max(d[4], 1), 1),
      "inner iters", d[3], "leaf iters", d[4], "services", d[5], "lanes/service", round(d[6] / max(d[5], 1), 1), flush=True)
for env in [{}, {"MIPT_LEAF_DEN": "8"}, {"MIPT_LEAF_DEN": "2"}, {"MIPT_LEAF_NUM": "3", "MIPT_LEAF_DEN": "4"}, {"MIPT_SERVICE_DEN": "2"}, {"MIPT_SERVICE_NUM": "3", "MIPT_SERVICE_DEN": "4"},
            {"MIPT_SERVICE_DEN": "2", "MIPT_LEAF_DEN": "2"}, {"MIPT_BLOCKS_PER_CU": "1"}]:
    e = dict(env, MIPT_KERNEL="2")
    t, out, _ = run(e)
    print("v2", env, t, "identical" if np.array_equal(out.view(np.uint32), ref.view(np.uint32)) else "DIFFERENT", flush=True)

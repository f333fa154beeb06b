"""GPU box: A/B of kernel build variants on config M in ONE process (scene generated and BVH built once).

    python tools/ab.py [--tris N] [--reps R] lib1.so lib2.so ...

Each library is loaded privately (RTLD_LOCAL), gets its own scene replica, renders the frame `reps` times in both
traversal modes' default (culled, margin 2^-7) and reports min/median kernel ms plus the CRC of the frame, so a variant that
is not bit-identical to the first library shows at once.  Variants are built with
    make -C rust_ray_tracing_amd/csrc variant NAME=foo EXTRA="-DMIPT_FOO=1"
"""
import argparse, ctypes as C, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--world", type=int, default=1, help="render only rank 0's tile share of this many ranks")
ap.add_argument("--shading", type=int, default=0)
ap.add_argument("--depth", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--seed-mode", type=int, default=0, help="1 = per-sample seeds (rt_compute.wgsl:102)")
ap.add_argument("--traversal", type=int, default=1, help="0 = the CPU backend's un-culled traversal, 1 = culled (margin 2^-7)")
ap.add_argument("--count", action="store_true", help="one extra counting launch: rays / inner steps / tri tests")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()

tris, mats, texs, cam = synth.atrium_scene(n_target=args.tris, tex_size=1024)
sc = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
del tris
sc.build_bvh_device(0)
sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
w, h = args.width, args.height
n_out = int(rrt.load().mipt_packed_pixels(w, h, args.world)) if args.world > 1 else w * h
buf = np.zeros(n_out * 3, dtype=np.float32)
desc = sc.desc()
for spec in args.libs:
    # "lib.so@VAR=VAL,VAR2=VAL2": environment set while this library renders (knobs of a `make TUNING=1` / -DMIPT_TUNING build)
    path, _, envs = spec.partition("@")
    set_env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    for k in [k for k in os.environ if k.startswith("MIPT_")]: del os.environ[k]
    os.environ.update(set_env)
    lib = C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)
    vp = C.c_void_p
    lib.mipt_scene_create.argtypes = [C.POINTER(L.MiptSceneDesc), C.c_int, C.POINTER(vp)]
    lib.mipt_render.argtypes = [vp, vp, C.POINTER(L.MiptOptions), vp, vp, C.POINTER(L.MiptStats)]
    lib.mipt_scene_destroy.argtypes = [vp]
    lib.mipt_last_error.restype = C.c_char_p
    hnd = vp()
    rc = lib.mipt_scene_create(C.byref(desc), 0, C.byref(hnd))
    assert rc == 0, lib.mipt_last_error()
    ts = []
    for rep in range(args.reps):
        o = rrt.make_options(w, h, args.spp, args.depth, seed_mode=args.seed_mode, traversal=args.traversal, flags=L.FLAG_PACKED if args.world > 1 else 0, tile_rank=0,
                             tile_world=args.world, shading=args.shading)
        st = L.MiptStats()
        rc = lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st))
        assert rc == 0, lib.mipt_last_error()
        ts.append(st.kernel_ms)
    extra = ""
    if args.count:
        o = rrt.make_options(w, h, args.spp, args.depth, seed_mode=args.seed_mode, traversal=args.traversal, flags=L.FLAG_COUNT | (L.FLAG_PACKED if args.world > 1 else 0), tile_rank=0,
                             tile_world=args.world, shading=args.shading)
        st = L.MiptStats()
        assert lib.mipt_render(hnd, L.ptr(sc.camera.uniform), C.byref(o), L.ptr(buf), None, C.byref(st)) == 0
        extra = f"  rays {st.rays} inner {st.inner_steps} tri {st.tri_tests}  -> {st.rays / min(ts) / 1e3:.0f} Mray/s, {(st.inner_steps + st.tri_tests) / min(ts) / 1e6:.2f} G lane-steps/s"
        dg = list(st.diag)
        extra += (f"\n    wave iterations {dg[0]} (inner branch in {dg[3]}, leaf branch in {dg[4]}), lanes per iteration inner {dg[1] / max(dg[3], 1):.1f} leaf {dg[2] / max(dg[4], 1):.1f};"
                  f" service passes {dg[5]} with {dg[6] / max(dg[5], 1):.1f} lanes; wave-cycles (100 MHz stamps) in service {dg[7] / max(dg[8], 1):.3f}, tail {dg[10] / max(dg[8], 1):.3f} of {dg[8]}")
    lib.mipt_scene_destroy(hnd)
    print(f"{os.path.basename(spec):40s} min {min(ts):8.3f} ms  median {sorted(ts)[len(ts) // 2]:8.3f} ms  crc {zlib.crc32(buf.tobytes()):08x}{extra}", flush=True)

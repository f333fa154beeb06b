"""CPU: the C-ABI library loads, exports every symbol include/mipt.h declares, and its argument
validation / error conventions work without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="mipt.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mipt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(rrt):
    from rust_ray_tracing_amd import _lib as L
    lib = rrt.load()
    syms = header_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mipt.h but not exported by libmipt.so"
    assert sorted(L.EXPORTS) == syms, "binding list and header disagree"
    assert lib.mipt_abi_version() == 4


def _dynamic_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_exports_are_exactly_the_header(rrt):
    """libmipt.so is built with -fvisibility=hidden + a version script: its dynamic symbol table is include/mipt.h, no more
    (no internal helper, no C++ symbol, no test hook) and no less."""
    from rust_ray_tracing_amd import _lib as L
    assert _dynamic_symbols(L.LIB_PATH) == header_symbols()


def test_multi_rank_test_build_is_not_the_product(rrt):
    """libmipt_multitest.so (RCCL test double, logical ranks on one device) is test infrastructure: the product links the real
    librccl, carries nothing of the double and refuses a device listed twice (GPU: tests/test_gpu_multi.py)."""
    import subprocess
    from rust_ray_tracing_amd import _lib as L
    deps = subprocess.run(["ldd", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in deps
    strings = subprocess.run(["strings", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl double" not in strings and "rccl_double" not in strings
    assert os.path.exists(L.MULTITEST_LIB_PATH)
    tdeps = subprocess.run(["ldd", L.MULTITEST_LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in tdeps and "libamdhip64" in tdeps
    assert _dynamic_symbols(L.MULTITEST_LIB_PATH) == sorted(header_symbols() + ["rccl_double_inject"])
    # the only thing that differs between the two builds of mipt_multi.cpp is keyed on the double's header
    src = open(os.path.join(ROOT, "rust_ray_tracing_amd", "csrc", "mipt_multi.cpp")).read()
    assert src.count("#ifdef MIPT_RCCL_DOUBLE") == 1 and "#include <rccl/rccl.h>" in src


def test_diag_probe_is_a_separate_library(rrt):
    """The device-arithmetic probe (include/mipt_diag.h) lives in libmipt_diag.so; the product exports none of it."""
    import subprocess
    from rust_ray_tracing_amd import _lib as L
    diag = rrt.load_diag()
    syms = header_symbols("mipt_diag.h")
    assert sorted(L.DIAG_EXPORTS) == syms
    for s in syms:
        assert hasattr(diag, s)
    assert _dynamic_symbols(L.DIAG_LIB_PATH) == syms
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "debug_eval" not in out and "mipt_diag" not in out and "mipt_internal" not in out
    # and the product reads no tuning knob from the environment (the knobs live in tools/experiments/ab_layout_macros_and_tuning_knobs.patch)
    strings = subprocess.run(["strings", L.LIB_PATH], capture_output=True, text=True).stdout
    for knob in ("MIPT_LDS_TOP", "MIPT_REVERSE_TILES", "MIPT_SERVICE_NUM", "MIPT_SERVICE_DEN", "MIPT_BLOCKS_PER_CU", "MIPT_LEAF_PERIOD", "MIPT_LEAF_DEN"):
        assert knob not in strings, knob


def test_no_oracle_or_cpu_fallback_linked(rrt):
    """The product must not contain the oracle: no orc_* symbol, no dependency on libpt_oracle."""
    import subprocess
    from rust_ray_tracing_amd import _lib as L
    out = subprocess.run(["nm", "-D", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out
    deps = subprocess.run(["ldd", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps and "libamdhip64" in deps
    for root, _, files in os.walk(os.path.join(ROOT, "rust_ray_tracing_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle/" not in src.replace("oracle/ for tests", "").replace("see oracle/", "") or f == "host.py", f
                assert "import oracle" not in src and "from oracle" not in src, f


def test_struct_sizes_match_header(rrt):
    from rust_ray_tracing_amd import _lib as L
    assert C.sizeof(L.MiptOptions) == 16 * 4
    assert C.sizeof(L.MiptStats) == 8 + 22 * 8          # ABI v3: + touched_lines[2]
    assert C.sizeof(L.MiptSceneDesc) == 4 * 16
    assert C.sizeof(L.MiptTexture) == 16


def test_validation_errors_without_gpu(rrt):
    from rust_ray_tracing_amd import _lib as L
    lib = rrt.load()
    h = C.c_void_p()
    d = L.MiptSceneDesc()                      # all null
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_INVALID_ARG
    assert b"no triangles" in lib.mipt_last_error()
    assert lib.mipt_scene_create(None, 0, C.byref(h)) == L.ERR_INVALID_ARG
    # a structurally broken BVH is rejected before any device work
    from rust_ray_tracing_amd import NODE, TRIANGLE, material_default
    tris = np.zeros(2, dtype=TRIANGLE)
    nodes = np.zeros(2, dtype=NODE)            # even node count
    mats = np.array([material_default()])
    d = L.MiptSceneDesc(L.ptr(tris), 2, L.ptr(nodes), 2, L.ptr(mats), 1, None, 0)
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_BVH
    nodes = np.zeros(3, dtype=NODE)
    nodes[0]["first_tri_or_child"] = 2         # even child index
    d = L.MiptSceneDesc(L.ptr(tris), 2, L.ptr(nodes), 3, L.ptr(mats), 1, None, 0)
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_BVH
    tris["material_id"] = 7
    nodes = np.zeros(1, dtype=NODE)
    nodes[0]["num_tris"] = 2
    d = L.MiptSceneDesc(L.ptr(tris), 2, L.ptr(nodes), 1, L.ptr(mats), 1, None, 0)
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_INVALID_ARG
    assert b"material_id" in lib.mipt_last_error()
    # bounds beyond 2^40 (or non-finite) are outside the exact-division fast path's guard: rejected at upload
    tris["material_id"] = 0
    nodes = np.zeros(1, dtype=NODE)
    nodes[0]["num_tris"] = 2
    nodes[0]["bounds_max"] = (1e13, 1.0, 1.0)
    d = L.MiptSceneDesc(L.ptr(tris), 2, L.ptr(nodes), 1, L.ptr(mats), 1, None, 0)
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_SCENE_LIMIT
    nodes[0]["bounds_max"] = (np.inf, 1.0, 1.0)
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_SCENE_LIMIT
    # render entry points reject null scenes / bad options with the reference's messages (renderer.rs:15-26)
    opt = rrt.make_options(0, 10, 1, 1)
    assert lib.mipt_render(None, None, C.byref(opt), None, None, None) == L.ERR_INVALID_ARG
    assert lib.mipt_bvh_build(None, 0, None, 0, None, 0) == L.ERR_INVALID_ARG
    assert lib.mipt_packed_pixels(1920, 1080, 8) == ((240 * 135 + 7) // 8) * 64
    assert lib.mipt_packed_pixels(61, 37, 3) == ((8 * 5 + 2) // 3) * 64


def test_fails_loudly_without_a_device(rrt):
    """On a box without a GPU a well-formed scene must come back as MIPT_ERR_HIP -- never a CPU render."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU-only CI box")
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.cornell_box()
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    h = C.c_void_p()
    d = sc.desc()
    rc = rrt.load().mipt_scene_create(C.byref(d), 0, C.byref(h))
    assert rc == L.ERR_HIP, rc
    r = rrt.Renderer.new(rrt.RendererOptions(samples=1, max_ray_depth=1, output_image_dimensions=(8, 8), output_image_path="x.png"))
    import pytest
    with pytest.raises(rrt.MiptError):
        r.render_buffers(sc)


def test_scene_create_survives_random_bvh_garbage(rrt):
    """Fuzz of the upload-time validation (no GPU needed: it runs before any device work).  Random node arrays -- child indices,
    triangle ranges and leaf sizes drawn at random, and mutations of a valid tree -- must come back with a status code quickly:
    MIPT_ERR_BVH / INVALID_ARG / SCENE_LIMIT for malformed input, MIPT_ERR_HIP (CPU box) or MIPT_OK (GPU box) for the few that happen
    to be well-formed.  Never a crash, a hang or an unbounded allocation (a 55-node DAG once took 4 s and 1.2 GB)."""
    import time
    from rust_ray_tracing_amd import NODE, TRIANGLE, material_default
    from rust_ray_tracing_amd import _lib as L
    from rust_ray_tracing_amd import synth
    lib = rrt.load()
    rng = np.random.default_rng(1234)
    mats = np.array([material_default()])
    base = rrt.Scene.from_arrays(synth.make_scene("atrium", n_target=2000, tex_size=8)[0], [material_default()])
    seen = set()
    t0 = time.time()
    for trial in range(400):
        if trial % 2 == 0:                                   # pure garbage
            n_nodes = int(rng.integers(1, 60)) | 1
            n_tris = int(rng.integers(1, 40))
            nodes = np.zeros(n_nodes, dtype=NODE)
            nodes["bounds_max"] = 1.0
            nodes["first_tri_or_child"] = rng.integers(0, 2 * n_nodes, n_nodes)
            nodes["num_tris"] = np.where(rng.random(n_nodes) < 0.5, 0, rng.integers(0, 5, n_nodes))
            tris = np.zeros(n_tris, dtype=TRIANGLE)
        else:                                                # a valid tree with a few fields damaged
            nodes = base.bvh_nodes.copy()
            tris = base.tris
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(0, len(nodes)))
                f = ("first_tri_or_child", "num_tris")[int(rng.integers(0, 2))]
                nodes[i][f] = int(rng.integers(0, 2 * len(nodes)))
        h = C.c_void_p()
        d = L.MiptSceneDesc(L.ptr(tris), len(tris), L.ptr(nodes), len(nodes), L.ptr(mats), 1, None, 0)
        rc = lib.mipt_scene_create(C.byref(d), 0, C.byref(h))
        assert rc in (L.OK, L.ERR_HIP, L.ERR_BVH, L.ERR_INVALID_ARG, L.ERR_SCENE_LIMIT), rc
        seen.add(rc)
        if rc == L.OK:
            lib.mipt_scene_destroy(h)
        else:
            assert h.value is None and len(lib.mipt_last_error()) > 0
    assert L.ERR_BVH in seen and time.time() - t0 < 20.0

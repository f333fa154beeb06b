"""The C++ host-side mirror of Renderer / Scene (include/mipt_host.hpp) over the C ABI."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "test_host")
    lib_dir = os.path.join(ROOT, "rust_ray_tracing_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_host.cpp"),
                           "-o", exe, "-L", lib_dir, "-l:libmipt.so", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_mirror_cpu(built, tmp_path):
    from rust_ray_tracing_amd import synth
    obj = synth.write_cornell_obj(str(tmp_path))
    out = subprocess.run([_build(tmp_path), "cpu", obj], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Width and height must be greater than 0" in out.stderr and "Could not find scene" in out.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_renders_config1(built, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "cornell_256x256_4spp_rgba.npz"))
    from rust_ray_tracing_amd import synth
    obj = synth.write_cornell_obj(str(tmp_path))
    want = tmp_path / "want.rgba"
    want.write_bytes(g["rgba"].tobytes())
    png = tmp_path / "out.png"
    out = subprocess.run([_build(tmp_path), "gpu", obj, str(want), str(png)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    # Renderer::render also saved the frame (renderer.rs:66-83); the file holds the same pixels
    import rust_ray_tracing_amd as rrt
    t = rrt.Texture.load(str(png))
    assert t is not None and np.array_equal(t.pixel_data[::-1], g["rgba"].reshape(256, 256, 4))


def test_host_code_under_asan_ubsan(built, tmp_path):
    """GPU sanitizers are unavailable: the host-side C++ that parses untrusted files (OBJ/MTL, PNG, JPEG, TGA, BMP) and the BVH builder
    run under AddressSanitizer + UBSan on the CPU, with truncated / bit-flipped inputs."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_obj_loader import _png
    from rust_ray_tracing_amd import synth
    synth.write_cornell_obj(str(tmp_path))
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (24, 31, 4), dtype=np.uint8)
    _png(str(tmp_path / "tex.png"), img, 6, filters=[0, 1, 2, 3, 4], chunk=211)
    (tmp_path / "t.mtl").write_text("newmtl a\nmap_Kd tex.png\n")
    (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    (tmp_path / "neg.obj").write_text("v 0 0 0\nf -1 -2 -3\n")
    src = os.path.join(ROOT, "rust_ray_tracing_amd", "csrc")
    exe = str(tmp_path / "sanitize_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(src, "bvh_build.cpp"), os.path.join(src, "obj_loader.cpp"), os.path.join(src, "png_decode.cpp"),
                           os.path.join(src, "jpeg_decode.cpp"), os.path.join(src, "tga_bmp_decode.cpp"), os.path.join(ROOT, "tests", "cpp", "layout_order.cpp"),
                           os.path.join(ROOT, "tests", "cpp", "sanitize_host.cpp"), "-o", exe, "-lpthread"])
    jpegs = []
    try:
        from PIL import Image
        from test_obj_loader import _jpeg_test_image
        for i, kw in enumerate([dict(subsampling=2), dict(subsampling=1, progressive=True), dict(subsampling=0, restart_marker_blocks=2)]):
            p = str(tmp_path / f"j{i}.jpg")
            try:
                Image.fromarray(_jpeg_test_image(29, 43)).save(p, "JPEG", quality=85, **kw)
            except (TypeError, ValueError):
                continue
            jpegs.append(p)
        rgba = np.random.default_rng(3).integers(0, 256, (19, 23, 4), dtype=np.uint8)
        rgba[:, :9] = rgba[:1, :1]
        for name, kw in (("a.tga", dict(compression="tga_rle")), ("b.tga", dict()), ("c.bmp", dict())):
            Image.fromarray(rgba, "RGBA").save(str(tmp_path / name), **kw)
            jpegs.append(str(tmp_path / name))
        Image.fromarray(rgba[..., :3], "RGB").convert("P").save(str(tmp_path / "d.bmp"))
        jpegs.append(str(tmp_path / "d.bmp"))
    except ImportError:
        pass
    out = subprocess.run([exe, str(tmp_path), str(tmp_path / "tex.png")] + jpegs, capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr[-3000:]
    assert "sanitize_host ok" in out.stdout
    # and under ThreadSanitizer: mipt_bvh_build splits sub-trees over std::threads and stitches them (bvh_build.cpp)
    exe_t = str(tmp_path / "sanitize_host_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(src, "bvh_build.cpp"), os.path.join(src, "obj_loader.cpp"), os.path.join(src, "png_decode.cpp"),
                           os.path.join(src, "jpeg_decode.cpp"), os.path.join(src, "tga_bmp_decode.cpp"), os.path.join(ROOT, "tests", "cpp", "layout_order.cpp"),
                           os.path.join(ROOT, "tests", "cpp", "sanitize_host.cpp"), "-o", exe_t, "-lpthread"])
    out = subprocess.run([exe_t, str(tmp_path), str(tmp_path / "tex.png")] + jpegs, capture_output=True, text=True,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "WARNING: ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]


def test_oracle_under_asan_ubsan(built, tmp_path):
    """The oracle itself (test infrastructure) under ASan/UBSan: a small render through a standalone driver."""
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include "pt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void) {
    enum { N = 200 };
    OrcTriangle *t = calloc(N, sizeof *t);
    unsigned s = 12345;
    for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) {
        float *p = &t[i].vertices[k].position.x;
        for (int c = 0; c < 3; c++) { s = s * 1664525u + 1013904223u; p[c] = (float)(s >> 8) / 16777216.0f * 4.0f - 2.0f + (c == 0 ? -4.0f : 0.0f); }
        t[i].vertices[k].normal.x = 1.0f;
    }
    OrcNode *nodes = calloc(2 * N, sizeof *nodes);
    uint32_t nn = 0;
    if (orc_bvh_build(t, N, nodes, 2 * N, &nn)) return 1;
    OrcMaterial m; memset(&m, 0, sizeof m); m.base_color.x = m.base_color.y = m.base_color.z = 0.8f;
    m.base_color_tex_id = m.emission_tex_id = 0xffffffffu;
    OrcCamera cam; float pos[3] = {3, 0, 0}; orc_camera_from_pose(pos, 0.0f, 0.0f, &cam);
    OrcOptions o; memset(&o, 0, sizeof o); o.width = 48; o.height = 32; o.samples = 3; o.max_ray_depth = 16; o.threads = 3;
    float *hdr = calloc(48 * 32 * 3, 4); unsigned char *rgba = calloc(48 * 32 * 4, 1);
    OrcStats st;
    for (int cull = 0; cull < 2; cull++) { o.cull = cull; o.cull_margin = 0.0078125f; if (orc_render(t, N, nodes, nn, &m, 1, NULL, 0, &cam, &o, hdr, rgba, &st)) return 2; }
    printf("oracle asan ok: %llu rays\n", (unsigned long long)st.rays);
    free(t); free(nodes); free(hdr); free(rgba);
    return 0;
}
''')
    exe = str(tmp_path / "drv")
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                           "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "oracle"), str(drv), os.path.join(ROOT, "oracle", "pt_oracle.c"),
                           "-o", exe, "-lm", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr[-3000:]
    assert "oracle asan ok" in out.stdout
    # the same driver under ThreadSanitizer: the pixel loop shares the scene read-only across its worker threads (the reference's
    # rayon workers share &Scene, cpu.rs:13,24) and merges per-thread counters under a mutex -- TSan must see no race
    exe_t = str(tmp_path / "drv_tsan")
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "oracle"),
                           str(drv), os.path.join(ROOT, "oracle", "pt_oracle.c"), "-o", exe_t, "-lm", "-lpthread"])
    out = subprocess.run([exe_t], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "WARNING: ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]

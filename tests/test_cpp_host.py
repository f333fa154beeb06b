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
    out = subprocess.run([_build(tmp_path), "gpu", obj, str(want)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr

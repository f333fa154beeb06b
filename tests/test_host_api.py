"""CPU: the host-side mirror of Renderer / RendererOptions / Scene (renderer.rs, scene.rs) -- validation and
error behaviour of the reference, no compute."""
import numpy as np


def test_renderer_new_validation(rrt, capsys):  # renderer.rs:14-34
    RO, R, B = rrt.RendererOptions, rrt.Renderer, rrt.RendererBackend
    assert R.new(RO(output_image_dimensions=(0, 10), output_image_path="a.png")) is None
    assert "Width and height must be greater than 0" in capsys.readouterr().err
    assert R.new(RO(max_ray_depth=0, output_image_path="a.png")) is None
    assert "Max ray depth must be greater than 0" in capsys.readouterr().err
    assert R.new(RO(samples=0, output_image_path="a.png")) is None
    assert "Sample count must be greater than 0" in capsys.readouterr().err
    assert R.new(RO(output_image_path=None, is_realtime=False)) is None
    assert "Output image path must be Some" in capsys.readouterr().err
    assert R.new(RO(backend=B.MI355X, is_realtime=True)) is None
    assert "Only the GPU backend is supported for realtime mode" in capsys.readouterr().err
    r = R.new(RO(samples=4, max_ray_depth=6, output_image_dimensions=(16, 8), output_image_path="a.png"))
    assert r is not None and r.options.backend == B.MI355X


def test_defaults_match_reference(rrt):
    o = rrt.RendererOptions()
    assert (o.samples, o.max_ray_depth, o.output_image_dimensions) == (1, 6, (1920, 1080))     # renderer.rs:106-116
    m = rrt.material_default()                                                                 # scene.rs:148-167
    assert tuple(m["base_color"]) == (np.float32(0.8),) * 3 and m["ior"] == np.float32(1.45)
    assert m["roughness"] == 1 and m["transparency"] == 1 and m["metallic"] == 0 and m["transmission"] == 0
    assert all(m[k] == 0xFFFFFFFF for k in m.dtype.names if k.endswith("tex_id"))


def test_camera_matches_oracle(rrt, orc):
    for pos, pitch, yaw in [((0, 0, 0), 0, 0), ((-11.204422, 2.1092458, -0.12164927), 1.5998944, -179.10223), ((3, 1, -2), -35.5, 77.25)]:
        cam = rrt.Camera(position=pos, pitch=pitch, yaw=yaw)
        cam.update_view()
        assert cam.uniform.tobytes() == orc.camera_from_pose(pos, pitch, yaw).tobytes()


def test_other_backends_are_not_silently_substituted(rrt):
    import pytest
    from rust_ray_tracing_amd import synth
    tris, mats, texs, cam = synth.cornell_box()
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    r = rrt.Renderer.new(rrt.RendererOptions(output_image_path="x.png", backend=rrt.RendererBackend.CPU))
    with pytest.raises(NotImplementedError):
        r.render_buffers(sc)

"""Regenerates the golden fixtures from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be built here (no Rust toolchain), so these are
emitted by the oracle itself ("parity unpinned" by the reference; pinned by the Appendix-B known
answers in tests/test_oracle_kat.py).  Each fixture is DATA: the inputs are regenerated from the
seeded generators, the .npz holds expected outputs only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CASES = {
    # name: (scene kind, generator kwargs, width, height, spp, depth, seed_mode)
    "cornell_64x64_4spp": ("cornell", {}, 64, 64, 4, 16, 0),
    "cornell_256x256_4spp_rgba": ("cornell", {}, 256, 256, 4, 64, 0),          # BASELINE.json configs[0]
    "helmet4k_96x54_4spp": ("helmet", dict(n_target=4000, tex_size=64), 96, 54, 4, 12, 0),
    "atrium20k_64x36_2spp": ("atrium", dict(n_target=20000, tex_size=32), 64, 36, 2, 16, 0),
    "atrium20k_64x36_4spp_persample": ("atrium", dict(n_target=20000, tex_size=32), 64, 36, 4, 8, 1),
}


def render_case(name):
    from oracle import orc
    from rust_ray_tracing_amd import synth
    kind, kw, w, h, spp, depth, seed_mode = CASES[name]
    tris, mats, texs, cam = synth.make_scene(kind, **kw)
    t, nodes = orc.bvh_build(tris)
    mats_arr = np.array(list(mats.values()))
    camera = orc.camera_from_pose(*cam)
    hdr, rgba, st = orc.render(t, nodes, mats_arr, texs, camera, w, h, spp, depth, seed_mode=seed_mode)
    return hdr, rgba, st


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        hdr, rgba, st = render_case(name)
        if name.endswith("_rgba"):
            np.savez_compressed(os.path.join(here, name + ".npz"), rgba=rgba, rays=np.uint64(st["rays"]),
                                hdr_checksum=np.uint64(int(hdr.view(np.uint32).astype(np.uint64).sum())))
        else:
            np.savez_compressed(os.path.join(here, name + ".npz"), hdr=hdr, rgba=rgba, rays=np.uint64(st["rays"]),
                                inner_steps=np.uint64(st["inner_steps"]), tri_tests=np.uint64(st["tri_tests"]))
        print(name, hdr.shape, st["rays"])

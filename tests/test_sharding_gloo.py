"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- tile-interleaved pixel shards written
rank-packed, ONE all-gather, de-interleave -- must reproduce the single-rank frame bit for bit, and the
sample-sharded sum-reduce (config 5) must equal the rank-ordered sum of the per-rank partials.
Without a GPU the pixels are produced by the CPU oracle; the -m gpu variant makes every rank render ITS OWN packed tile slice /
sample-range partial sum with the HIP kernel through the C ABI (both ranks share the box's one GPU) and gathers / reduces
those over gloo, checking the result against the oracle's whole frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, q, use_kernel=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from rust_ray_tracing_amd import sharding, synth
    tris, mats, texs, cam = synth.make_scene("atrium", n_target=20000, tex_size=32)
    t, nodes = orc.bvh_build(tris)
    m = np.array(list(mats.values()))
    camera = orc.camera_from_pose(*cam)
    # --- config 4: tile shards ---
    full, _, _ = orc.render(t, nodes, m, texs, camera, w, h, 2, 8, threads=2)
    pix = sharding.slot_pixels(w, h, rank, world)
    if use_kernel:                                                # this rank's slice from the HIP kernel, rank-packed
        import ctypes as C
        import rust_ray_tracing_amd as rrt
        from rust_ray_tracing_amd import _lib as L
        sc = rrt.Scene.from_arrays(tris, mats, texs)
        sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))

        def kernel(opt, n_floats):
            out = np.zeros(n_floats, dtype=np.float32)
            L.check(rrt.load().mipt_render(sc.upload(0), L.ptr(sc.camera.uniform), C.byref(opt), L.ptr(out), None, None), "mipt_render")
            return out
        local = kernel(rrt.make_options(w, h, 2, 8, flags=L.FLAG_PACKED, tile_rank=rank, tile_world=world), len(pix) * 3).reshape(-1, 3)
    else:
        local = np.zeros((len(pix), 3), dtype=np.float32)
        local[pix >= 0] = full.reshape(-1, 3)[pix[pix >= 0]]      # what this rank's kernel writes, packed
    gathered = torch.empty(world * local.size, dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, torch.from_numpy(local.reshape(-1)))
    frame = sharding.unpack(gathered.numpy().reshape(world, -1, 3), w, h, world)
    ok_tiles = np.array_equal(frame.view(np.uint32), full.reshape(-1, 3).view(np.uint32))
    # --- config 5: sample shards, per-sample seeds, sum-reduce ---
    spp = 5
    s0, n = sharding.sample_ranges(spp, world)[rank]
    if use_kernel:
        part = kernel(rrt.make_options(w, h, n, 8, seed_mode=L.SEED_PER_SAMPLE, flags=L.FLAG_SUM, sample_begin=s0), w * h * 3).reshape(h, w, 3)
    else:
        part, _, _ = orc.render(t, nodes, m, texs, camera, w, h, n, 8, seed_mode=1, sample_begin=s0, sum_only=1, threads=2)
    red = torch.from_numpy(part.copy())
    dist.all_reduce(red)                                          # world 2: a + b is order-independent
    parts = [torch.empty_like(red) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(part.copy()))
    ordered = parts[0].clone()
    for p in parts[1:]:
        ordered += p
    ok_samples = torch.equal(red, ordered)
    whole, _, _ = orc.render(t, nodes, m, texs, camera, w, h, spp, 8, seed_mode=1, sum_only=1, threads=2)
    close = np.allclose(red.numpy(), whole, rtol=1e-6, atol=1e-6)  # sequential vs rank-tree f32 summation
    if rank == 0:
        q.put((ok_tiles, ok_samples, close))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(w, h, use_kernel):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + w + (7 if use_kernel else 0)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q, use_kernel)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == (True, True, True)


@pytest.mark.parametrize("w,h", [(64, 40), (61, 37)])
def test_two_rank_tile_gather_and_sample_reduce(built, w, h):
    _run_two_ranks(w, h, use_kernel=False)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(64, 40), (61, 37)])
def test_two_rank_gather_and_reduce_of_the_kernels_own_shards(built, w, h):
    _run_two_ranks(w, h, use_kernel=True)


def test_slot_layout_is_a_partition():
    sys.path.insert(0, ROOT)
    from rust_ray_tracing_amd import sharding
    for (w, h, world) in [(1920, 1080, 8), (61, 37, 3), (8, 8, 4), (100, 9, 8)]:
        seen = np.zeros(w * h, dtype=np.int32)
        for r in range(world):
            pix = sharding.slot_pixels(w, h, r, world)
            assert len(pix) == sharding.packed_pixels(w, h, world)
            np.add.at(seen, pix[pix >= 0], 1)
        assert np.all(seen == 1)
    assert sharding.sample_ranges(1024, 8) == [(1 + 128 * r, 128) for r in range(8)]
    assert sharding.sample_ranges(5, 2) == [(1, 3), (4, 2)]

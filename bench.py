#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json: Mray/s at 1920x1080, 8 spp on a Sponza-scale
(~10 M-triangle) BVH, 1/2/4/8 GPUs, with the HBM roofline of the trace kernel and the CPU path
timed beside it.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full frame of the hot path: every rank traces its share of the 8x8 image tiles
(scene replicated per GPU, inputs resident in HBM) and, for N > 1, the frame is assembled with one
RCCL all-gather of the rank-packed tile slices plus a de-interleave kernel.  A "ray" is one
traverse_bvh invocation (reference src/renderer/backend/cpu/ray.rs:150); ray counts come from the
kernel's counting build on the same inputs (deterministic), run outside the timed region.
Data is synthetic (no assets ship with the reference): the seeded "atrium" stand-in of
rust_ray_tracing_amd/synth.py.  max_ray_depth = 64 is the reference's shipped value (src/main.rs:20).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rust_ray_tracing_amd as rrt  # noqa: E402
from rust_ray_tracing_amd import _lib as L  # noqa: E402
from rust_ray_tracing_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes(st: dict, n_samples: int) -> int:
    """BASELINE.md section 3 / SURVEY.md 8(d): reference layouts (Node 32 B, Triangle 112 B, Material 80 B)."""
    return (32 * st["rays"] + 64 * st["inner_steps"] + 112 * st["tri_tests"] + 80 * st["hits"]
            + 4 * st["texel_fetches"] + 12 * n_samples)


def pmc_traffic_bytes(n_tris_requested, w, h, spp, depth, traversal):
    """HBM-side bytes per launch of the trace kernel from the committed rocprofv3 PMC summary (a separate --pmc run of
    this same command, tools/pmc.sh): FETCH_SIZE [KiB] x 1024 x 2 (gfx950 counts a 128-B line fill as 64 B --
    MI355X_MICROARCH.md, HBM; confirmed by TCC_MISS_sum x 128 B) + WRITE_SIZE [KiB] x 1024.  Only valid for the
    configuration it was measured on (the default config M); otherwise None."""
    if (n_tris_requested, w, h, spp, depth, traversal) != (10_000_000, 1920, 1080, 8, 64, "culled"):
        return None
    path = os.path.join(ROOT, "profiles", "r1_pmc_summary_final.csv")
    try:
        vals = {}
        for line in open(path).read().splitlines()[1:]:
            parts = line.split(",")
            vals[parts[1]] = float(parts[2])
        return int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    except Exception:
        return None


def baseline_metric():
    """The metric string of BASELINE.json (the file travels with the repo); the literal is its value at the time of writing."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Mray/s at 1920x1080, 8 spp, Sponza BVH; 1/2/4/8-GPU scaling + % HBM roofline"


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--traversal", choices=["culled", "reference"], default="culled")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 code path on fewer GPUs than ranks (collectives staged through host memory)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X backend has no CPU fallback")
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    on_host = args.dist_backend == "gloo"

    def all_reduce_(t, op=dist.ReduceOp.SUM):
        if on_host:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)
    lib = rrt.load()

    # ---- scene (identical on every rank: seeded generator + deterministic builder) ----
    t0 = time.time()
    tris, mats, texs, cam = synth.atrium_scene(n_target=args.tris, tex_size=args.tex_size)
    scene = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
    del tris
    t1 = time.time()
    scene.build_bvh()
    t2 = time.time()
    scene.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    handle = scene.upload(local_rank)
    t3 = time.time()
    log(rank, f"scene: {len(scene.tris)} tris, {len(scene.bvh_nodes)} nodes; gen {t1 - t0:.1f}s, bvh {t2 - t1:.1f}s, upload {t3 - t2:.1f}s")

    w, h, spp, depth = args.width, args.height, args.spp, args.depth
    trav = L.TRAVERSAL_CULLED if args.traversal == "culled" else L.TRAVERSAL_REFERENCE
    n_pix = w * h
    packed = world > 1
    flags = L.FLAG_PACKED if packed else 0
    n_slot = int(lib.mipt_packed_pixels(w, h, world)) if packed else n_pix
    d_local = torch.empty(n_slot * 3, dtype=torch.float32, device=dev)
    d_all = torch.empty(world * n_slot * 3, dtype=torch.float32, device=dev) if packed else None
    d_frame = torch.empty(n_pix * 3, dtype=torch.float32, device=dev) if packed else d_local
    cam_ptr = L.ptr(scene.camera.uniform)
    stream = torch.cuda.current_stream(dev)

    def render(extra_flags=0, traversal=trav, out=None):
        opt = rrt.make_options(w, h, spp, depth, L.SEED_PIXEL_STREAM, traversal, flags | extra_flags, rank, world,
                               cull_margin=L.CULL_MARGIN_SAFE)
        st = L.MiptStats()
        buf = d_local if out is None else out
        L.check(lib.mipt_render_device(handle, cam_ptr, C.byref(opt), C.c_void_p(buf.data_ptr()), None,
                                       C.c_void_p(stream.cuda_stream), C.byref(st)), "mipt_render_device")
        return st.as_dict()

    def step():
        st = render()
        if packed and on_host:
            c_all = torch.empty(d_all.shape, dtype=d_all.dtype)
            dist.all_gather_into_tensor(c_all, d_local.cpu())
            d_all.copy_(c_all)
        elif packed:
            dist.all_gather_into_tensor(d_all, d_local)
        if packed:
            L.check(lib.mipt_unpack_tiles(C.c_void_p(d_all.data_ptr()), w, h, world, C.c_void_p(d_frame.data_ptr()),
                                          C.c_void_p(stream.cuda_stream)), "mipt_unpack_tiles")
        return st

    # ---- counting build, outside the timed region (deterministic: same counts as the timed launches) ----
    cst = render(extra_flags=L.FLAG_COUNT)
    counts = torch.tensor([cst[k] for k in ("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "pixels")],
                          dtype=torch.int64, device=dev)
    local_counts = {k: int(v) for k, v in zip(("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "pixels"), counts.tolist())}
    if world > 1:
        all_reduce_(counts)
    tot = {k: int(v) for k, v in zip(("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "pixels"), counts.tolist())}
    assert tot["pixels"] == n_pix, (tot["pixels"], n_pix)
    log(rank, f"counts: {tot}")

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(step()["kernel_ms"])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce_(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    value = tot["rays"] * args.steps / elapsed / 1e6
    # roofline of the dominant kernel (pt_trace_kernel), this rank's launches
    avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
    alg_bytes = algorithmic_bytes(local_counts, local_counts["pixels"] * spp)
    achieved = alg_bytes / avg_kernel_s / 1e9
    result = {
        "metric": baseline_metric(),
        "value": round(value, 3), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"atrium stand-in for Intel Sponza + curtains: {len(scene.tris)} tris, {len(scene.bvh_nodes)} BVH nodes, "
                               f"{w}x{h}, {spp} spp, max_ray_depth {depth}, traversal {args.traversal}" + (f" (margin {L.CULL_MARGIN_SAFE})" if args.traversal == "culled" else "") + ", pixel-stream seeds",
                   "sharding": f"8x8 image tiles round-robin over {world} rank(s)" + (", one RCCL all-gather of packed tile slices per frame" if packed else ""),
                   "rays_per_frame": tot["rays"], "paths_per_frame": n_pix * spp},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": pmc_traffic_bytes(args.tris, w, h, spp, depth, args.traversal) if world == 1 else None,
                     "traffic_note": "bytes per launch, rocprofv3 PMC (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), separate run: profiles/r1_pmc_summary_final.csv",
                     "kernel": "pt_trace_kernel", "kernel_ms": round(avg_kernel_s * 1e3, 3),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "bytes_per_ray": round(alg_bytes / max(local_counts["rays"], 1), 1),
                     "mray_s_kernel": round(local_counts["rays"] / avg_kernel_s / 1e6, 2)},
    }

    # ---- parity evidence at the bench size (outside the timed region) ----
    if rank == 0 and not args.no_parity:
        par = {}
        frame = d_frame.clone()
        if world > 1:
            # the all-gathered + de-interleaved frame must equal a frame rendered by this rank alone, bit for bit
            solo = torch.empty(n_pix * 3, dtype=torch.float32, device=dev)
            opt1 = rrt.make_options(w, h, spp, depth, L.SEED_PIXEL_STREAM, trav, 0, 0, 1, cull_margin=L.CULL_MARGIN_SAFE)
            L.check(lib.mipt_render_device(handle, cam_ptr, C.byref(opt1), C.c_void_p(solo.data_ptr()), None,
                                           C.c_void_p(stream.cuda_stream), None), "mipt_render_device")
            torch.cuda.synchronize(dev)
            par["gathered_frame_equals_single_gpu_frame"] = bool(torch.equal(frame.view(torch.int32), solo.view(torch.int32)))
            del solo
        if world == 1 and args.traversal == "culled":
            ref_buf = torch.empty_like(d_local)
            render(traversal=L.TRAVERSAL_REFERENCE, out=ref_buf)
            torch.cuda.synchronize(dev)
            par["culled_equals_reference_traversal"] = bool(torch.equal(frame.view(torch.int32), ref_buf.view(torch.int32)))
            del ref_buf
        result["parity"] = par

    # ---- CPU baseline: the oracle (C restatement of the rayon backend) on a strided pixel sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc
        mats_arr = scene.materials_array()

        def cpu(stride):
            return orc.render(scene.tris, scene.bvh_nodes, mats_arr, scene.textures, scene.camera.uniform, w, h, spp, depth,
                              cull=0, pix_stride=stride, want_rgba8=False)
        probe_stride = 8191
        _, _, ps = cpu(probe_stride)
        rate = ps["rays"] / max(ps["seconds"], 1e-6)
        want_rays = rate * args.cpu_seconds
        stride = max(1, int(tot["rays"] / max(want_rays, 1)))
        stride = stride | 1                       # odd stride: samples every image column
        hdr_cpu, _, cs = cpu(stride)
        result["cpu_baseline"] = {"value": round(cs["rays"] / cs["seconds"] / 1e6, 4), "unit": "Mray/s", "cores": cs["threads_used"],
                                  "kind": "port",
                                  "sample": f"every {stride}th pixel of the same frame ({cs['rays']} rays, {cs['seconds']:.1f} s), "
                                            "C restatement of the reference's rayon backend (no t-max cull), all samples and bounces"}
        if not args.no_parity:
            got = d_frame.cpu().numpy().reshape(h, w, 3)
            idx = np.arange(0, n_pix, stride)
            a = got.reshape(-1, 3)[idx].view(np.uint32)
            b = hdr_cpu.reshape(-1, 3)[idx].view(np.uint32)
            result["parity"]["oracle_bit_exact_on_sample"] = bool(np.array_equal(a, b))
            result["parity"]["sample_pixels"] = int(len(idx))
            result["parity"]["rmse"] = float(np.sqrt(np.mean((got.reshape(-1, 3)[idx].astype(np.float64) - hdr_cpu.reshape(-1, 3)[idx]) ** 2)))

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json: Mray/s at 1920x1080, 8 spp on a Sponza-scale
(~10 M-triangle) BVH, 1/2/4/8 GPUs, with the roofline of the trace kernel and the CPU path timed beside it.

    python bench.py --gpus N --steps K --warmup W [--mode tiles|samples]
        one process drives all N GPUs through the library's own multi-GPU entry (mipt_render_multi_device: scene replicas, a host
        thread and an RCCL communicator per device, ONE ncclGather / ncclReduce per frame) -- the reference's single-process host model
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        one process per GPU, collectives through torch.distributed (backend nccl = RCCL)

A "step" is one full frame of the hot path (scene replicated per GPU, inputs resident in HBM):
  --mode tiles (default; BASELINE config 4's split and the metric's configuration): every rank traces its share of the
      8x8 image tiles with the CPU backend's per-pixel seeds (reference src/renderer/backend/cpu.rs:28-29); for N > 1 the
      frame is assembled with ONE RCCL all-gather of the rank-packed tile slices plus a de-interleave kernel.
  --mode samples (BASELINE config 5's split): every rank traces ALL pixels for its share of the samples with the
      per-sample seeds of rt_compute.wgsl:102 into un-normalised sums; ONE RCCL sum-reduce to rank 0, then / spp.
With N > 1 the line also carries the other mode's rate ("other_mode"): the tile split is floored by the reference's
sequential per-pixel RNG chain, the sample split is not.
A "ray" is one traverse_bvh invocation (reference src/renderer/backend/cpu/ray.rs:150); ray counts come from the kernel's
counting build on the same inputs (deterministic), run outside the timed region.  Data is synthetic (no assets ship with
the reference): the seeded "atrium" stand-in of rust_ray_tracing_amd/synth.py.  max_ray_depth = 64 is the reference's
shipped value (src/main.rs:20).
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
from rust_ray_tracing_amd import sharding, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
COUNT_KEYS = ("rays", "inner_steps", "tri_tests", "hits", "texel_fetches", "pixels")


def algorithmic_bytes(st: dict, n_samples: int) -> int:
    """BASELINE.md section 3 / SURVEY.md 8(d): bytes in the REFERENCE layouts (Node 32 B, Triangle 112 B, Material 80 B)."""
    return (32 * st["rays"] + 64 * st["inner_steps"] + 112 * st["tri_tests"] + 80 * st["hits"]
            + 4 * st["texel_fetches"] + 12 * n_samples)


def device_bytes(st: dict, n_pixels: int) -> int:
    """Bytes the kernel's loads actually request in the DEVICE layout: one 64-B record per inner step,
    64 B per triangle test (one 64-B-strided record), per hit 64 B attributes + 64 B material, 4 B per
    texel, 12 B per pixel written."""
    return (64 * st["inner_steps"] + 64 * st["tri_tests"] + 128 * st["hits"] + 4 * st["texel_fetches"] + 12 * n_pixels)


from rust_ray_tracing_amd.provenance import kernel_source_sha  # noqa: E402


def pmc_traffic(n_tris_requested, w, h, spp, depth, traversal, mode):
    """Memory-side bytes per launch of the trace kernel from the committed rocprofv3 PMC summary (a separate --pmc run of this
    same command, tools/pmc.sh): TCC_EA0_RDREQ_sum x 128 B + WRITE_SIZE.  Every read request is a 128-B line fill -- also for
    this kernel's per-lane 64-B gathers (calibrated: profiles/r2_fetch_calibration.csv; FETCH_SIZE tallies them at 64 B).
    Quoted only for the configuration AND kernel source it was measured on; otherwise (None, reason)."""
    if (n_tris_requested, w, h, spp, depth, traversal, mode) != (10_000_000, 1920, 1080, 8, 64, "culled", "tiles"):
        return None, "not the profiled configuration"
    why = "no PMC summary under profiles/"
    for name in ("r4_pmc_summary.csv", "r3_pmc_summary.csv"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            vals, meta = {}, {"file": f"profiles/{name}"}
            for line in open(path).read().splitlines():
                if line.startswith("#"):
                    for kv in line[1:].split():
                        if "=" in kv:
                            k, v = kv.split("=", 1)
                            meta[k] = v
                    continue
                parts = line.split(",")
                if len(parts) >= 3 and parts[0] != "kernel":
                    vals[parts[1]] = float(parts[2])
            if meta.get("kernel_sha") != kernel_source_sha():
                why = f"profiles/{name} was measured on kernel source {meta.get('kernel_sha')}, this is {kernel_source_sha()}"
                continue
            return int(vals["TCC_EA0_RDREQ_sum"] * 128 + vals["WRITE_SIZE"] * 1024), meta
        except Exception as e:  # noqa: BLE001
            why = f"profiles/{name}: {e}"
    return None, why


def step_roof():
    """G lane-steps/s a traversal-shaped chain reaches on this chip with the product's cache-hit mix and occupancy (all lanes busy, the
    product's slab arithmetic included): tools/calib/step_roof.hip -> profiles/r3_step_roof.jsonl.  None if the file is missing."""
    try:
        for line in open(os.path.join(ROOT, "profiles", "r3_step_roof.jsonl")):
            d = json.loads(line)
            if d["variant"].startswith("gather_slab (hit mix, 64 lanes)"):
                return float(d["G_lane_steps_s"])
    except Exception:
        pass
    return None


def baseline_metric():
    """The metric string of BASELINE.json (the file travels with the repo); the literal is its value at the time of writing."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Mray/s at 1920x1080, 8 spp, Sponza BVH; 1/2/4/8-GPU scaling + % HBM roofline"


def cpu_quota_cores():
    """CPU time this process may use, in cores, from the cgroup (v2 cpu.max, v1 cfs quota); None = unlimited.  A GPU box of the pool
    shows all 256 hardware threads of its host but grants a 1-GPU lease 16 cores' worth of time: 256 runnable threads are then
    throttled to 1/16 each (round 2's 13 kray/s per thread -- ~1 us per node visit -- was that, not memory placement)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["tiles", "samples"], default="tiles")
    ap.add_argument("--single-process", action="store_true",
                    help="drive all --gpus devices from THIS process through mipt_render_multi (RCCL inside the library: the reference's "
                         "one-process host model).  Implied when --gpus N > 1 is started without torch.distributed.run")
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--traversal", choices=["culled", "reference"], default="culled")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="target CPU time of the cpu_baseline sample (both thread counts together)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="N > 1: skip the second (other sharding mode) measurement")
    ap.add_argument("--no-render-multi", action="store_true", help="N = 1: skip the in-library mipt_render_multi (RCCL) leg")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 code path on fewer GPUs than ranks (collectives staged through host memory)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 when a communicator is created (seen from
    # mipt_render_multi and torch.distributed alike), so fd 1 is pointed at stderr for the whole run and the JSON line is
    # written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # ---- launch model ----
    #  * one process per GPU: started by torch.distributed.run (RANK / WORLD_SIZE set), collectives through torch.distributed (RCCL);
    #  * ONE process for all GPUs: `python bench.py --gpus N` as is (or --single-process): mipt_multi_create over N devices, every
    #    frame ONE mipt_render_multi_device call -- a host thread per device, ncclCommInitAll communicators, one ncclGather /
    #    ncclReduce, assemble kernel on device 0.  This is how the reference's single-threaded host (src/main.rs:46,
    #    src/renderer.rs:50-63) would drive a node.  Nothing is re-launched or exec'd in either model.
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    single = args.single_process or (args.gpus > 1 and not launched)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if single:
        if launched and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("bench.py: --single-process drives all GPUs from one process; do not start it under torch.distributed.run with more than one rank")
        rank, world, local_rank = 0, args.gpus, 0
    else:
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start N ranks with torch.distributed.run --nproc-per-node N, "
                             "or no launcher at all (one process then drives all N GPUs through mipt_render_multi)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X backend has no CPU fallback")
    lib = rrt.load()
    if single:
        visible = int(lib.mipt_device_count())
        if visible < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {max(visible, 0)} HIP device(s) visible")
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not single:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    on_host = args.dist_backend == "gloo"
    multi_proc = world > 1 and not single

    def all_reduce_(t, op=dist.ReduceOp.SUM):
        if on_host:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    # ---- scene (identical on every rank: seeded generator + deterministic builder) ----
    t0 = time.time()
    tris, mats, texs, cam = synth.atrium_scene(n_target=args.tris, tex_size=args.tex_size)
    scene = rrt.Scene.from_arrays(tris, mats, texs, build_bvh=False)
    del tris
    scene.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    # the frame buffers first: like any host that owns a framebuffer before it loads a scene, and it keeps the process's very first
    # device allocation (~60-80 ms of one-time HIP runtime work, whoever makes it) out of the scene-setup figure below
    w, h, spp, depth = args.width, args.height, args.spp, args.depth
    n_pix = w * h
    d_frame = torch.empty(n_pix * 3, dtype=torch.float32, device=dev)
    d_rgba = torch.empty(n_pix * 4, dtype=torch.uint8, device=dev)       # cpu::render_scene's Vec<u8> (cpu.rs:60-67): part of every timed frame
    d_frame.zero_(); d_rgba.zero_()
    torch.cuda.synchronize(dev)
    t1 = time.time()
    # Scene setup = ONE call: the triangle array crosses PCIe once, BVH::build (bvh.rs:13-161) and the device layout are produced in
    # HBM (mipt_scene_create_from_triangles; identical tree and byte-identical layout to the host path, tests/test_gpu_scene_device.py).
    if single:
        multi = scene.upload_multi(list(range(world)), from_triangles=True)   # device 0 builds; replicas are device-to-device copies
        handle = None
        first = lib.mipt_multi_scene(multi, 0)
    else:
        multi = None
        handle = first = scene.upload_from_triangles(local_rank)
    t2 = time.time()
    setup_info = scene.info(first)
    replica_ms = []
    if single:
        for i in range(1, world):
            replica_ms.append(round(scene.info(lib.mipt_multi_scene(multi, i))["total_ms"], 1))
    # for the oracle legs below (outside every timed region): the tree and the triangle order BVH::build leaves in the host's Scene
    scene._fetch_bvh(first)
    t3 = time.time()
    log(rank, f"scene: {len(scene.tris)} tris, {len(scene.bvh_nodes)} nodes; gen {t1 - t0:.1f}s, setup call {t2 - t1:.3f}s "
              f"(upload {setup_info['upload_ms']:.0f} ms, device bvh {setup_info['build_ms']:.1f} ms, device layout {setup_info['layout_ms']:.1f} ms), "
              f"tree read-back for the oracle {t3 - t2:.2f}s" + (f" ({world} replicas, one process: {replica_ms} ms)" if single else ""))

    trav = L.TRAVERSAL_CULLED if args.traversal == "culled" else L.TRAVERSAL_REFERENCE
    cam_ptr = L.ptr(scene.camera.uniform)
    stream = torch.cuda.current_stream(dev)
    if not single:
        n_slot = int(lib.mipt_packed_pixels(w, h, world)) if world > 1 else n_pix
        d_local = torch.empty(max(n_slot, n_pix) * 3, dtype=torch.float32, device=dev)
        d_all = torch.empty(world * n_slot * 3, dtype=torch.float32, device=dev) if world > 1 else None
    s_begin, s_count = sharding.sample_ranges(spp, world)[rank]

    def sync_devices():
        if single:
            for i in range(world):
                torch.cuda.synchronize(i)
        else:
            torch.cuda.synchronize(dev)

    def tonemap(divisor=1.0):
        """sRGB + quantise epilogue of cpu.rs:60-64 on the assembled frame (N > 1 / samples mode: after the collective)."""
        L.check(lib.mipt_tonemap_device(C.c_void_p(d_frame.data_ptr()), n_pix, C.c_float(divisor), C.c_void_p(d_rgba.data_ptr()),
                                        C.c_void_p(stream.cuda_stream)), "mipt_tonemap_device")

    def render(mode, extra_flags=0, traversal=trav, out=None, rgba=None):
        """One launch of this rank's share in `mode`; returns MiptStats as a dict."""
        if mode == "tiles":
            opt = rrt.make_options(w, h, spp, depth, L.SEED_PIXEL_STREAM, traversal, (L.FLAG_PACKED if world > 1 else 0) | extra_flags,
                                   rank, world, cull_margin=L.CULL_MARGIN_SAFE)
        else:
            opt = rrt.make_options(w, h, max(s_count, 1), depth, L.SEED_PER_SAMPLE, traversal, L.FLAG_SUM | extra_flags,
                                   sample_begin=s_begin, cull_margin=L.CULL_MARGIN_SAFE)
        st = L.MiptStats()
        buf = d_local if out is None else out
        L.check(lib.mipt_render_device(handle, cam_ptr, C.byref(opt), C.c_void_p(buf.data_ptr()), None if rgba is None else C.c_void_p(rgba.data_ptr()),
                                       C.c_void_p(stream.cuda_stream), C.byref(st)), "mipt_render_device")
        return st.as_dict()

    def render_node(mode, extra_flags=0):
        """One frame on all devices through the library's own multi-GPU entry; the frame lands in d_frame (device 0)."""
        opt = rrt.make_options(w, h, spp, depth, L.SEED_PIXEL_STREAM if mode == "tiles" else L.SEED_PER_SAMPLE, trav, extra_flags,
                               cull_margin=L.CULL_MARGIN_SAFE)
        st = L.MiptMultiStats()
        L.check(lib.mipt_render_multi_device(multi, cam_ptr, C.byref(opt), L.MULTI_TILES if mode == "tiles" else L.MULTI_SAMPLES,
                                             C.c_void_p(d_frame.data_ptr()), C.c_void_p(d_rgba.data_ptr()), C.byref(st)), "mipt_render_multi_device")
        return st.as_dict()

    def step(mode):
        """One whole frame = cpu::render_scene (cpu.rs:13-68): trace, (collective,) mean, sRGB, RGBA8."""
        if single:
            return render_node(mode)
        if mode == "tiles" and world == 1:
            return render(mode, out=d_frame, rgba=d_rgba)            # trace kernel + tonemap kernel on the same stream, one call
        if mode == "samples" and s_count == 0:                      # more ranks than samples: this rank contributes zeros
            d_local.zero_()
            st = {"kernel_ms": 0.0}
        else:
            st = render(mode)
        if mode == "tiles":
            if world > 1:
                part = d_local[: n_slot * 3]
                if on_host:
                    c_all = torch.empty(d_all.shape, dtype=d_all.dtype)
                    dist.all_gather_into_tensor(c_all, part.cpu())
                    d_all.copy_(c_all)
                else:
                    dist.all_gather_into_tensor(d_all, part)
                L.check(lib.mipt_unpack_tiles(C.c_void_p(d_all.data_ptr()), w, h, world, C.c_void_p(d_frame.data_ptr()),
                                              C.c_void_p(stream.cuda_stream)), "mipt_unpack_tiles")
            tonemap()
        else:
            part = d_local[: n_pix * 3]
            if world > 1:
                if on_host:
                    c = part.cpu()
                    dist.reduce(c, dst=0, op=dist.ReduceOp.SUM)
                    part.copy_(c)
                else:
                    dist.reduce(part, dst=0, op=dist.ReduceOp.SUM)   # ONE ncclReduce(sum, f32) of the HDR sums
            torch.div(part, float(spp), out=d_frame)                 # cpu.rs:60, once, on the reduced sum (rank 0's is the frame)
            tonemap()
        return st

    def count(mode):
        """Counting build, outside the timed region (deterministic: same counts as the timed launches).
        Returns (whole-job totals, the counts of rank 0's / device 0's own launch)."""
        if single:
            tot_st = render_node(mode, extra_flags=L.FLAG_COUNT | L.FLAG_TOUCHED)
            d0 = L.MiptStats()
            L.check(lib.mipt_multi_device_stats(multi, 0, C.byref(d0)), "mipt_multi_device_stats")
            local = {k: int(getattr(d0, k)) for k in COUNT_KEYS}
            local["touched_lines"] = list(d0.touched_lines)
            return {k: int(tot_st[k]) for k in COUNT_KEYS}, local
        if mode == "samples" and s_count == 0:
            cst = {k: 0 for k in COUNT_KEYS}
        else:
            cst = render(mode, extra_flags=L.FLAG_COUNT | L.FLAG_TOUCHED)
        counts = torch.tensor([cst[k] for k in COUNT_KEYS], dtype=torch.int64, device=dev)
        local_counts = {k: int(v) for k, v in zip(COUNT_KEYS, counts.tolist())}
        local_counts["touched_lines"] = list(cst.get("touched_lines", [0, 0]))
        if world > 1:
            all_reduce_(counts)
        return {k: int(v) for k, v in zip(COUNT_KEYS, counts.tolist())}, local_counts

    def measure(mode):
        tot, local_counts = count(mode)
        want_pixels = n_pix if mode == "tiles" else n_pix * min(world, spp)
        assert tot["pixels"] == want_pixels, (tot["pixels"], want_pixels)
        for _ in range(args.warmup):
            step(mode)
        if multi_proc:
            dist.barrier()
        sync_devices()
        t_start = time.perf_counter()
        steps_st = []
        for _ in range(args.steps):
            steps_st.append(step(mode))
        if multi_proc:
            dist.barrier()
        sync_devices()
        elapsed = time.perf_counter() - t_start
        if multi_proc:
            el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            all_reduce_(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
        return elapsed, tot, local_counts, steps_st

    mode = args.mode
    elapsed, tot, local_counts, steps_st = measure(mode)
    log(rank, f"{mode}: counts {tot}")
    value = tot["rays"] * args.steps / elapsed / 1e6
    frame_primary = d_frame.clone()
    import zlib
    frame_crc = zlib.crc32(frame_primary.cpu().numpy().tobytes()) & 0xFFFFFFFF     # identifies the frame across runs / launch models

    # roofline of the dominant kernel (pt_trace_kernel): the launches of rank 0 (one process per GPU) / device 0 (one process)
    if single:
        kernel_ms = [st["device_kernel_ms"][0] for st in steps_st]
    else:
        kernel_ms = [st["kernel_ms"] for st in steps_st]
    avg_kernel_s = max(float(np.mean(kernel_ms)), 1e-9) * 1e-3
    n_local_samples = local_counts["pixels"] * (spp if mode == "tiles" else s_count)
    alg_bytes = algorithmic_bytes(local_counts, n_local_samples)
    dev_bytes = device_bytes(local_counts, local_counts["pixels"])
    achieved = alg_bytes / avg_kernel_s / 1e9
    lane_steps = local_counts["inner_steps"] + local_counts["tri_tests"]
    roof = step_roof() if (args.tris, w, h) == (10_000_000, 1920, 1080) and mode == "tiles" and args.traversal == "culled" else None
    unique_bytes = (sum(local_counts["touched_lines"]) * 128 + 12 * local_counts["pixels"]) if sum(local_counts["touched_lines"]) else None
    traffic, prov = pmc_traffic(args.tris, w, h, spp, depth, args.traversal, mode) if world == 1 else (None, "N > 1")
    seeds = "pixel-stream seeds (cpu.rs:28-29)" if mode == "tiles" else "per-sample seeds (rt_compute.wgsl:102)"
    how = ("one process, mipt_multi_create (ncclCommInitAll) + one mipt_render_multi_device call per frame" if single
           else "one process per GPU over torch.distributed (RCCL)" if world > 1 else "one process, one GPU")
    if mode == "tiles":
        shard = (f"8x8 image tiles round-robin over {world} GPU(s)"
                 + ((", ONE ncclGather of rank-packed f32 tile slices to device 0 + de-interleave kernel per frame" if single else
                     ", one RCCL all-gather of packed tile slices per frame") if world > 1 or single else "") + f"; {how}")
    else:
        shard = (f"{spp} samples split over {world} GPU(s), every GPU all pixels"
                 + (", ONE ncclReduce(sum, f32) of the HDR sums to device 0 per frame" if world > 1 or single else "") + f"; {how}")
    result = {
        "metric": baseline_metric(),
        "value": round(value, 3), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"atrium stand-in for Intel Sponza + curtains: {len(scene.tris)} tris, {len(scene.bvh_nodes)} BVH nodes, "
                               f"{w}x{h}, {spp} spp, max_ray_depth {depth}, whole frame incl. the sRGB + RGBA8 epilogue (cpu.rs:60-67), traversal {args.traversal}"
                               + (f" (margin {L.CULL_MARGIN_SAFE})" if args.traversal == "culled" else "") + f", {seeds}",
                   "mode": mode, "sharding": shard, "launch": "single-process" if single else ("torch.distributed.run" if world > 1 else "single-gpu"),
                   "rays_per_frame": tot["rays"], "paths_per_frame": n_pix * spp, "frame_crc32": f"{frame_crc:08x}"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_is": "algorithmic (the contract's definition: SURVEY 8(d) bytes / kernel time / 8 TB/s) -- NOT bytes moved; the physical "
                                "figures are memside_frac_measured (memory side, PMC) and step_roof.frac (the kernel's own roof)",
                     "basis": "ALGORITHMIC bytes in the reference's layouts (Node 32 B, Triangle 112 B, Material 80 B; SURVEY 8(d)) / kernel time. "
                              "A cache-side figure of merit that may exceed 1: most of these bytes are L1/L2 hits and the device layout is leaner -- "
                              "see device_GBs (requested by the kernel's loads) and traffic / memside_frac_measured (memory side, measured)",
                     "traffic": traffic,
                     "traffic_note": ("bytes per launch at the L2's memory side (HBM + Infinity Cache together), rocprofv3 PMC in a separate run of this command: "
                                      "TCC_EA0_RDREQ_sum x 128 B (every request is a 128-B line fill, calibrated for per-lane 64-B gathers in "
                                      "profiles/r2_fetch_calibration.csv) + WRITE_SIZE; provenance in traffic_provenance") if traffic else
                                     f"null: {prov}",
                     "traffic_provenance": prov if traffic else None,
                     "memside_frac_measured": round(traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                     # compulsory traffic: distinct 128-B lines of the BVH / triangle / attribute streams this launch reads (device bitmap
                     # in the counting launch, MIPT_FLAG_TOUCHED) x 128 B + the framebuffer; traffic / this = how often a line is re-fetched
                     "unique_line_bytes": unique_bytes,
                     "unique_lines": {"bvh_pairs_and_triangle_stream": local_counts["touched_lines"][0], "triangle_attributes": local_counts["touched_lines"][1]},
                     "refetch_factor": round(traffic / unique_bytes, 1) if traffic and unique_bytes else None,
                     "kernel": "pt_trace_kernel", "kernel_ms": round(avg_kernel_s * 1e3, 3), "kernel_source_sha": kernel_source_sha(),
                     "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_ray": round(alg_bytes / max(local_counts["rays"], 1), 1),
                     "device_bytes_per_launch": dev_bytes, "device_GBs": round(dev_bytes / avg_kernel_s / 1e9, 1),
                     "mray_s_kernel": round(local_counts["rays"] / avg_kernel_s / 1e6, 2),
                     # a roof in the kernel's own unit: one lane-step = one 64-B record gathered at an address that depends on the last one
                     "step_roof": ({"unit": "G lane-steps/s", "achieved": round(lane_steps / avg_kernel_s / 1e9, 1), "peak": roof,
                                    "frac": round(lane_steps / avg_kernel_s / 1e9 / roof, 3),
                                    "basis": "peak = a dependent 4 x 16-B gather chain + the product's slab arithmetic, every lane busy, same occupancy, "
                                             "the product's measured L1 / L2 / memory-side hit mix (tools/calib/step_roof.hip, profiles/r3_step_roof.jsonl); "
                                             "achieved = (inner steps + triangle tests) / kernel time"} if roof else None)},
    }
    if single:
        result["single_process"] = {
            "entry": "mipt_render_multi_device (frame + RGBA8 stay on device 0; no D2H in the timed region)",
            "n_gt_1_evidence": "the n > 1 code of this entry runs with 2 / 3 / 8 logical ranks on one GPU against an RCCL test double "
                               "(tests/test_gpu_multirank.py); real RCCL with more than one rank has only run where this line has n_gpus > 1",
            "device_kernel_ms": [round(float(np.mean([st["device_kernel_ms"][i] for st in steps_st])), 3) for i in range(world)],
            "collective_ms": round(float(np.mean([st["collective_ms"] for st in steps_st])), 3),
            "call_wall_ms": round(float(np.mean([st["wall_ms"] for st in steps_st])), 3)}

    # ---- N > 1: the other sharding mode, same steps (the tile split is floored by the per-pixel RNG chain; samples are not) ----
    if world > 1 and not args.no_other_mode:
        other = "samples" if mode == "tiles" else "tiles"
        e2, tot2, _, st2 = measure(other)
        k2 = [st["device_kernel_ms"][0] for st in st2] if single else [st["kernel_ms"] for st in st2]
        result["other_mode"] = {"mode": other, "value": round(tot2["rays"] * args.steps / e2 / 1e6, 3), "unit": "Mray/s",
                                "ms_per_step": round(e2 / args.steps * 1e3, 3), "rays_per_frame": tot2["rays"],
                                "kernel_ms_rank0": round(float(np.mean(k2)), 3)}

    # ---- parity evidence at the bench size (outside the timed region) ----
    if rank == 0 and not args.no_parity:
        par = {"against": "oracle/pt_oracle.c = C restatement of the rayon CPU backend incl. its libm (glibc 2.35 cosf/log10f/powf restated, "
                          "== the host's libm on all 2^32 arguments: tests/test_libm_pin.py)"}
        if single:
            handle = scene.upload(0)                  # a plain single-GPU replica next to the node handle, for the comparisons below

        def solo_frame(seed_mode, traversal=trav):
            """The whole frame rendered by ONE launch on this GPU (no sharding)."""
            buf = torch.empty(n_pix * 3, dtype=torch.float32, device=dev)
            opt1 = rrt.make_options(w, h, spp, depth, seed_mode, traversal, 0, 0, 1, cull_margin=L.CULL_MARGIN_SAFE)
            st1 = L.MiptStats()
            L.check(lib.mipt_render_device(handle, cam_ptr, C.byref(opt1), C.c_void_p(buf.data_ptr()), None,
                                           C.c_void_p(stream.cuda_stream), C.byref(st1)), "mipt_render_device")
            torch.cuda.synchronize(dev)
            return buf, st1.as_dict()
        if (world > 1 or single) and mode == "tiles":
            # the gathered + de-interleaved frame must equal a frame rendered by one GPU alone, bit for bit
            solo, _ = solo_frame(L.SEED_PIXEL_STREAM)
            par["gathered_frame_equals_single_gpu_frame"] = bool(torch.equal(frame_primary.view(torch.int32), solo.view(torch.int32)))
            del solo
        if (world > 1 or single) and mode == "samples":
            solo, _ = solo_frame(L.SEED_PER_SAMPLE)
            par["reduced_frame_max_rel_diff_vs_single_gpu"] = float(((frame_primary - solo).abs() / solo.abs().clamp_min(1e-6)).max().item())
            del solo
        if world == 1 and args.traversal == "culled" and mode == "tiles":
            ref_buf, rst = solo_frame(L.SEED_PIXEL_STREAM, traversal=L.TRAVERSAL_REFERENCE)
            par["culled_equals_reference_traversal"] = bool(torch.equal(frame_primary.view(torch.int32), ref_buf.view(torch.int32)))
            # the headline uses the culled traversal (identical frame, re-checked above on every run); the CPU backend's own
            # un-culled traversal (ray.rs:69-81) on the same frame, one launch:
            result["reference_traversal"] = {"kernel_ms": round(rst["kernel_ms"], 3),
                                             "mray_s_kernel": round(local_counts["rays"] / rst["kernel_ms"] / 1e3, 2)}
            del ref_buf
        result["parity"] = par

    # outside the timed region, for the record: what it took to get the scene onto the GPU(s)
    result["scene_setup"] = {"entry": "mipt_multi_create_from_triangles" if single else "mipt_scene_create_from_triangles",
                             "call_s": round(t2 - t1, 3), "total_ms": round(setup_info["total_ms"], 1),
                             "upload_ms": round(setup_info["upload_ms"], 1), "device_bvh_build_ms": round(setup_info["build_ms"], 2),
                             "device_layout_ms": round(setup_info["layout_ms"], 2), "replica_ms": replica_ms,
                             "geometry_bytes": setup_info["geometry_bytes"],
                             "note": "one call (after the process's frame buffers exist, i.e. not the process's first device allocation): "
                                     "triangles host -> device once (staged through a pinned ring), BVH::build (bvh.rs:13-161) and the whole "
                                     "device layout in HBM; tree identical to mipt_bvh_build's, layout byte-identical to the host restatement in "
                                     "libmipt_diag.so (tests/test_gpu_scene_device.py, tests/test_gpu_fullsize.py); round 3: 0.44 s + 0.89 s"}
    # ---- N = 1: the in-library multi-GPU path (mipt_render_multi: RCCL communicator + gather / reduce behind the C ABI) ----
    if rank == 0 and world == 1 and not single and not args.no_render_multi:
        try:
            r = rrt.Renderer.new(rrt.RendererOptions(samples=spp, max_ray_depth=depth, output_image_dimensions=(w, h), output_image_path="/dev/null",
                                                     traversal=trav, cull_margin=L.CULL_MARGIN_SAFE))
            hdr_m, _, mst = r.render_buffers_multi(scene, mode=L.MULTI_TILES if mode == "tiles" else L.MULTI_SAMPLES, device_ids=[local_rank],
                                                   want_rgba8=False)
            same = bool(np.array_equal(hdr_m.reshape(-1).view(np.uint32), frame_primary.cpu().numpy().view(np.uint32)))
            result["render_multi"] = {"entry": "mipt_render_multi (one process, ncclCommInitAll)", "n_devices": mst["n_devices"],
                                      "kernel_ms": round(mst["kernel_ms"], 3), "collective_ms": round(mst["collective_ms"], 3),
                                      "wall_ms_incl_d2h": round(mst["wall_ms"], 3), "equals_bench_frame": same}
        except Exception as e:  # noqa: BLE001  -- the leg is evidence, not the metric
            result["render_multi"] = {"error": str(e)[:200]}

    # ---- CPU baseline: the oracle (C restatement of the rayon backend) on a strided pixel sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc
        mats_arr = scene.materials_array()
        seed_mode = 0 if mode == "tiles" else 1

        hw_threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = cpu_quota_cores()
        n_thr = max(1, min(hw_threads, int(np.ceil(quota)))) if quota else hw_threads     # one thread per core the cgroup grants

        def cpu(stride, threads=n_thr, spread=0):
            return orc.render(scene.tris, scene.bvh_nodes, mats_arr, scene.textures, scene.camera.uniform, w, h, spp, depth,
                              cull=0, pix_stride=stride, want_rgba8=False, threads=threads, seed_mode=seed_mode, spread_pages=spread)
        _, _, ps = cpu(8191)
        rate = ps["rays"] / max(ps["seconds"], 1e-6)
        want_rays = rate * args.cpu_seconds / 3
        stride = max(1, int(tot["rays"] / max(want_rays, 1))) | 1   # odd stride: samples every image column
        hdr_cpu, _, cs = cpu(stride)

        def summary(s):
            v = s["rays"] / s["seconds"] / 1e6
            return {"value": round(v, 4), "threads": s["threads_used"], "per_thread_mray_s": round(v / s["threads_used"], 5),
                    "seconds": round(s["seconds"], 2), "uniform_blocks": s["n_blocks"],
                    "slowest_block_over_mean_block": round(s["block_sec_max"] / max(s["block_sec_mean"], 1e-9), 3)}
        head = summary(cs)
        result["cpu_baseline"] = {"value": head["value"], "unit": "Mray/s", "cores": cs["threads_used"], "kind": "port",
                                  "sample": f"every {stride}th pixel of the same frame ({cs['rays']} rays, {cs['seconds']:.1f} s), "
                                            "C restatement of the reference's rayon backend (no t-max cull), all samples and bounces, "
                                            "uniform contiguous pixel blocks as cpu.rs:22-26",
                                  "host": {"hardware_threads_visible": hw_threads, "cgroup_cpu_quota_cores": quota,
                                           "note": "threads = the cores the cgroup grants this process (cpu.max); rayon would start one thread per "
                                                   "visible hardware thread and be throttled to the same total"},
                                  "memory_placement": "reference-faithful: the scene arrays are allocated and filled by ONE thread (the reference builds its "
                                                      "Vecs on the main thread, scene.rs:44-85), so on a multi-socket host every page sits on one NUMA node; "
                                                      "numa_spread_run is the same sample with the pages first-touched share by share by the worker threads",
                                  "per_thread_mray_s": head["per_thread_mray_s"],
                                  "slowest_block_over_mean_block": head["slowest_block_over_mean_block"]}
        _, _, cn = cpu(stride, spread=1)
        result["cpu_baseline"]["numa_spread_run"] = summary(cn)
        if n_thr != hw_threads:
            _, _, cp = cpu(stride, threads=hw_threads)
            result["cpu_baseline"]["all_hardware_threads_run"] = summary(cp)       # what a rayon pool sized by available_parallelism would do here
        if not args.no_parity:
            got = frame_primary.cpu().numpy().reshape(h, w, 3)
            idx = np.arange(0, n_pix, stride)
            a = got.reshape(-1, 3)[idx].view(np.uint32)
            b = hdr_cpu.reshape(-1, 3)[idx].view(np.uint32)
            result["parity"]["oracle_bit_exact_on_sample"] = bool(np.array_equal(a, b))
            result["parity"]["sample_pixels"] = int(len(idx))
            result["parity"]["rmse"] = float(np.sqrt(np.mean((got.reshape(-1, 3)[idx].astype(np.float64) - hdr_cpu.reshape(-1, 3)[idx]) ** 2)))

    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if multi_proc:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

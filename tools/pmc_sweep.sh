#!/bin/bash
# GPU box: scene-size sweep of the memory-side traffic -- the same view and frame (1920x1080, 8 spp, depth 64, culled) on the atrium at
# 1 / 2 / 5 / 10 M triangles: line fills per ray (TCC_EA0_RDREQ_sum), L2 requests and hits, kernel time, and the frame's distinct lines
# (bench.py roofline.unique_lines).  Shows where the footprint leaves the L2 (4 MiB per XCD) and the 256 MiB Infinity Cache.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sweep; mkdir -p $out
for n in 1000000 2000000 5000000 10000000; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $out/t$n -- python3 bench.py --tris $n --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-render-multi > $out/bench_$n.json 2> $out/t$n.err
done
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
with open(out + "/scene_size_sweep.csv", "w") as f:
    f.write("# atrium, same camera, 1920x1080 x 8 spp, depth 64, culled 2^-7; counters = mean of 3 launches after 1 warm-up (rocprofv3 --pmc); scene bytes = pair records + triangle stream + attribute stream\n")
    f.write("triangles,scene_MB,rays,kernel_ms,line_fills_per_ray,l2_requests_per_ray,l2_hit_rate,unique_lines,unique_MB,refetch_factor,memside_GBs\n")
    for n in (1000000, 2000000, 5000000, 10000000):
        b = json.load(open(f"{out}/bench_{n}.json"))
        per = collections.defaultdict(dict)
        for fn in glob.glob(f"{out}/t{n}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if "pt_trace_kernel<false" in r["Kernel_Name"]:
                    d = per[int(r["Dispatch_Id"])]
                    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp = sorted(per)[1:]
        m = {k: sum(per[d][k] for d in disp) / len(disp) for k in ("TCC_EA0_RDREQ_sum", "TCC_REQ_sum", "TCC_HIT_sum")}
        rays = b["config"]["rays_per_frame"]
        ul = b["roofline"]["unique_lines"]
        lines = ul["bvh_pairs_and_triangle_stream"] + ul["triangle_attributes"]
        import re
        tris = int(re.search(r"(\d+) tris", b["config"]["workload"]).group(1)); nodes = int(re.search(r"(\d+) BVH nodes", b["config"]["workload"]).group(1))
        scene_mb = ((nodes - 1) // 2 * 64 + tris * 128) / 1e6
        kms = b["roofline"]["kernel_ms"]
        f.write(f"{tris},{scene_mb:.0f},{rays},{kms},{m['TCC_EA0_RDREQ_sum'] / rays:.2f},{m['TCC_REQ_sum'] / rays:.2f},{m['TCC_HIT_sum'] / m['TCC_REQ_sum']:.3f},"
                f"{lines},{lines * 128 / 1e6:.0f},{m['TCC_EA0_RDREQ_sum'] / lines:.1f},{m['TCC_EA0_RDREQ_sum'] * 128 / kms / 1e6:.0f}\n")
print(open(out + "/scene_size_sweep.csv").read())
PY

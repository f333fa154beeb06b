#!/bin/bash
# GPU box: PMC counters of the device BVH builder's kernels on the 10 M-triangle scene (one build per pass):
#   bash tools/bvh_pmc.sh <tag>   ->  gpurun_out/bvhpmc_<tag>/summary.csv  (per kernel: launches, sum of every counter)
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/bvhpmc_$tag; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/p1 -- python3 tools/bvhdev_prof.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $out/p2 -- python3 tools/bvhdev_prof.py > $out/p2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p3 -- python3 tools/bvhdev_prof.py > $out/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p4 -- python3 tools/bvhdev_prof.py > $out/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/p5 -- python3 tools/bvhdev_prof.py > $out/p5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bvhdev_prof.py > $out/trace.log 2>&1
python3 - "$out" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for d in sorted(glob.glob(out + "/p*/")):
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in k else k[:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if d.rstrip("/").endswith("p1") and (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); launches[k] += 1
dur = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"]
        k = k.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in k else k[:40]
        dur[k] = dur.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6
names = sorted({c for v in agg.values() for c in v})
with open(out + "/summary.csv", "w") as fh:
    fh.write("kernel,launches,total_ms," + ",".join(names) + "\n")
    for k in sorted(agg, key=lambda k: -dur.get(k, 0)):
        fh.write(f"{k.replace(',', ';')},{launches[k]},{dur.get(k, 0):.3f}," + ",".join(f"{agg[k].get(c, 0):.0f}" for c in names) + "\n")
print(open(out + "/summary.csv").read())
PY

#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes for the trace kernel on config M; writes gpurun_out/pmc_<tag>/ with
#   kernel_stats.csv  (--kernel-trace --stats of the default bench command)
#   pmc_summary.csv   (mean per launch of every counter, first line = provenance: kernel source hash, date, command)
# Counters are collected in their own runs, one group per pass (TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2; SQ 8).
tag=${1:-run}
extra=${2:-}            # e.g. "--mode samples" or shading variants; recorded in the provenance line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity --no-render-multi $extra"
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-render-multi $extra > $out/bench.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p1 -- $B > /dev/null 2> $out/p1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p2 -- $B > /dev/null 2> $out/p2.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_REQ_sum --output-format csv -d $out/p5 -- $B > /dev/null 2> $out/p5.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/p3 -- $B > /dev/null 2> $out/p3.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/p4 -- $B > /dev/null 2> $out/p4.err
python3 - $out "$extra" <<'PY'
import csv, sys, glob, collections, os, datetime
sys.path.insert(0, os.getcwd())
out, extra = sys.argv[1], sys.argv[2]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py")
rows = []
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_trace_kernel<false" in r["Kernel_Name"]:
            rows.append((r["Counter_Name"], float(r["Counter_Value"])))
agg = collections.defaultdict(list)
for n, v in rows: agg[n].append(v)
import hashlib
h = hashlib.sha256()
for f in ("pt_kernel.hip", "pt_device_math.h", "pt_kernel.h", "glibc_flt32_data.h", "mipt_api.cpp", "bvh_build.cpp"):
    h.update(open(os.path.join("rust_ray_tracing_amd", "csrc", f), "rb").read())
with open(out + "/pmc_summary.csv", "w") as f:
    f.write(f"# kernel_sha={h.hexdigest()[:16]} date={datetime.date.today().isoformat()} tool=tools/pmc.sh command=bench.py--steps2--warmup0{extra.replace(' ', '')}\n")
    f.write("kernel,counter,mean_per_launch,launches\n")
    for n, v in sorted(agg.items()):
        f.write(f"pt_trace_kernel<false;true>,{n},{sum(v) / len(v):.1f},{len(v)}\n")
print(open(out + "/pmc_summary.csv").read())
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    txt = open(f).read(); open(out + "/kernel_stats.csv", "w").write(txt); print(txt)
PY

#!/bin/bash
# GPU box: issue / wait / memory counters of library variants on config M, one rocprofv3 --pmc pass per counter group over
# tools/ab.py (2 launches per variant; the trace kernel's dispatches are attributed to the variants by order).
#   tools/ab_sq.sh <tag> "<ab.py args>" lib1.so lib2.so ...
tag=$1; shift; abargs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/absq_$tag; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 tools/ab.py --reps 2 $abargs "$@" > $out/run$i.txt 2> $out/p$i.err || { tail -5 $out/p$i.err; exit 1; }
done
python3 - $out "$@" <<'PY'
import csv, glob, sys, collections
out, libs = sys.argv[1], sys.argv[2:]
table = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/p*/")):
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pt_trace_kernel" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = sorted(per)
    reps = len(disp) // len(libs)
    for i, lib in enumerate(libs):
        ds = disp[i * reps:(i + 1) * reps][-1:]        # the second (warm) launch
        for k in per[ds[0]]:
            table[k][lib] = per[ds[0]][k]
with open(out + "/sq_by_variant.csv", "w") as f:
    f.write("counter," + ",".join(l.split("/")[-1] for l in libs) + "\n")
    for k in sorted(table):
        f.write(k + "," + ",".join(f"{table[k].get(l, 0):.0f}" for l in libs) + "\n")
print(open(out + "/sq_by_variant.csv").read())
PY

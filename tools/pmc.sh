#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes for the trace kernel on config M; writes gpurun_out/pmc_<tag>/
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity"
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity > $out/bench.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p1 -- $B > /dev/null 2> $out/p1.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p2 -- $B > /dev/null 2> $out/p2.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/p3 -- $B > /dev/null 2> $out/p3.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/p4 -- $B > /dev/null 2> $out/p4.err
python - $out <<'PY'
import csv,sys,glob,collections,os
out=sys.argv[1]
rows=[]
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_trace_kernel<false" in r["Kernel_Name"]:
            rows.append((r["Counter_Name"], float(r["Counter_Value"]), r.get("VGPR_Count",""), r.get("LDS_Block_Size",""), r.get("Grid_Size","")))
agg=collections.defaultdict(list)
for n,v,*rest in rows: agg[n].append(v)
with open(out+"/pmc_summary.csv","w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for n,v in sorted(agg.items()):
        f.write(f"pt_trace_kernel<false;true>,{n},{sum(v)/len(v):.1f},{len(v)}\n")
print(open(out+"/pmc_summary.csv").read())
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True): print(open(f).read())
PY

#!/bin/bash
# GPU box: texture-addresser / L1 (TCP) counters of the trace kernel on config M -- is the L1 path the limiter?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_mem; mkdir -p $out
B="python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-parity"
i=0
# at most two counters of one block per pass (more: "exceeds the capabilities of the hardware" and rocprofv3 aborts)
for set in "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $out/m$i -- $B > /dev/null 2> $out/m$i.err || echo "set $i failed: $set" | tee -a $out/fail.txt
done
python - $out <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(out+"/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_trace_kernel<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out+"/pmc_mem_summary.csv","w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for n,v in sorted(agg.items()):
        f.write(f"pt_trace_kernel<false;true>,{n},{sum(v)/len(v):.1f},{len(v)}\n")
print(open(out+"/pmc_mem_summary.csv").read())
PY

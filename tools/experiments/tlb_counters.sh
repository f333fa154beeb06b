#!/bin/bash
# GPU box: address-translation counters of the trace kernel on config M (does the random walk over a 1.7 GB scene miss the CU's TLB?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tlb; mkdir -p $out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-render-multi"
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum --output-format csv -d $out/p1 -- $B > /dev/null 2> $out/p1.err
rocprofv3 --pmc TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum --output-format csv -d $out/p2 -- $B > /dev/null 2> $out/p2.err
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "pt_trace_kernel<false, true, 0>" in r["Kernel_Name"]:
            d = int(r["Dispatch_Id"]); per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in sorted(per)[1:]:
        for k, v in per[d].items(): acc[k].append(v)
with open(out + "/tlb_summary.csv", "w") as f:
    f.write("counter,mean_per_launch,launches\n")
    for k in sorted(acc):
        line = f"{k},{sum(acc[k]) / len(acc[k]):.1f},{len(acc[k])}"; f.write(line + "\n"); print(line)
PY

#!/bin/bash
# GPU box: builds gather_calib, runs it plain (timings) and under rocprofv3 --pmc (one pass per counter group, the program
# itself after `--`), and writes gpurun_out/calib/summary.csv (copied to profiles/r2_fetch_calibration.csv).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/calib; mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $out/gather_calib tools/calib/gather_calib.hip
$out/gather_calib > $out/timing.jsonl
cat $out/timing.jsonl
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p1 -- $out/gather_calib > /dev/null 2> $out/p1.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_MISS_sum --output-format csv -d $out/p2 -- $out/gather_calib > /dev/null 2> $out/p2.err
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/p3 -- $out/gather_calib > /dev/null 2> $out/p3.err
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
t = {}
for line in open(out + "/timing.jsonl"):
    d = json.loads(line); t[d["kernel"]] = d
vals = collections.defaultdict(dict)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in t:
            vals[k][r["Counter_Name"]] = vals[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
cols = ["FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_HIT_sum", "TCC_REQ_sum", "TCC_READ_sum"]
with open(out + "/summary.csv", "w") as f:
    f.write("kernel,records,payload_bytes,ms,payload_GBs," + ",".join(cols) + ",FETCH_SIZE_bytes,payload_over_FETCH_SIZE,bytes_per_RDREQ_if_payload,RDREQ_per_record\n")
    for k, d in t.items():
        v = vals.get(k, {})
        fs = v.get("FETCH_SIZE", 0.0) * 1024
        rd = v.get("TCC_EA0_RDREQ_sum", 0.0)
        f.write(f"{k},{d['records']},{d['payload_bytes']},{d['ms']},{d['payload_GBs']}," + ",".join(f"{v.get(c, float('nan')):.0f}" for c in cols) +
                f",{fs:.0f},{d['payload_bytes'] / fs if fs else float('nan'):.3f},{d['payload_bytes'] / rd if rd else float('nan'):.1f},{rd / d['records'] if rd else float('nan'):.3f}\n")
print(open(out + "/summary.csv").read())
PY

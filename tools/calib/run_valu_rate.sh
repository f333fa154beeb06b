#!/bin/bash
# GPU box: VALU issue-rate table (tools/calib/valu_rate.hip) + what SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES count for
# kernels of known instruction mix.  Writes gpurun_out/vr/{valu_rate.jsonl,pmc.csv}
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/vr; mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $out/valu_rate tools/calib/valu_rate.hip
timeout -k 10 150 $out/valu_rate > $out/valu_rate.jsonl
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- $out/valu_rate 5 > /dev/null 2> $out/p1.err
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
v = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        v[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE"]
with open(out + "/pmc.csv", "w") as f:
    f.write("kernel," + ",".join(cols) + "\n")
    for k in sorted(v, key=lambda s: int(s.split("<")[1].split(">")[0]) if "<" in s else -1):
        f.write(k.replace(",", ";") + "," + ",".join(f"{v[k][c][-1]:.0f}" if v[k][c] else "nan" for c in cols) + "\n")
print(open(out + "/pmc.csv").read())
PY

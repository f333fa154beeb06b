"""Summarise the rocprofv3 --pmc passes under <out>/p*/ for one kernel: mean per launch over the launches after the first (warm-up)
launch of every pass, with a provenance line (kernel source hash = bench.py's kernel_source_sha).  Used by tools/pmc.sh / pmc_variant.sh."""
import collections
import csv
import datetime
import glob
import os
import sys

out, filt, command = sys.argv[1], sys.argv[2], sys.argv[3]
want = filt.replace(" ", "")
agg = collections.defaultdict(list)
for d in sorted(glob.glob(out + "/p*/")):
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"].replace(" ", ""):
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = sorted(per)
    for dsp in disp[1:] if len(disp) > 1 else disp:          # drop the first (cold) launch
        for n, v in per[dsp].items():
            agg[n].append(v)
sys.path.insert(0, os.getcwd())
from rust_ray_tracing_amd.provenance import kernel_source_sha  # noqa: E402  (no torch, no library load)
with open(out + "/pmc_summary.csv", "w") as f:
    f.write(f"# kernel_sha={kernel_source_sha()} date={datetime.date.today().isoformat()} tool=tools/pmc_summary.py command={command} launches=after-1-warm-up\n")
    f.write("kernel,counter,mean_per_launch,launches\n")
    for n, v in sorted(agg.items()):
        f.write(f"{filt.replace(',', ';')},{n},{sum(v) / len(v):.1f},{len(v)}\n")
print(open(out + "/pmc_summary.csv").read())
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    txt = open(f).read()
    open(out + "/kernel_stats.csv", "w").write(txt)
    print(txt)

#!/bin/bash
# GPU box: A/B of library variants on config M -- kernel time (tools/ab.py, plain run) and the L2 / memory-side counters of the
# same variants (second run of ab.py under rocprofv3 --pmc; the trace kernel's dispatches are attributed to the variants by order).
#   tools/ab_pmc.sh <tag> "<ab.py args>" lib1.so lib2.so ...
tag=$1; shift; abargs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/abpmc_$tag; mkdir -p $out
python3 tools/ab.py --reps 6 $abargs "$@" > $out/times.txt 2> $out/times.err || { tail -5 $out/times.err; exit 1; }
cat $out/times.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $out/pmc -- python3 tools/ab.py --reps 2 $abargs "$@" > $out/pmc_run.txt 2> $out/pmc.err || { tail -5 $out/pmc.err; exit 1; }
python3 - $out "$@" <<'PY'
import csv, glob, sys, collections
out, libs = sys.argv[1], sys.argv[2:]
rows = []
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_trace_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
disp = sorted({d for d, _, _ in rows})
per = collections.defaultdict(dict)
for d, n, v in rows:
    per[d][n] = per[d].get(n, 0.0) + v
reps = len(disp) // len(libs)
with open(out + "/pmc_by_variant.csv", "w") as f:
    f.write("variant,launches,TCC_REQ_sum,TCC_HIT_sum,TCC_EA0_RDREQ_sum\n")
    for i, lib in enumerate(libs):
        ds = disp[i * reps:(i + 1) * reps]
        m = {k: sum(per[d].get(k, 0.0) for d in ds) / max(len(ds), 1) for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_EA0_RDREQ_sum")}
        line = f"{lib},{len(ds)},{m['TCC_REQ_sum']:.0f},{m['TCC_HIT_sum']:.0f},{m['TCC_EA0_RDREQ_sum']:.0f}"
        f.write(line + "\n"); print(line)
PY

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bvhtrace -- python3 tools/bvhdev_prof.py > gpurun_out/bvhtrace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/bvhtrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[1].split("::")[-1] if "anonymous" in r["Kernel_Name"] else r["Kernel_Name"][:30]) for r in rows]
rows.sort()
t0 = [s for s, e, n in rows if n.startswith("make_proxies")][0]
rows = [(s - t0, e - t0, n) for s, e, n in rows if s >= t0]
# group into levels: a new level starts at every build_* / big_setup launch burst after a gap; simpler: print a compact timeline
out = []
for s, e, n in rows:
    out.append(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  {n}")
open("gpurun_out/bvhtrace_timeline.txt", "w").write("\n".join(out))
print("\n".join(out[:400]))
PY

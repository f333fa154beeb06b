import csv, glob, sys
f = glob.glob("gpurun_out/setup_trace/*/*hip_api_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows]
rows.sort()
# first call = from the first hipGetDeviceCount after the last __hipRegister... to the first hipEventDestroy burst; simpler: print all non-register calls > 0.5 ms in the first 2 s after the first hipHostMalloc
t0 = [s for s, e, n in rows if n == "hipHostMalloc"][0] - 300_000_000
out = []
for s, e, n in rows:
    if s < t0 or n.startswith("__hip"): continue
    if (e - s) > 300_000 or n in ("hipHostMalloc", "hipMalloc"):
        out.append(f"{(s - t0) / 1e6:9.2f} ms  +{(e - s) / 1e6:8.2f} ms  {n}")
print("\n".join(out[:90]))

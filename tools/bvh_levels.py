"""Per-level summary of gpurun_out/bvhtrace_timeline.txt (written by tools/bvh_trace.sh): start, duration, the four longest kernels."""
import collections, sys
L = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bvhtrace_timeline.txt").read().split("\n")
rows = []
for l in L:
    p = l.split()
    if len(p) >= 5: rows.append((float(p[0]), float(p[1]), float(p[2]), p[4]))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, d, n in rows: tot[n] += d; cnt[n] += 1
for n, v in tot.most_common(16): print(f"{n:32s} {v/1e3:8.2f} ms  {cnt[n]} launches")
# A level ends with level_mark kernels (one per stream that had work).  The level's last mark is the one with nothing running when it
# ends and nothing starting for the next 7 us (the host's round trip); marks of streams that finish early sit inside the level.
skip = ("sizes_level", "bases_level", "sizes_run", "bases_run", "extract_order", "emit_nodes", "gather_tris", "make_proxies", "init_root")
work = [r for r in rows if not r[3].startswith("__amd") and r[3] not in skip and r[3] != "level_mark"]
marks = [r for r in rows if r[3] == "level_mark"]
bounds = []
for ms_, me, md, mn in marks:
    running = any(s < me and e > me for s, e, d, n in work)
    soon = any(me <= s < me + 7 for s, e, d, n in work)
    if not running and not soon: bounds.append(me)
lvl = []; cur = None; bi = 0
for s, e, d, n in work:
    while bi < len(bounds) and s >= bounds[bi]:
        bi += 1
        if cur: lvl.append(cur); cur = None
    if cur is None or (not marks and s > cur[1] + 40): 
        if cur: lvl.append(cur)
        cur = [s, e, {}]
    cur[1] = max(cur[1], e); cur[2][n] = cur[2].get(n, 0) + d
if cur: lvl.append(cur)
for i, (s, e, k) in enumerate(lvl):
    top = sorted(k.items(), key=lambda x: -x[1])[:4]
    print(i, f"{s/1e3:7.2f} {(e-s)/1e3:6.2f} ms ", " ".join(f"{n.replace('build_level', 'BL')}={v:.0f}" for n, v in top))
last = [r for r in rows if r[3] == "gather_tris"]
if last: print("end of gather_tris", last[-1][1] / 1e3, "ms")

"""CPU: which 128-B lines do a ray's traversal records fall into under a candidate device layout?

The memory side moves whole 128-B lines (profiles/r2_fetch_calibration.csv) and the kernel's step gathers one 64-B record, so the
order of the records in HBM decides how many line fills a ray costs (config M: 12.8 per ray, profiles/r2_pmc_summary.csv).  This
script logs, with the oracle (orc_visit_log: test infrastructure, which is why the script lives under tests/), the sequence of BVH
records a sample of the frame's rays touch -- inner steps by child pair, triangle tests, the winning triangle's attribute fetch --
and replays it against candidate layouts with a small cache model:

  * a record access hits if the same ray touched the same line within its last W record accesses (the L2 keeps a line for a few
    traversal steps: 4 MiB per XCD turn over in ~8 us at 33 G fills/s, a step is ~0.9 us);
  * the H most frequently touched lines are always hits (the top of the tree lives in L2).

W and H are calibrated so that the CURRENT layout reproduces the measured 12.8 fills per ray; the model is then used to RANK
layouts, and the winner is measured on the GPU (TCC_EA0_RDREQ_sum).

    python tests/tools/layout_model.py [--tris N] [--pixels P]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rust_ray_tracing_amd as rrt  # noqa: E402
from oracle import orc  # noqa: E402
from rust_ray_tracing_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=10_000_000)
ap.add_argument("--pixels", type=int, default=1500)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--depth", type=int, default=64)
ap.add_argument("--cache", default="/tmp/layout_model_cache.npz")
args = ap.parse_args()

W_, H_ = 1920, 1080


def get_trace():
    key = f"{args.tris}_{args.pixels}_{args.spp}_{args.depth}"
    if os.path.exists(args.cache):
        z = np.load(args.cache)
        if str(z["key"]) == key:
            return z["nodes"].view(rrt.NODE), z["log"]
    t0 = time.time()
    tris, mats, texs, cam = synth.atrium_scene(n_target=args.tris, tex_size=64)
    sc = rrt.Scene.from_arrays(tris, mats, texs)
    sc.set_camera(rrt.Camera(position=cam[0], pitch=cam[1], yaw=cam[2]))
    print(f"scene {len(sc.tris)} tris, {len(sc.bvh_nodes)} nodes in {time.time() - t0:.1f}s", flush=True)
    lib = orc.load()
    lib.orc_visit_log.restype = C.c_uint64
    lib.orc_visit_log.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(orc.OrcTexture), C.c_uint32,
                                  C.c_void_p, C.POINTER(orc.OrcOptions), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
    opt = orc.OrcOptions(W_, H_, args.spp, args.depth, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0.0078125, 0)
    cap = args.pixels * args.spp * 16 * 64 + (1 << 20)
    log = np.zeros(cap, dtype=np.uint32)
    stride = (W_ * H_) // args.pixels | 1
    mats_arr = sc.materials_array()
    texs_c = orc._tex_array(sc.textures)
    t0 = time.time()
    n = lib.orc_visit_log(sc.tris.ctypes.data, len(sc.tris), sc.bvh_nodes.ctypes.data, len(sc.bvh_nodes), mats_arr.ctypes.data, len(mats_arr),
                          texs_c, len(sc.textures), sc.camera.uniform.ctypes.data, C.byref(opt), 0, stride, args.pixels, log.ctypes.data, cap)
    assert n <= cap, (n, cap)
    print(f"logged {n} words in {time.time() - t0:.1f}s", flush=True)
    log = log[:n].copy()
    np.savez(args.cache, key=key, nodes=sc.bvh_nodes.view(np.uint8), log=log)
    return sc.bvh_nodes, log


nodes, log = get_trace()
n_nodes = len(nodes)
n_pairs = (n_nodes - 1) // 2
num_tris = nodes["num_tris"].astype(np.int64)
first = nodes["first_tri_or_child"].astype(np.int64)
n_tris = int((first + num_tris)[num_tris > 0].max())

# ---- the access stream ----
is_start = log == 0xFFFFFFFF
ray_id = np.cumsum(is_start) - 1
acc = ~is_start
kind = (log >> 30).astype(np.int64)            # 0/1 = pair, 2 = tri test, 3 = attr
kind[kind == 1] = 0
idx = (log & 0x3FFFFFFF).astype(np.int64)
ray_of = ray_id[acc]
kind = kind[acc]
idx = idx[acc]
n_rays = int(is_start.sum())
pos = np.arange(len(idx))
print(f"{n_rays} rays; per ray: {np.sum(kind == 0) / n_rays:.2f} inner steps, {np.sum(kind == 2) / n_rays:.2f} triangle tests, "
      f"{np.sum(kind == 3) / n_rays:.2f} attribute fetches", flush=True)

# ---- tree facts ----
l_child = 2 * np.arange(n_pairs) + 1
r_child = l_child + 1
leafL, leafR = num_tris[l_child] > 0, num_tris[r_child] > 0
print(f"pairs {n_pairs}: LL {np.mean(leafL & leafR):.3f}  LI {np.mean(leafL ^ leafR):.3f}  II {np.mean(~leafL & ~leafR):.3f};  "
      f"leaves with 1 / 2 / 3+ tris: {np.mean(num_tris[num_tris > 0] == 1):.3f} / {np.mean(num_tris[num_tris > 0] == 2):.3f} / {np.mean(num_tris[num_tris > 0] > 2):.3f}")


def pair_order_product():
    """record index -> pair (0xffffffff = pad): the product's breadth-first couples order (bvh_build.cpp)."""
    fn = rrt.load_diag().mipt_internal_pair_order
    out = np.zeros(2 * n_pairs + 2, dtype=np.uint32)
    n = C.c_uint32(0)
    assert fn(nodes.ctypes.data, n_nodes, out.ctypes.data, out.size, C.byref(n)) == 0
    return out[: n.value]


order = pair_order_product()
rec_of_pair = np.zeros(n_pairs, dtype=np.int64)
real = order != 0xFFFFFFFF
rec_of_pair[order[real]] = np.flatnonzero(real)
n_pair_recs = len(order)
# parent pair / sibling facts
child_pair = np.full(n_nodes, -1, dtype=np.int64)
inner = num_tris == 0
child_pair[inner] = (first[inner] - 1) // 2
parent_node_of_pair = np.zeros(n_pairs, dtype=np.int64)        # node whose children are this pair
parent_node_of_pair[child_pair[inner]] = np.flatnonzero(inner)
# a pair is "single" if its parent node's sibling is a leaf (so it has no sibling pair)
pn = parent_node_of_pair
sib = np.where(pn == 0, 0, np.where(pn % 2 == 1, pn + 1, pn - 1))
is_single = (pn != 0) & (num_tris[sib] > 0)
print(f"single pairs (no sibling pair): {is_single.mean():.3f}")
# leaf -> (pair, which)
leaf_nodes = np.flatnonzero(num_tris > 0)
leaf_of_tri = np.zeros(n_tris, dtype=np.int64)
for nd_chunk in np.array_split(leaf_nodes, 64):
    reps = num_tris[nd_chunk]
    starts = first[nd_chunk]
    ids = np.repeat(nd_chunk, reps)
    offs = np.arange(reps.sum()) - np.repeat(np.cumsum(reps) - reps, reps)
    leaf_of_tri[np.repeat(starts, reps) + offs] = ids
pair_of_leaf = (leaf_nodes - 1) // 2


def simulate(name, pair_line, tri_line, attr_line, W, H, verbose=True):
    line = np.empty(len(idx), dtype=np.int64)
    for kk, arr in ((0, pair_line), (2, tri_line), (3, attr_line)):
        m = kind == kk
        line[m] = arr[idx[m]]
    # hot set: the H most frequently touched lines
    ul, inv, cnt = np.unique(line, return_inverse=True, return_counts=True)
    hot = np.zeros(len(ul), dtype=bool)
    if H > 0:
        hot[np.argsort(-cnt)[:H]] = True
    is_hot = hot[inv]
    # previous access to the same line by the same ray
    o = np.lexsort((pos, inv, ray_of))
    same = (ray_of[o][1:] == ray_of[o][:-1]) & (inv[o][1:] == inv[o][:-1])
    dist = np.full(len(line), 1 << 40, dtype=np.int64)
    d = pos[o][1:] - pos[o][:-1]
    dist[o[1:][same]] = d[same]
    hit = is_hot | (dist <= W)
    miss = ~hit
    res = {"total": miss.sum() / n_rays, "pair": (miss & (kind == 0)).sum() / n_rays, "tri": (miss & (kind == 2)).sum() / n_rays,
           "attr": (miss & (kind == 3)).sum() / n_rays, "lines": len(ul)}
    if verbose:
        print(f"{name:46s} W={W:3d} H={H:6d}: fills/ray {res['total']:6.2f}  (pair {res['pair']:5.2f}  tri {res['tri']:5.2f}  attr {res['attr']:5.2f})", flush=True)
    return res


# ---------------- layouts: each returns line ids (disjoint ranges for the three regions) ----------------
PAIR0, TRI0, ATTR0 = 0, 1 << 32, 1 << 33


def layout_current():
    return PAIR0 + rec_of_pair // 2, TRI0 + np.arange(n_tris) // 2, ATTR0 + np.arange(n_tris) // 2


def layout_depth_first():
    return PAIR0 + np.arange(n_pairs) // 2, TRI0 + np.arange(n_tris) // 2, ATTR0 + np.arange(n_tris) // 2


def tri_region_aligned():
    """triangle records re-packed so that a leaf pair's run of triangles (both leaf children of one pair, or one leaf) does not
    straddle a line when it would fit in one: returns slot index per triangle (slot // 2 = line)."""
    slot = np.zeros(n_tris, dtype=np.int64)
    # runs: per pair, the leaf children's triangles are consecutive [a, a + nL + nR) when both are leaves; else single leaf run
    both = leafL & leafR
    run_start = np.concatenate([first[l_child[both]], first[l_child[leafL & ~leafR]], first[r_child[leafR & ~leafL]]])
    run_len = np.concatenate([num_tris[l_child[both]] + num_tris[r_child[both]], num_tris[l_child[leafL & ~leafR]], num_tris[r_child[leafR & ~leafL]]])
    o = np.argsort(run_start)
    run_start, run_len = run_start[o], run_len[o]
    assert run_start[0] == 0 and np.all(run_start[1:] == run_start[:-1] + run_len[:-1])
    # sequential packing with "pad one slot if the run has 2 triangles and would start on an odd slot"
    cur = 0
    starts = np.zeros(len(run_len), dtype=np.int64)
    # vectorised: process in python is 7 M iterations -- too slow; closed form: pad needed depends on the running parity
    parity = 0
    pads = np.zeros(len(run_len), dtype=np.int64)
    rl = run_len.tolist()
    p_list = []
    for n in rl:
        p = 1 if (n == 2 and (cur & 1)) else 0
        p_list.append(p)
        cur += p + n
    pads = np.array(p_list, dtype=np.int64)
    starts = np.cumsum(pads + run_len) - run_len
    slot = np.repeat(starts, run_len) + (np.arange(run_len.sum()) - np.repeat(np.cumsum(run_len) - run_len, run_len))
    print(f"  aligned triangle stream: {pads.sum()} pad slots for {n_tris} triangles")
    return slot


def layout_aligned_tris():
    slot = tri_region_aligned()
    return PAIR0 + rec_of_pair // 2, TRI0 + slot // 2, ATTR0 + np.arange(n_tris) // 2


def inline_candidates():
    """per pair: triangle index of a single-triangle leaf child to ride in the pair's line (-1 = none); prefers the left child."""
    t = np.full(n_pairs, -1, dtype=np.int64)
    r1 = leafR & (num_tris[r_child] == 1)
    t[r1] = first[r_child[r1]]
    l1 = leafL & (num_tris[l_child] == 1)
    t[l1] = first[l_child[l1]]
    return t


def layout_inline(which):
    """which = 'singles': only pairs without a sibling pair take their leaf's triangle into their line;
    'all': every pair with a single-triangle leaf child does (couples are broken up for them)."""
    cand = inline_candidates()
    take = (cand >= 0) & (is_single if which == "singles" else True)
    if which == "all_deep":
        pass
    # pair lines: pairs with an inline triangle get a line of their own; the others keep the product order among themselves
    pair_line = np.zeros(n_pairs, dtype=np.int64)
    keep = ~take
    # records of the kept pairs in product order, re-compacted (couples stay adjacent if BOTH are kept, else the partner is a single)
    kept_order = order[real][keep[order[real]]]
    # re-pack: walk product order; couples were (even, odd) record neighbours
    rec = rec_of_pair
    mate = np.full(n_pairs, -1, dtype=np.int64)
    even = np.flatnonzero(real[0::2] & np.pad(real[1::2], (0, len(real[0::2]) - len(real[1::2])), constant_values=False)) * 2
    a, b = order[even].astype(np.int64), order[even + 1].astype(np.int64)
    mate[a], mate[b] = b, a
    coupled_kept = keep & (mate >= 0) & keep[np.maximum(mate, 0)]
    lone_kept = keep & ~coupled_kept
    n_lines = 0
    # coupled kept pairs: one line per couple
    ck = np.flatnonzero(coupled_kept)
    first_of_couple = ck[rec[ck] % 2 == 0]
    pair_line[first_of_couple] = np.arange(len(first_of_couple))
    pair_line[mate[first_of_couple]] = np.arange(len(first_of_couple))
    n_lines += len(first_of_couple)
    lk = np.flatnonzero(lone_kept)
    lk = lk[np.argsort(rec[lk])]
    pair_line[lk] = n_lines + np.arange(len(lk)) // 2
    n_lines += (len(lk) + 1) // 2
    tk = np.flatnonzero(take)
    pair_line[tk] = n_lines + np.arange(len(tk))
    tri_line = TRI0 + np.arange(n_tris) // 2
    tri_line = tri_line.copy()
    tri_line[cand[tk]] = PAIR0 + pair_line[tk]
    print(f"  inline[{which}]: {len(tk)} pairs carry a triangle ({len(tk) / n_pairs:.3f} of pairs, {len(tk) / n_tris:.3f} of triangles); pair region {n_lines + len(tk)} lines (was {n_pair_recs // 2})")
    return PAIR0 + pair_line, tri_line, ATTR0 + np.arange(n_tris) // 2


def layout_tri_with_attr():
    """[tri_pos | tri_attr] of one triangle in one line."""
    return PAIR0 + rec_of_pair // 2, TRI0 + np.arange(n_tris), TRI0 + np.arange(n_tris)


def layout_parent_child():
    """pair + its left inner child's pair in one line (depth-first order gives exactly that for the left child): here a greedy
    chain decomposition in DFS order = layout_depth_first; kept for reference."""
    return layout_depth_first()



def surface_half_area(node_idx):
    lo = nodes["bounds_min"][node_idx].astype(np.float64)
    hi = nodes["bounds_max"][node_idx].astype(np.float64)
    e = hi - lo
    return e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]


def bfs_levels():
    """list of arrays of pair indices per depth (root's children pair = level 0)."""
    levels = [np.array([0], dtype=np.int64)]
    while True:
        k = levels[-1]
        kids = np.concatenate([child_pair[2 * k + 1], child_pair[2 * k + 2]])
        kids = kids[kids >= 0]
        if len(kids) == 0:
            break
        levels.append(kids)
    return levels


LEVELS = bfs_levels()
depth_of_pair = np.zeros(n_pairs, dtype=np.int64)
for dd, lv in enumerate(LEVELS):
    depth_of_pair[lv] = dd
print(f"tree depth {len(LEVELS)} levels; widest level {max(len(l) for l in LEVELS)}")


def layout_parent_child(weight="area", inline_tri=False, start_parity=0, tri_lines=None):
    """Greedy top-down matching: an unmatched pair takes its preferred unmatched inner child's pair as line mate; a pair that was
    taken by its parent leaves its own children to start new lines.  weight: 'area' = the child with the larger box (SAH's own
    hit-probability proxy), 'visits' = the child pair the trace visits more often (upper bound, not available to a builder).
    inline_tri: an unmatched pair with no inner child takes the first triangle of a leaf child into its line."""
    vis = np.bincount(idx[kind == 0], minlength=n_pairs).astype(np.float64)
    mate = np.full(n_pairs, -1, dtype=np.int64)
    taken = np.zeros(n_pairs, dtype=bool)
    tri_mate = np.full(n_pairs, -1, dtype=np.int64)
    for dd, lv in enumerate(LEVELS):
        free = lv[~taken[lv]]
        if dd < start_parity:
            continue
        cl, cr = child_pair[2 * free + 1], child_pair[2 * free + 2]
        if weight == "area":
            wl, wr = surface_half_area(2 * free + 1), surface_half_area(2 * free + 2)
        else:
            wl = np.where(cl >= 0, vis[np.maximum(cl, 0)], -1.0)
            wr = np.where(cr >= 0, vis[np.maximum(cr, 0)], -1.0)
        wl = np.where(cl >= 0, wl, -1.0)
        wr = np.where(cr >= 0, wr, -1.0)
        pick = np.where(wl >= wr, cl, cr)
        ok = pick >= 0
        mate[free[ok]] = pick[ok]
        mate[pick[ok]] = free[ok]
        taken[pick[ok]] = True
        if inline_tri:
            lone = free[~ok]                                   # both children are leaves
            tri_mate[lone] = first[2 * lone + 1]               # the left leaf's first triangle
    # line ids: matched couples share one; lone pairs: own line if they carry a triangle, else packed two by two
    pair_line = np.full(n_pairs, -1, dtype=np.int64)
    heads = np.flatnonzero((mate >= 0) & (depth_of_pair < depth_of_pair[np.maximum(mate, 0)]))
    pair_line[heads] = np.arange(len(heads))
    pair_line[mate[heads]] = np.arange(len(heads))
    n_lines = len(heads)
    withtri = np.flatnonzero((mate < 0) & (tri_mate >= 0))
    pair_line[withtri] = n_lines + np.arange(len(withtri))
    n_lines += len(withtri)
    lone = np.flatnonzero((mate < 0) & (tri_mate < 0))
    pair_line[lone] = n_lines + np.arange(len(lone)) // 2
    n_lines += (len(lone) + 1) // 2
    tl = (TRI0 + np.arange(n_tris) // 2) if tri_lines is None else tri_lines.copy()
    if inline_tri:
        tl = tl.copy()
        tl[tri_mate[withtri]] = PAIR0 + pair_line[withtri]
    print(f"  parent-child[{weight}, inline_tri={inline_tri}]: {len(heads)} parent+child lines, {len(withtri)} pair+triangle lines, {len(lone)} lone pairs; "
          f"pair region {n_lines} lines (now {n_pair_recs // 2})")
    return PAIR0 + pair_line, tl, ATTR0 + np.arange(n_tris) // 2


W0, H0 = 8, 8192
print(f"--- candidates at W={W0}, H={H0} (the calibration that reproduces the measured 12.8 fills per ray on the current layout) ---")
simulate("current: BFS couples, tris 2 per line", *layout_current(), W0, H0)
simulate("reference order (depth-first)", *layout_depth_first(), W0, H0)
simulate("current + aligned triangle runs", *layout_aligned_tris(), W0, H0)
simulate("inline triangle for single pairs", *layout_inline("singles"), W0, H0)
simulate("inline triangle for every pair that can", *layout_inline("all"), W0, H0)
simulate("[tri_pos | tri_attr] per triangle", *layout_tri_with_attr(), W0, H0)
simulate("parent+child by box area", *layout_parent_child("area"), W0, H0)
simulate("parent+child by trace visits (upper bound)", *layout_parent_child("visits"), W0, H0)
simulate("parent+child by area, other level parity", *layout_parent_child("area", start_parity=1), W0, H0)
simulate("parent+child by area + LL pair | triangle", *layout_parent_child("area", inline_tri=True), W0, H0)
slot = tri_region_aligned()
simulate("parent+child by area + aligned triangle runs", *layout_parent_child("area", tri_lines=TRI0 + slot // 2), W0, H0)
for H in (4096, 16384):
    simulate("current", *layout_current(), W0, H)
    simulate("parent+child by box area", *layout_parent_child("area"), W0, H)

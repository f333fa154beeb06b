"""CPU: the order of the device's pair records (csrc/bvh_build.cpp mipt_internal_pair_order, used by mipt_scene_create) --
breadth-first with sibling pairs in one 128-B line.  Topology is untouched: the function only permutes record positions."""
import ctypes as C

import numpy as np
import pytest


def _order(rrt, nodes):
    lib = rrt.load()
    fn = lib.mipt_internal_pair_order
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    fn.restype = C.c_int
    n_pairs = (len(nodes) - 1) // 2
    out = np.zeros(2 * n_pairs + 2, dtype=np.uint32)
    n = C.c_uint32(0)
    assert fn(nodes.ctypes.data, len(nodes), out.ctypes.data, out.size, C.byref(n)) == 0
    return out[: n.value]


def _check(nodes, order):
    n_pairs = (len(nodes) - 1) // 2
    real = order[order != 0xFFFFFFFF]
    assert sorted(real.tolist()) == list(range(n_pairs))                      # a permutation of all pairs ...
    if n_pairs:
        assert order[0] == 0                                                  # ... with the root's children first
    pos = np.zeros(n_pairs, dtype=np.int64)
    pos[real] = np.flatnonzero(order != 0xFFFFFFFF)
    couples = 0
    for k in range(n_pairs):
        kids = []
        for w in range(2):
            n = nodes[2 * k + 1 + w]
            if n["num_tris"] == 0:
                kids.append((int(n["first_tri_or_child"]) - 1) // 2)
        for c in kids:
            assert pos[c] > pos[k]                                            # children after parents (breadth-first)
        if len(kids) == 2:                                                    # the two child pairs of a node share one 128-B line
            assert pos[kids[1]] == pos[kids[0]] + 1 and pos[kids[0]] % 2 == 0
            couples += 1
    assert int((order == 0xFFFFFFFF).sum()) <= max(1, n_pairs)                # pads: at most one per level
    return couples


@pytest.mark.parametrize("kind,kw", [("cornell", {}), ("helmet", dict(n_target=3000, tex_size=8)), ("atrium", dict(n_target=20000, tex_size=8))])
def test_pair_order_of_real_trees(rrt, kind, kw):
    from rust_ray_tracing_amd import synth
    tris = synth.make_scene(kind, **kw)[0]
    sc = rrt.Scene.from_arrays(tris, [rrt.material_default()])
    couples = _check(sc.bvh_nodes, _order(rrt, sc.bvh_nodes))
    assert couples > 0 or len(sc.bvh_nodes) <= 3


def test_pair_order_of_degenerate_trees(rrt):
    from rust_ray_tracing_amd import NODE
    one = np.zeros(1, dtype=NODE); one["num_tris"] = 3                        # root leaf: no pairs at all
    assert len(_order(rrt, one)) == 0
    # a left-leaning chain of depth 40: every level holds one pair, so every second record is a pad
    depth = 40
    nodes = np.zeros(2 * depth + 1, dtype=NODE)
    for d in range(depth):
        i = 0 if d == 0 else 2 * d - 1                                        # the inner node of level d (root, then always the left child)
        nodes[i]["num_tris"] = 0; nodes[i]["first_tri_or_child"] = 2 * d + 1
        nodes[2 * d + 2]["num_tris"] = 1; nodes[2 * d + 2]["first_tri_or_child"] = d
    nodes[2 * depth - 1]["num_tris"] = 1; nodes[2 * depth - 1]["first_tri_or_child"] = depth
    order = _order(rrt, nodes)
    _check(nodes, order)
    assert len(order) == 2 * depth - 1 and (order[1::2] == 0xFFFFFFFF).all()


def test_shared_child_bvh_is_refused_quickly(rrt):
    """A DAG handed in as a BVH (both children of pair k pointing at pair k+1): the breadth-first order would double per level.
    mipt_scene_create must return MIPT_ERR_BVH from its validation loop, and the order helper must stop at its cap inside the loop
    (the old code needed 4 s and 1.2 GB for 27 pairs)."""
    import time
    from rust_ray_tracing_amd import NODE, TRIANGLE
    from rust_ray_tracing_amd import _lib as L
    n_pairs = 41
    nodes = np.zeros(2 * n_pairs + 1, dtype=NODE)
    nodes["bounds_max"] = 1.0
    nodes[0]["first_tri_or_child"] = 1
    for k in range(n_pairs - 1):
        for w in range(2):
            nodes[2 * k + 1 + w]["first_tri_or_child"] = 2 * (k + 1) + 1
    for w in range(2):
        nodes[2 * (n_pairs - 1) + 1 + w]["num_tris"] = 1
    lib = rrt.load()
    fn = lib.mipt_internal_pair_order
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    fn.restype = C.c_int
    out = np.zeros(2 * n_pairs + 2, dtype=np.uint32)
    n = C.c_uint32(0)
    t0 = time.time()
    assert fn(nodes.ctypes.data, len(nodes), out.ctypes.data, out.size, C.byref(n)) == L.ERR_SCENE_LIMIT
    tris = np.zeros(1, dtype=TRIANGLE)
    sc = rrt.Scene()
    sc.tris, sc.bvh_nodes, sc.materials = tris, nodes, {"m": rrt.material_default()}
    h = C.c_void_p()
    d = sc.desc()
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_BVH
    assert b"more than one inner node" in lib.mipt_last_error()
    assert time.time() - t0 < 1.0
    # an orphan pair (never referenced) is refused as well
    orphan = np.zeros(5, dtype=NODE)
    orphan["bounds_max"] = 1.0
    orphan[0]["first_tri_or_child"] = 1
    for i in (1, 2, 3, 4):
        orphan[i]["num_tris"] = 1
    sc.bvh_nodes = orphan
    d = sc.desc()
    assert lib.mipt_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_BVH
    assert b"not the children" in lib.mipt_last_error()

"""GPU box: per tree level, how many nodes of each builder class are SPLIT there (the work lists the device builder runs level by
level), derived from the finished tree -- the sizes the level kernels' cost model in bvh_build_device.hip was fitted on.
Classes by triangle count: SUB <= 4 (whole subtree in one thread: only its root is listed), TINY 5..8, G16 9..16, G32 17..32,
WAVE 33..64, WAVE_A 65..128, WAVE_M 129..512, WAVE_L 513..2048, BIG > 2048."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_ray_tracing_amd as rrt
from rust_ray_tracing_amd import synth

tris = synth.make_scene("atrium", n_target=int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, tex_size=16)[0]
b = rrt.Scene.from_arrays(tris, [rrt.material_default()], build_bvh=False)
b.build_bvh_device(0)
nodes = b.bvh_nodes
first = nodes["first_tri_or_child"].astype(np.int64)
ntri = nodes["num_tris"].astype(np.int64)
levels = [np.array([0], dtype=np.int64)]
while True:
    cur = levels[-1]
    inner = cur[ntri[cur] == 0]
    if len(inner) == 0:
        break
    levels.append(np.concatenate([first[inner], first[inner] + 1]))
size = ntri.copy()                                                  # triangles under every node, bottom-up
for lv in reversed(levels[:-1]):
    inner = lv[ntri[lv] == 0]
    size[inner] = size[first[inner]] + size[first[inner] + 1]
edges = [0, 4, 8, 16, 32, 64, 128, 512, 2048, 1 << 62]
names = ["SUB", "TINY", "G16", "G32", "WAVE", "WAVE_A", "WAVE_M", "WAVE_L", "BIG"]
print("level " + " ".join(f"{n:>9s}" for n in names) + "   (nodes handed to a level kernel; a SUB root's descendants are not listed again)")
in_sub = np.zeros(len(nodes), dtype=bool)                           # below a SUB root: finished by that thread
for d, lv in enumerate(levels):
    lv = lv[~in_sub[lv]]
    cand = lv[(size[lv] > 1) | (ntri[lv] == 0)]                     # the builder tries to split every node with > 1 triangle ... leaves with n > 1 were tried too
    cls = np.searchsorted(edges, size[cand], side="left") - 1
    cnt = np.bincount(cls, minlength=len(names))
    tri = np.bincount(cls, weights=size[cand], minlength=len(names)).astype(np.int64)
    print(f"{d:5d} " + " ".join(f"{c:9d}" for c in cnt) + "   tris " + " ".join(f"{t:9d}" for t in tri))
    sub_roots = cand[size[cand] <= 4]
    inner = sub_roots[ntri[sub_roots] == 0]
    while len(inner):                                               # mark everything under a SUB root
        kids = np.concatenate([first[inner], first[inner] + 1])
        in_sub[kids] = True
        inner = kids[ntri[kids] == 0]

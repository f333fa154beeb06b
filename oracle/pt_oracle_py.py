"""Second, independent CPU restatement of the reference's rayon path tracer, BVH builder and OBJ/MTL flatten -- TEST INFRASTRUCTURE ONLY.

Written from the Rust sources again, in pure Python over numpy float32 scalars (one rounded IEEE operation per Rust
operator), for SMALL cases only (a few hundred pixels): tests/test_oracle_second_reading.py requires it to agree bit for bit
with the C oracle (pt_oracle.c), so that a misreading of the reference in one restatement does not go unnoticed.  Like the
C oracle it is "parity unpinned" against the real Rust binary (no Rust toolchain here, the reference has no tests or
fixtures); cos / log10 are calls into the platform libm (glibc), exactly as in the Rust binary.

Reference lines followed: src/renderer/backend/cpu.rs:13-68, src/renderer/backend/cpu/ray.rs:19-227, src/math.rs:6-24,
src/math/vec3.rs:66-109,130-205,252-366, src/math/mat4.rs:143-152, src/texture.rs:33-38, src/bvh.rs:13-203,
src/scene.rs:44-85,114-167, src/loader/obj.rs:16-436 (geometry and scalar material keys; texture maps are not followed).
"""
import math
import struct

import numpy as np

F = np.float32
U32 = 0xFFFFFFFF
F32_MAX = F(3.4028234663852886e38)
MISS = F(1e30)


# ---- cos / log10: the platform libm itself -----------------------------------------------------------------------------
# Rust std's f32::cos / f32::log10 (math.rs:16-18) are calls into the platform libm; this reading makes the same calls
# (glibc's cosf / log10f through ctypes) instead of restating them, so agreement with the C oracle -- which evaluates its
# glibc 2.35 restatement, oracle/glibc_flt32.h -- also checks that restatement on every argument these renders produce.
import ctypes as _C

_libm = _C.CDLL("libm.so.6")
_libm.cosf.restype = _C.c_float
_libm.cosf.argtypes = [_C.c_float]
_libm.log10f.restype = _C.c_float
_libm.log10f.argtypes = [_C.c_float]


def libm_cosf(x):
    return F(_libm.cosf(float(F(x))))


def libm_log10f(x):
    return F(_libm.log10f(float(F(x))))


# pow is only needed by the sRGB epilogue; it is taken from the C oracle's known-answer entry point in the test instead of
# being restated a third time (the test compares radiance, which does not involve it).


# ---- math.rs / vec3.rs ------------------------------------------------------------------------------------------------
def xor_shift(s):                                           # math.rs:6-13
    x = s[0]
    x ^= (x << 13) & U32
    x ^= x >> 17
    x ^= (x << 5) & U32
    s[0] = x
    return x


def rand_f32(s):                                            # math.rs:22-24 (u32::MAX as f32 == 2^32)
    return F(xor_shift(s)) / F(4294967296.0)


def rand_f32_nd(s):                                         # math.rs:15-19
    theta = F(6.283185) * rand_f32(s)
    rho = np.sqrt(F(-2.0) * libm_log10f(rand_f32(s)))
    return F(rho * libm_cosf(theta))


def v_add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def v_sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def v_mul(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
def v_scale(a, k): return (a[0] * k, a[1] * k, a[2] * k)
def v_div(a, k): return (a[0] / k, a[1] / k, a[2] / k)
def dot(a, b): return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2])                       # vec3.rs:130-134
def cross(a, b):                                                                            # vec3.rs:136-144
    return ((a[1] * b[2]) - (a[2] * b[1]), (a[2] * b[0]) - (a[0] * b[2]), (a[0] * b[1]) - (a[1] * b[0]))
def length(a): return np.sqrt((a[0] * a[0]) + (a[1] * a[1]) + (a[2] * a[2]))               # vec3.rs:93-97
def normalized(a): return v_div(a, length(a))                                              # vec3.rs:105-109
def vec(p): return (F(p[0]), F(p[1]), F(p[2]))


def rand_in_unit_sphere(s):                                 # vec3.rs:66-68: x, y, z drawn in that order
    x = rand_f32_nd(s)
    y = rand_f32_nd(s)
    z = rand_f32_nd(s)
    return normalized((x, y, z))


def fmin(a, b): return F(np.fmin(a, b))                     # f32::min / max: a NaN operand is dropped
def fmax(a, b): return F(np.fmax(a, b))


# ---- ray.rs -----------------------------------------------------------------------------------------------------------
def intersect_tri(o, d, tri):                               # ray.rs:19-67
    v1, v2, v3 = (vec(tri["vertices"][k]["position"]) for k in range(3))
    e1, e2 = v_sub(v2, v1), v_sub(v3, v1)
    rce2 = cross(d, e2)
    det = dot(e1, rce2)
    inv = F(1.0) / det
    s = v_sub(o, v1)
    u = inv * dot(s, rce2)
    sce1 = cross(s, e1)
    v = inv * dot(d, sce1)
    t = inv * dot(e2, sce1)
    front = bool(det > 0)
    n0, n1, n2 = (vec(tri["vertices"][k]["normal"]) for k in range(3))
    w = F(1.0) - u - v
    normal = v_add(v_add(v_scale(n0, w), v_scale(n1, u)), v_scale(n2, v))
    if not front:
        normal = (-normal[0], -normal[1], -normal[2])
    t0, t1, t2 = ((F(tri["vertices"][k]["tex_coord_x"]), F(tri["vertices"][k]["tex_coord_y"])) for k in range(3))
    uv = ((t0[0] * w + t1[0] * u) + t2[0] * v, (t0[1] * w + t1[1] * u) + t2[1] * v)
    has_hit = bool(t > 0) and not (det < 0 and det > F(-0.0)) and not (u < 0 or u > 1) and not (v < 0 or u + v > 1)
    return dict(has_hit=has_hit, point=v_add(o, v_scale(d, t)), normal=normal, distance=t, uv=uv,
                material_id=int(tri["material_id"]), front_face=front)


def intersect_node(o, d, node):                             # ray.rs:69-81
    lo, hi = vec(node["bounds_min"]), vec(node["bounds_max"])
    tmin = tuple((lo[k] - o[k]) / d[k] for k in range(3))
    tmax = tuple((hi[k] - o[k]) / d[k] for k in range(3))
    t1 = tuple(fmin(tmin[k], tmax[k]) for k in range(3))
    t2 = tuple(fmax(tmin[k], tmax[k]) for k in range(3))
    t_near = fmax(fmax(t1[0], t1[1]), t1[2])
    t_far = fmin(fmin(t2[0], t2[1]), t2[2])
    return t_near if (t_near <= t_far and t_far > 0) else MISS


def traverse_bvh(o, d, nodes, tris, hit, counters):         # ray.rs:84-139
    stack = []
    node = nodes[0]
    while True:
        if node["num_tris"] > 0:
            for i in range(int(node["num_tris"])):
                counters["tri_tests"] += 1
                h = intersect_tri(o, d, tris[int(node["first_tri_or_child"]) + i])
                if h["has_hit"] and h["distance"] < hit["distance"]:
                    hit.clear()
                    hit.update(h)
            if not stack:
                break
            node = stack.pop()
            continue
        counters["inner_steps"] += 1
        c1 = nodes[int(node["first_tri_or_child"])]
        c2 = nodes[int(node["first_tri_or_child"]) + 1]
        d1, d2 = intersect_node(o, d, c1), intersect_node(o, d, c2)
        if d1 > d2:
            d1, d2 = d2, d1
            c1, c2 = c2, c1
        if d1 == MISS:
            if not stack:
                break
            node = stack.pop()
        else:
            node = c1
            if d2 < MISS:
                if len(stack) >= 32:
                    raise OverflowError("traversal stack beyond 32 entries: the reference panics (ray.rs:85)")
                stack.append(c2)


def color_at(tex, uv):                                      # texture.rs:33-38 (tex: (h, w, 4) u8, rows as stored)
    h, w = tex.shape[:2]
    def fract(x): return x - np.trunc(x)
    def as_i32(x):                                          # Rust `as i32`: saturating, NaN -> 0
        if x != x: return 0
        return int(max(-2147483648.0, min(2147483647.0, float(np.trunc(x)))))
    i = as_i32(fract(uv[0]) * F(w))
    j = as_i32(fract(uv[1]) * F(h))
    idx = i + j * w
    if idx < 0 or idx >= w * h:
        raise IndexError("texture index out of range: the reference panics here (SURVEY T10)")
    px = tex.reshape(-1, 4)[idx]
    return (F(px[0]) / F(255.0), F(px[1]) / F(255.0), F(px[2]) / F(255.0))


def trace(o, d, max_bounces, nodes, tris, materials, textures, rng, counters):   # ray.rs:141-202
    one = F(1.0)
    ray_color, incoming, emitted = (one, one, one), (F(0), F(0), F(0)), (F(0), F(0), F(0))
    bounces = 0
    while bounces < max_bounces:
        hit = dict(has_hit=False, distance=MISS)
        counters["rays"] += 1
        traverse_bvh(o, d, nodes, tris, hit, counters)
        if hit["has_hit"]:
            m = materials[hit["material_id"]]
            if int(m["base_color_tex_id"]) != U32:
                ray_color = v_mul(ray_color, color_at(textures[int(m["base_color_tex_id"])], hit["uv"]))
            else:
                ray_color = v_mul(ray_color, vec(m["base_color"]))
            if int(m["emission_tex_id"]) != U32:
                emitted = v_add(emitted, color_at(textures[int(m["emission_tex_id"])], hit["uv"]))
            else:
                emitted = v_add(emitted, vec(m["emission"]))
            incoming = v_add(incoming, v_mul(emitted, ray_color))
            new_dir = normalized(v_add(hit["normal"], rand_in_unit_sphere(rng)))
            o = v_add(hit["point"], v_scale(new_dir, F(0.0001)))
            d = new_dir
            bounces += 1
        else:
            ray_color = v_mul(ray_color, (one, one, one))
            emitted = v_add(emitted, (one, one, one))
            incoming = v_add(incoming, v_mul(emitted, ray_color))
            break
    return incoming if bounces == 0 else v_div(incoming, F(bounces))


def render(tris, nodes, materials, textures, camera, width, height, samples, max_ray_depth, pixels=None):
    """cpu.rs:13-68 for the pixel indices in `pixels` (default: all).  Returns ({index: (r, g, b) float32 mean radiance},
    counters); the sRGB / quantisation epilogue is left to the caller."""
    look = np.asarray(camera["look_at"], dtype=np.float32).reshape(4, 4)
    pos = vec(np.asarray(camera["position"], dtype=np.float32).reshape(-1)[:3])
    out, counters = {}, dict(rays=0, inner_steps=0, tri_tests=0)
    w, h = width, height
    with np.errstate(all="ignore"):
        for index in (range(w * h) if pixels is None else pixels):
            rng = [(987612486 * ((index + 87636354) & U32)) & U32]                                  # cpu.rs:28-29
            final = (F(0), F(0), F(0))
            x, y = index % w, h - (index // w)                                                      # cpu.rs:31-32
            sx = (((F(x) / F(w)) * F(2.0)) - F(1.0)) * (F(w) / F(h))
            sy = ((F(y) / F(h)) * F(2.0)) - F(1.0)
            for _ in range(samples):
                jx = (rand_f32(rng) * F(2.0) - F(1.0)) * F(0.0005)
                jy = (rand_f32(rng) * F(2.0) - F(1.0)) * F(0.0005)
                r = (-sx + jx, sy + jy, F(1.0))
                d = tuple(look[0][k] * r[0] + look[1][k] * r[1] + look[2][k] * r[2] for k in range(3))   # mat4.rs:143-152
                final = v_add(final, trace(pos, normalized(d), max_ray_depth, nodes, tris, materials, textures, rng, counters))
            out[index] = v_div(final, F(samples))
    return out, counters


# ---- bvh.rs -----------------------------------------------------------------------------------------------------------
def _bounds_mid(tri, axis):                                 # scene.rs:114-126, one component
    p = [F(tri["vertices"][k]["position"][axis]) for k in range(3)]
    mn, mx = F32_MAX, -F32_MAX
    for q in p:
        mn, mx = fmin(mn, q), fmax(mx, q)
    return (mn + mx) / F(2.0)


class _Node:
    def __init__(self):
        self.lo, self.hi = [F32_MAX] * 3, [-F32_MAX] * 3
        self.first, self.n = 0, 0

    def grow(self, tri):                                    # bvh.rs:185-194
        for k in range(3):
            for a in range(3):
                q = F(tri["vertices"][k]["position"][a])
                self.lo[a], self.hi[a] = fmin(self.lo[a], q), fmax(self.hi[a], q)

    def area(self):                                         # bvh.rs:196-203
        e = [self.hi[a] - self.lo[a] for a in range(3)]
        return (e[0] * e[2]) + (e[0] * e[1]) + (e[2] * e[1])


def build_bvh(tris):
    """BVH::build (bvh.rs:13-161): returns (reordered copy of tris, nodes as a list of dicts)."""
    tris = tris.copy()
    nodes = []
    root = _Node()
    for t in tris:
        root.grow(t)
    root.n = len(tris)
    nodes.append(root)

    def sah(node, axis, pos):                               # bvh.rs:138-161
        left, right = _Node(), _Node()
        for i in range(node.n):
            t = tris[node.first + i]
            if _bounds_mid(t, axis) < pos:
                left.grow(t); left.n += 1
            else:
                right.grow(t); right.n += 1
        cost = F(left.n) * left.area() + F(right.n) * right.area()
        return cost if cost > 0 else F32_MAX

    def split(index):                                       # bvh.rs:56-136
        used = len(nodes)
        node = nodes[index]
        parent_cost = F(node.n) * node.area()
        best_axis, best_pos, best_cost = 0, F(0), F32_MAX
        for axis in range(3):
            cmin, cmax = F32_MAX, -F32_MAX
            for i in range(node.n):
                m = _bounds_mid(tris[node.first + i], axis)
                cmin, cmax = fmin(cmin, m), fmax(cmax, m)
            if cmin == cmax:
                continue
            scale = (cmax - cmin) / F(8)
            for i in range(1, 8):
                pos = cmin + F(i) * scale
                c = sah(node, axis, pos)
                if c < best_cost:
                    best_axis, best_pos, best_cost = axis, pos, c
        if best_cost >= parent_cost:
            return
        i, j = node.first, node.first + node.n - 1
        while i <= j:
            if _bounds_mid(tris[i], best_axis) < best_pos:
                i += 1
            else:
                tmp = tris[i].copy(); tris[i] = tris[j]; tris[j] = tmp
                j -= 1
        a_count = i - node.first
        if a_count == 0 or a_count == node.n:
            return
        a, b = _Node(), _Node()
        a.first, a.n = node.first, a_count
        b.first, b.n = i, node.n - a_count
        node.first, node.n = used, 0
        for k in range(a.n):
            a.grow(tris[a.first + k])
        for k in range(b.n):
            b.grow(tris[b.first + k])
        nodes.append(a)
        nodes.append(b)
        split(used)
        split(used + 1)

    with np.errstate(all="ignore"):
        split(0)
    return tris, [dict(bounds_min=tuple(n.lo), first_tri_or_child=n.first, bounds_max=tuple(n.hi), num_tris=n.n) for n in nodes]


# ---- loader/obj.rs + scene.rs:44-85 (second reading of the host flatten for config 1) ---------------------------------
def _parse_f32(s):
    """str::parse::<f32>: the decimal value correctly rounded to binary32 (no double rounding through binary64)."""
    from fractions import Fraction
    d = float(s)                                            # raises ValueError like parse().unwrap() panics
    f = np.float32(d)
    if not np.isfinite(f) or "n" in s.lower() or "i" in s.lower():
        return f
    try:
        exact = Fraction(s)
    except ValueError:
        return f
    best = f
    for cand in (np.nextafter(f, np.float32(-np.inf)), np.nextafter(f, np.float32(np.inf))):
        if not np.isfinite(cand):
            continue
        db, dc = abs(Fraction(float(best)) - exact), abs(Fraction(float(cand)) - exact)
        if dc < db or (dc == db and (int(np.float32(cand).view(np.uint32)) & 1) == 0):
            best = np.float32(cand)
    return best


def material_default():                                     # scene.rs:148-167
    return dict(base_color=[F(0.8)] * 3, transmission=F(0), specular_tint=[F(1)] * 3, ior=F(1.45), emission=[F(0)] * 3,
                roughness=F(1), metallic=F(0), transparency=F(1), base_color_tex_id=U32, transparency_tex_id=U32,
                roughness_tex_id=U32, metallic_tex_id=U32, emission_tex_id=U32, normal_tex_id=U32)


def load_mtl(path):                                         # obj.rs:131-265 (texture maps are not followed here)
    mats = {}                                               # insertion-ordered stand-in for the HashMap (its order is random)
    lines = iter(open(path).read().splitlines())
    for line in lines:
        if not line.startswith("newmtl "):
            continue
        name, m = line[len("newmtl "):], material_default()
        for line2 in lines:
            tok = line2.split()
            if not tok:
                break
            p, a = tok[0], tok[1:]
            if p in ("Kd", "Ks", "Ke"):
                key = {"Kd": "base_color", "Ks": "specular_tint", "Ke": "emission"}[p]
                m[key] = list(m[key])
                for i, v in enumerate(a):
                    m[key][i] = _parse_f32(v)               # a 4th value would panic (index out of bounds)
            elif p in ("Ni", "Pr", "Pm", "Tf", "d"):
                m[{"Ni": "ior", "Pr": "roughness", "Pm": "metallic", "Tf": "transmission", "d": "transparency"}[p]] = _parse_f32(a[0])
        mats[name] = m
    return mats


def load_obj(path):
    """OBJ::load + Scene::from(OBJ) up to (not including) BVH::build: returns (fat triangles as a TRIANGLE-dtype array,
    [(name, material dict)] in id order)."""
    import os
    from rust_ray_tracing_amd import _lib as L
    lines = open(path).read().splitlines()
    mtl_line = next((ln for ln in lines if ln.lstrip().startswith("mtllib")), None)
    has_mtl = False
    mats = {"default_material": material_default()}
    if mtl_line is not None:
        rel = mtl_line[len("mtllib "):]
        mats = load_mtl(rel if os.path.isabs(rel) else os.path.join(os.path.dirname(path), rel))
        has_mtl = True
    pos, tex, nrm, tris = [], [], [], []
    active = 0
    def read_index(s):
        i = int(s) - 1
        if i < 0:
            raise ValueError("Tried to load negative indices from an OBJ file")
        return i
    def tri_from(groups):
        t = dict(p=[0, 0, 0], t=[0, 0, 0], n=[0, 0, 0])
        for g_id, g in enumerate(groups):
            if "//" in g:
                parts = g.split("//")
                t["p"][g_id] = read_index(parts[0]); t["n"][g_id] = read_index(parts[1])
            elif "/" in g:
                parts = g.split("/")
                if len(parts) == 2:
                    t["p"][g_id] = read_index(parts[0]); t["t"][g_id] = read_index(parts[1])
                elif len(parts) == 3:
                    t["p"][g_id] = read_index(parts[0]); t["t"][g_id] = read_index(parts[1]); t["n"][g_id] = read_index(parts[2])
            else:
                t["p"][g_id] = read_index(g)
        return t
    for line in lines:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            d = [F(0)] * 3
            for i, v in enumerate(tok[1:]):
                d[i] = _parse_f32(v)
            pos.append(d)
        elif tok[0] == "vt":
            d = [F(0)] * 2
            for i, v in enumerate(tok[1:]):
                d[i] = _parse_f32(v)
            tex.append(d)
        elif tok[0] == "vn":
            d = [F(0)] * 3
            for i, v in enumerate(tok[1:]):
                d[i] = _parse_f32(v)
            nrm.append(d)
        elif tok[0] == "usemtl":
            if has_mtl:
                name = line[len("usemtl "):]
                if name in mats:
                    active = list(mats.keys()).index(name)
        elif tok[0] == "f":
            g = line[len("f "):].split()
            if len(g) == 3:
                new = [tri_from(g)]
            elif len(g) == 4:
                new = [tri_from([g[0], g[1], g[3]]), tri_from([g[1], g[2], g[3]])]
            elif len(g) >= 5:
                new = [tri_from([g[0], g[i + 1], g[i + 2]]) for i in range(len(g) - 2)]
            else:
                raise ValueError("face with fewer than 3 vertices")
            for t in new:
                t["m"] = active
                tris.append(t)
    if not nrm:                                             # obj.rs:107-120: one flat normal per triangle
        with np.errstate(all="ignore"):
            for i, t in enumerate(tris):
                v1, v2, v3 = (vec(pos[t["p"][k]]) for k in range(3))
                nrm.append(list(normalized(cross(v_sub(v2, v1), v_sub(v3, v1)))))
                t["n"] = [i, i, i]
    out = np.zeros(len(tris), dtype=L.TRIANGLE)
    for k, t in enumerate(tris):                            # scene.rs:47-75: a missing index reads element 0, an absent buffer zeros
        for i in range(3):
            out[k]["vertices"][i]["position"] = pos[t["p"][i]] if t["p"][i] < len(pos) else [0, 0, 0]
            tc = tex[t["t"][i]] if t["t"][i] < len(tex) else [0, 0]
            out[k]["vertices"][i]["tex_coord_x"], out[k]["vertices"][i]["tex_coord_y"] = tc[0], tc[1]
            out[k]["vertices"][i]["normal"] = nrm[t["n"][i]] if t["n"][i] < len(nrm) else [0, 0, 0]
        out[k]["material_id"] = t["m"]
    return out, list(mats.items())


# ---- scene.rs:181-194 + mat4.rs:25-44 (host, once per frame; sin/cos are the platform libm's, as in the reference) -------
def camera_from_pose(position, pitch_deg, yaw_deg):
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.sinf.restype = libm.cosf.restype = ctypes.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [ctypes.c_float]
    sinf = lambda x: F(libm.sinf(ctypes.c_float(float(x))))
    cosf = lambda x: F(libm.cosf(ctypes.c_float(float(x))))
    rad = F(0.017453292519943295769236907684886)            # f32::to_radians multiplies by this constant
    yaw, pitch = F(yaw_deg) * rad, F(pitch_deg) * rad
    with np.errstate(all="ignore"):
        direction = (cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch))
        pos = vec(position)
        forward = normalized(direction)
        right = normalized(cross((F(0), F(1), F(0)), forward))
        up = cross(forward, right)
        # Mat4f::look_at(from = position, to = position + forward, up)
        to = v_add(pos, forward)
        f = normalized(v_sub(pos, to))
        r = normalized(cross(up, f))
        u = cross(f, r)
    m = np.zeros((4, 4), np.float32)
    for i in range(4):
        m[i][i] = 1
    m[0][:3], m[1][:3], m[2][:3], m[3][:3] = r, u, f, pos
    return m, np.array(pos, np.float32)

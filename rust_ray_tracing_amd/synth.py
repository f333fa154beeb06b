"""Seeded procedural stand-ins for the scenes BASELINE.json's configs name.

No assets ship with the reference (res/ holds only .keep), so every config uses a
deterministic synthetic scene with the stated triangle count.  Generator
seed = 0x5EED0001 + config number.  All geometry is f32; texture coordinates stay in [0, 8) so
Texture::color_at (reference src/texture.rs:33-38) never sees a negative uv.

Each generator returns ``(tris, materials, textures, camera_pose)`` where ``tris`` is a numpy array
of the reference's 112-byte Triangle, ``materials`` an ordered dict name -> 80-byte Material,
``textures`` a list of (h, w, 4) uint8 arrays and ``camera_pose`` = (position, pitch_deg, yaw_deg).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import numpy as np

from ._lib import MATERIAL, NO_TEXTURE, TRIANGLE

SEED0 = 0x5EED0001


# ---------------------------------------------------------------------------------------------
def material(base=(0.8, 0.8, 0.8), emission=(0.0, 0.0, 0.0), base_tex=NO_TEXTURE, emission_tex=NO_TEXTURE) -> np.ndarray:
    """Material::default() (scene.rs:148-167) with the fields the CPU path reads overridden."""
    m = np.zeros((), dtype=MATERIAL)
    m["base_color"] = base
    m["transmission"] = 0.0
    m["specular_tint"] = (1.0, 1.0, 1.0)
    m["ior"] = 1.45
    m["emission"] = emission
    m["roughness"] = 1.0
    m["metallic"] = 0.0
    m["transparency"] = 1.0
    for k in ("transparency_tex_id", "roughness_tex_id", "metallic_tex_id", "normal_tex_id"):
        m[k] = NO_TEXTURE
    m["base_color_tex_id"] = base_tex
    m["emission_tex_id"] = emission_tex
    return m


def tris_from(p: np.ndarray, n: np.ndarray, uv: np.ndarray, mat) -> np.ndarray:
    """p, n: (T,3,3); uv: (T,3,2); mat: int or (T,) -> (T,) TRIANGLE."""
    t = np.zeros(p.shape[0], dtype=TRIANGLE)
    t["vertices"]["position"] = p.astype(np.float32)
    t["vertices"]["normal"] = n.astype(np.float32)
    t["vertices"]["tex_coord_x"] = uv[..., 0].astype(np.float32)
    t["vertices"]["tex_coord_y"] = uv[..., 1].astype(np.float32)
    t["material_id"] = mat
    return t


def grid_mesh(P: np.ndarray, UV: np.ndarray, mat, N: np.ndarray | None = None) -> np.ndarray:
    """Triangulate a (nu+1, nv+1) vertex grid.  Smooth normals from central differences if N is None."""
    P = P.astype(np.float32)
    if N is None:
        du = np.gradient(P.astype(np.float64), axis=0)
        dv = np.gradient(P.astype(np.float64), axis=1)
        N = np.cross(du, dv)
        ln = np.linalg.norm(N, axis=-1, keepdims=True)
        N = np.where(ln > 1e-20, N / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
    a, b, c, d = (slice(None, -1), slice(None, -1)), (slice(1, None), slice(None, -1)), (slice(1, None), slice(1, None)), (slice(None, -1), slice(1, None))

    def tri(i0, i1, i2, arr):
        return np.stack([arr[i0], arr[i1], arr[i2]], axis=2).reshape(-1, 3, arr.shape[-1])

    p = np.concatenate([tri(a, b, c, P), tri(a, c, d, P)])
    n = np.concatenate([tri(a, b, c, N), tri(a, c, d, N)])
    uv = np.concatenate([tri(a, b, c, UV), tri(a, c, d, UV)])
    if not np.isscalar(mat):
        mat = np.concatenate([np.asarray(mat).reshape(-1), np.asarray(mat).reshape(-1)])
    return tris_from(p, n, uv, mat)


def quad(p0, p1, p2, p3, normal, mat, uv_scale=1.0) -> np.ndarray:
    """Two triangles (p0,p1,p3), (p1,p2,p3) -- the reference's quad split (obj.rs:412-419)."""
    P = np.array([[p0, p1, p3], [p1, p2, p3]], dtype=np.float32)
    N = np.broadcast_to(np.asarray(normal, dtype=np.float32), P.shape).copy()
    U = np.array([[[0, 0], [1, 0], [0, 1]], [[1, 0], [1, 1], [0, 1]]], dtype=np.float32) * uv_scale
    return tris_from(P, N, U, mat)


def value_noise_texture(rng: np.random.Generator, size: int, base, contrast=0.35, cells=16, checker=False) -> np.ndarray:
    """Tileable value-noise RGBA8 texture tinted by `base` (rows as Texture::load stores them)."""
    g = rng.random((cells, cells))
    t = np.linspace(0, cells, size, endpoint=False)
    i0 = np.floor(t).astype(int) % cells
    i1 = (i0 + 1) % cells
    f = t - np.floor(t)
    f = f * f * (3 - 2 * f)
    rows = g[i0][:, i0] * (1 - f)[None, :] + g[i0][:, i1] * f[None, :]
    rows1 = g[i1][:, i0] * (1 - f)[None, :] + g[i1][:, i1] * f[None, :]
    v = rows * (1 - f)[:, None] + rows1 * f[:, None]
    if checker:
        c = ((np.arange(size)[:, None] * 8 // size) + (np.arange(size)[None, :] * 8 // size)) % 2
        v = 0.5 * v + 0.5 * c
    shade = 1.0 - contrast + contrast * v
    rgb = np.clip(np.asarray(base)[None, None, :] * shade[:, :, None], 0, 1)
    out = np.empty((size, size, 4), dtype=np.uint8)
    out[..., :3] = np.floor(rgb * 255.0 + 0.5).astype(np.uint8)
    out[..., 3] = 255
    return out


# ---------------------------------------------------------------------------------------------
# Config 1: Cornell-box-style 12-triangle OBJ
CORNELL_CAMERA = ((3.2, 0.0, 0.0), 0.0, 0.0)   # pitch = yaw = 0: looks down -X

_CORNELL_QUADS = [  # (name, 4 corners, inward normal)
    ("white", [(-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1)], (0, 1, 0)),     # floor
    ("white", [(-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1)], (0, -1, 0)),        # ceiling
    ("white", [(-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1)], (1, 0, 0)),     # back wall
    ("red", [(-1, -1, -1), (-1, 1, -1), (1, 1, -1), (1, -1, -1)], (0, 0, 1)),       # z = -1 wall
    ("green", [(-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)], (0, 0, -1)),        # z = +1 wall
    ("light", [(-0.3, 0.98, -0.3), (-0.3, 0.98, 0.3), (0.3, 0.98, 0.3), (0.3, 0.98, -0.3)], (0, -1, 0)),
]
_CORNELL_MTL = OrderedDict([
    ("white", dict(Kd=(0.73, 0.73, 0.73), Ke=(0, 0, 0))),
    ("red", dict(Kd=(0.65, 0.05, 0.05), Ke=(0, 0, 0))),
    ("green", dict(Kd=(0.12, 0.45, 0.15), Ke=(0, 0, 0))),
    ("light", dict(Kd=(0.78, 0.78, 0.78), Ke=(4.0, 3.5, 3.0))),
])


def write_cornell_obj(directory: str) -> str:
    """Writes cornell.obj + cornell.mtl (quads as 4-index faces, v/vt/vn) and returns the .obj path."""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "cornell.mtl"), "w") as f:
        for name, m in _CORNELL_MTL.items():
            f.write(f"newmtl {name}\nKd {m['Kd'][0]} {m['Kd'][1]} {m['Kd'][2]}\nKe {m['Ke'][0]} {m['Ke'][1]} {m['Ke'][2]}\nNi 1.45\n\n")
    lines = ["# Cornell-style box: 5 walls + 1 emitter quad = 12 triangles", "mtllib cornell.mtl"]
    vt = [(0, 0), (1, 0), (1, 1), (0, 1)]
    for u, v in vt:
        lines.append(f"vt {u} {v}")
    vi = 1
    faces = []
    for qi, (name, corners, nrm) in enumerate(_CORNELL_QUADS):
        for c in corners:
            lines.append(f"v {c[0]} {c[1]} {c[2]}")
        lines.append(f"vn {nrm[0]} {nrm[1]} {nrm[2]}")
        faces.append((name, [f"{vi + k}/{k + 1}/{qi + 1}" for k in range(4)]))
        vi += 4
    for name, idx in faces:
        lines.append(f"usemtl {name}")
        lines.append("f " + " ".join(idx))
    path = os.path.join(directory, "cornell.obj")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def write_obj(directory: str, name: str, tris: np.ndarray, mats, texs=()) -> str:
    """Writes a scene as <name>.obj + <name>.mtl + <name>_tex<i>.png and returns the .obj path: the on-disk form the reference loads
    (src/loader/obj.rs).  Triangles go out in array order through libmipt_diag.so's writer (shortest round-trip decimals, so
    mipt_obj_load returns `tris` bit for bit); materials keep their names (a dict) or are named material_<id>, written in id order with every MTL key the
    loader reads (obj.rs:150-257); textures as 8-bit RGBA PNGs through mipt_image_save_png (rows flipped back: Texture::load flips)."""
    import ctypes as C
    from . import _lib as L
    os.makedirs(directory, exist_ok=True)
    lib, diag = L.load(), L.load_diag()
    tex_files = []
    for i, t in enumerate(texs):
        t = np.ascontiguousarray(np.asarray(t, dtype=np.uint8)[::-1])
        fn = f"{name}_tex{i}.png"
        L.check(lib.mipt_image_save_png(os.path.join(directory, fn).encode(), t.shape[1], t.shape[0], 8, L.ptr(t)), "mipt_image_save_png")
        tex_files.append(fn)
    if isinstance(mats, dict):
        mat_names = list(mats.keys())
        mats = list(mats.values())
    else:
        mat_names = [f"material_{i}" for i in range(len(mats))]
    mats = [np.asarray(m, dtype=MATERIAL).reshape(()) for m in mats]

    def num(x):
        return repr(float(np.float32(x)))

    with open(os.path.join(directory, name + ".mtl"), "w") as f:
        for i, m in enumerate(mats):
            f.write(f"newmtl {mat_names[i]}\n")
            for key, field in (("Kd", "base_color"), ("Ks", "specular_tint"), ("Ke", "emission")):
                f.write(f"{key} {num(m[field][0])} {num(m[field][1])} {num(m[field][2])}\n")
            for key, field in (("Ni", "ior"), ("Pr", "roughness"), ("Pm", "metallic"), ("Tf", "transmission"), ("d", "transparency")):
                f.write(f"{key} {num(m[field])}\n")
            for key, field in (("map_Kd", "base_color_tex_id"), ("map_d", "transparency_tex_id"), ("map_Pr", "roughness_tex_id"),
                               ("map_Pm", "metallic_tex_id"), ("map_Ke", "emission_tex_id"), ("map_Bump", "normal_tex_id")):
                if int(m[field]) != NO_TEXTURE:
                    f.write(f"{key} {tex_files[int(m[field])]}\n")
            f.write("\n")
    tris = np.ascontiguousarray(tris, dtype=TRIANGLE)
    names = (C.c_char_p * max(len(mats), 1))(*[n.encode() for n in mat_names])
    path = os.path.join(directory, name + ".obj")
    rc = diag.mipt_diag_write_obj(path.encode(), L.ptr(tris), len(tris), (name + ".mtl").encode(), names, len(mats))
    if rc != 0:
        raise OSError(f"mipt_diag_write_obj({path}) failed with {rc}")
    return path


def cornell_box():
    """The same box as arrays (quad split (0,1,3),(1,2,3) as obj.rs:412-419)."""
    names = list(_CORNELL_MTL.keys())
    mats = OrderedDict((n, material(base=m["Kd"], emission=m["Ke"])) for n, m in _CORNELL_MTL.items())
    parts = []
    vt = np.array([(0, 0), (1, 0), (1, 1), (0, 1)], dtype=np.float32)
    for name, c, nrm in _CORNELL_QUADS:
        c = np.asarray(c, dtype=np.float32)
        P = np.array([[c[0], c[1], c[3]], [c[1], c[2], c[3]]])
        U = np.array([[vt[0], vt[1], vt[3]], [vt[1], vt[2], vt[3]]])
        N = np.broadcast_to(np.asarray(nrm, dtype=np.float32), P.shape).copy()
        parts.append(tris_from(P, N, U, names.index(name)))
    return np.concatenate(parts), mats, [], CORNELL_CAMERA


# ---------------------------------------------------------------------------------------------
def _displaced_sphere(nu, nv, radius_fn, center=(0.0, 0.0, 0.0), uv_scale=4.0, mat=0, mat_fn=None):
    th = np.linspace(0.0, math.pi, nu + 1)[:, None]            # polar
    ph = np.linspace(0.0, 2.0 * math.pi, nv + 1)[None, :]      # azimuth
    r = radius_fn(th, ph)
    P = np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)], axis=-1) + np.asarray(center)
    UV = np.stack([np.broadcast_to(ph / (2 * math.pi) * uv_scale, r.shape), np.broadcast_to(th / math.pi * uv_scale, r.shape)], axis=-1)
    if mat_fn is not None:
        thc = 0.5 * (th[:-1] + th[1:])
        phc = 0.5 * (ph[:, :-1] + ph[:, 1:])
        mat = mat_fn(np.broadcast_to(thc, (nu, nv)), np.broadcast_to(phc, (nu, nv)))
    return grid_mesh(P, UV, mat)


def helmet_scene(n_target: int = 15000, tex_size: int = 512):
    """Config 2 stand-in ("Damaged Helmet", ~15k tris): displaced UV sphere + ground, 3 materials, one texture."""
    rng = np.random.default_rng(SEED0 + 2)
    n = max(4, int(round(math.sqrt((n_target - 2) / 2.0))))
    ph0 = rng.random(4) * 6.28

    def radius(th, ph):
        return (1.0 + 0.07 * np.sin(5 * th + ph0[0]) * np.sin(7 * ph + ph0[1]) + 0.03 * np.sin(17 * th + ph0[2]) * np.sin(13 * ph + ph0[3]))

    def mats_of(th, ph):   # an emissive "visor" band
        return np.where((np.abs(th - 1.35) < 0.12) & (np.abs(ph - math.pi) < 0.9), 2, 0).astype(np.uint32)

    helmet = _displaced_sphere(n, n, radius, uv_scale=4.0, mat_fn=mats_of)
    g = 12.0
    ground = quad((-g, -1.15, -g), (g, -1.15, -g), (g, -1.15, g), (-g, -1.15, g), (0, 1, 0), 1, uv_scale=6.0)
    tex = value_noise_texture(rng, tex_size, (0.85, 0.62, 0.45), contrast=0.5)
    mats = OrderedDict([
        ("helmet", material(base=(0.8, 0.8, 0.8), base_tex=0)),
        ("ground", material(base=(0.8, 0.8, 0.8))),
        ("visor", material(base=(0.2, 0.2, 0.25), emission=(2.0, 1.6, 1.0))),
    ])
    return np.concatenate([helmet, ground]), mats, [tex], ((3.0, 0.55, 0.0), 10.0, 0.0)


def dragon_scene(n_target: int = 870000):
    """Config 3 stand-in ("Chinese Dragon", ~870k tris): multi-octave displaced sphere with thin spikes."""
    rng = np.random.default_rng(SEED0 + 3)
    n = max(8, int(round(math.sqrt((n_target - 2) / 2.0))))
    oct_ = [(rng.integers(2, 40), rng.integers(2, 40), rng.random() * 6.28, rng.random() * 6.28, 0.12 / (k + 1)) for k in range(6)]

    def radius(th, ph):
        r = np.ones(np.broadcast(th, ph).shape)
        for f, g, p, q, a in oct_:
            r = r + a * np.sin(f * th + p) * np.sin(g * ph + q)
        spikes = np.maximum(0.0, np.sin(24 * th) * np.sin(24 * ph)) ** 10
        return r + 0.55 * spikes

    body = _displaced_sphere(n, n, radius, uv_scale=6.0, mat=0)
    g = 15.0
    ground = quad((-g, -1.6, -g), (g, -1.6, -g), (g, -1.6, g), (-g, -1.6, g), (0, 1, 0), 1, uv_scale=6.0)
    mats = OrderedDict([("jade", material(base=(0.45, 0.75, 0.5))), ("ground", material(base=(0.8, 0.8, 0.8)))])
    return np.concatenate([body, ground]), mats, [], ((3.4, 0.7, 0.3), 12.0, 5.0)


# ---------------------------------------------------------------------------------------------
SPONZA_CAMERA = ((-11.204422, 2.1092458, -0.12164927), 1.5998944, -179.10223)   # reference src/main.rs:41-43


def atrium_scene(n_target: int = 10_000_000, tex_size: int = 1024, seed_offset: int = 4):
    """Configs 4/5/M stand-in ("Intel Sponza + curtains", ~10M tris): a two-storey colonnaded atrium, open
    to the white sky, with finely tessellated curtains; ~25 materials, 10 textures."""
    rng = np.random.default_rng(SEED0 + seed_offset)
    # relative tessellation weights (cells at scale 1.0); curtains dominate
    base = dict(floor=60 * 24, wall=60 * 18, endwall=24 * 18, column=24 * 40, arch=24 * 12, slab=40 * 6, curtain=220 * 220, lamp=16 * 16)
    counts = dict(floor=1, wall=2, endwall=2, column=36, arch=32, slab=4, curtain=16, lamp=6)
    tot = sum(2 * base[k] * counts[k] for k in base)
    s = math.sqrt(max(n_target, 2000) / tot)

    def res(k, ratio=1.0):
        cells = base[k] * s * s
        a = max(2, int(round(math.sqrt(cells * ratio))))
        b = max(2, int(round(cells / a)))
        return a, b

    textures = []
    tints = [(0.78, 0.72, 0.62), (0.70, 0.66, 0.60), (0.66, 0.62, 0.58), (0.75, 0.2, 0.18), (0.2, 0.35, 0.7),
             (0.25, 0.6, 0.3), (0.8, 0.7, 0.25), (0.6, 0.3, 0.65), (0.85, 0.85, 0.8), (0.5, 0.45, 0.4)]
    for i, tint in enumerate(tints):
        textures.append(value_noise_texture(rng, tex_size, tint, contrast=0.4, cells=8 + 4 * (i % 3), checker=(i == 0)))
    mats = OrderedDict()
    mats["floor"] = material(base_tex=0)
    mats["wall_a"] = material(base_tex=1)
    mats["wall_b"] = material(base_tex=2)
    mats["endwall"] = material(base_tex=9)
    for i in range(5):
        mats[f"stone_{i}"] = material(base=(0.72 - 0.04 * i, 0.7 - 0.04 * i, 0.66 - 0.04 * i))
    mats["arch"] = material(base=(0.68, 0.66, 0.62))
    mats["slab"] = material(base_tex=8)
    for i in range(5):
        mats[f"fabric_tex_{i}"] = material(base_tex=3 + i)
    fabric_cols = [(0.7, 0.15, 0.12), (0.15, 0.3, 0.65), (0.2, 0.55, 0.25), (0.75, 0.65, 0.2), (0.55, 0.25, 0.6), (0.8, 0.8, 0.75)]
    for i, c in enumerate(fabric_cols):
        mats[f"fabric_{i}"] = material(base=c)
    mats["lamp_warm"] = material(base=(0.9, 0.9, 0.9), emission=(6.0, 4.5, 3.0))
    mats["lamp_cool"] = material(base=(0.9, 0.9, 0.9), emission=(3.0, 4.0, 6.0))
    mid = {k: i for i, k in enumerate(mats.keys())}
    parts = []

    X0, X1, ZW, H = -20.0, 20.0, 8.0, 12.0
    # floor with shallow tile relief
    a, b = res("floor", 2.5)
    x = np.linspace(X0, X1, a + 1)[:, None]
    z = np.linspace(-ZW, ZW, b + 1)[None, :]
    y = 0.015 * np.sin(3.0 * x) * np.sin(3.0 * z)
    P = np.stack([np.broadcast_to(x, y.shape), y, np.broadcast_to(z, y.shape)], axis=-1)
    UV = np.stack([np.broadcast_to((x - X0) / 5.0, y.shape), np.broadcast_to((z + ZW) / 2.0, y.shape)], axis=-1)
    parts.append(grid_mesh(P, UV, mid["floor"]))
    # long walls z = +-ZW (slightly rippled)
    for sgn, mname in ((-1.0, "wall_a"), (1.0, "wall_b")):
        a, b = res("wall", 3.3)
        x = np.linspace(X0, X1, a + 1)[:, None]
        yy = np.linspace(0.0, H, b + 1)[None, :]
        zz = sgn * (ZW + 0.02 * np.sin(2.0 * x) * np.sin(2.5 * yy))
        P = np.stack([np.broadcast_to(x, zz.shape), np.broadcast_to(yy, zz.shape), zz], axis=-1)
        UV = np.stack([np.broadcast_to((x - X0) / 5.0, zz.shape), np.broadcast_to(yy / 1.5, zz.shape)], axis=-1)
        parts.append(grid_mesh(P, UV, mid[mname]))
    # end walls x = X0, X1
    for xe in (X0, X1):
        a, b = res("endwall", 1.3)
        z = np.linspace(-ZW, ZW, a + 1)[:, None]
        yy = np.linspace(0.0, H, b + 1)[None, :]
        xx = xe + 0.02 * np.sin(2.0 * z) * np.sin(2.5 * yy)
        P = np.stack([xx, np.broadcast_to(yy, xx.shape), np.broadcast_to(z, xx.shape)], axis=-1)
        UV = np.stack([np.broadcast_to((z + ZW) / 2.0, xx.shape), np.broadcast_to(yy / 1.5, xx.shape)], axis=-1)
        parts.append(grid_mesh(P, UV, mid["endwall"]))
    # gallery slabs (top and underside) between the colonnade and the wall
    for sgn in (-1.0, 1.0):
        for yy in (5.5, 6.5):
            a, b = res("slab", 6.0)
            x = np.linspace(X0, X1, a + 1)[:, None]
            z = sgn * np.linspace(4.5, ZW, b + 1)[None, :]
            P = np.stack([np.broadcast_to(x, (a + 1, b + 1)), np.full((a + 1, b + 1), yy), np.broadcast_to(z, (a + 1, b + 1))], axis=-1)
            UV = np.stack([np.broadcast_to((x - X0) / 5.0, (a + 1, b + 1)), np.broadcast_to(np.abs(z) / 1.0, (a + 1, b + 1))], axis=-1)
            parts.append(grid_mesh(P, UV, mid["slab"]))
    # fluted columns: 2 rows x 2 storeys x 9
    col_x = np.linspace(-16.0, 16.0, 9)
    ci = 0
    for sgn in (-1.0, 1.0):
        for (y0, y1) in ((0.0, 5.5), (6.5, H)):
            for cx in col_x:
                a, b = res("column", 0.6)
                ph = np.linspace(0.0, 2 * math.pi, a + 1)[:, None]
                yy = np.linspace(y0, y1, b + 1)[None, :]
                t = (yy - y0) / (y1 - y0)
                r = (0.42 + 0.03 * np.cos(16 * ph)) * (1.0 + 0.25 * (np.exp(-30 * t) + np.exp(-30 * (1 - t))))
                P = np.stack([cx + r * np.cos(ph), np.broadcast_to(yy, r.shape), sgn * 5.0 + r * np.sin(ph)], axis=-1)
                UV = np.stack([np.broadcast_to(ph / (2 * math.pi) * 4.0, r.shape), np.broadcast_to(t * 6.0, r.shape)], axis=-1)
                parts.append(grid_mesh(P, UV, mid[f"stone_{ci % 5}"]))
                ci += 1
    # arches: half tori between neighbouring columns at both storey tops
    for sgn in (-1.0, 1.0):
        for ytop in (5.5 - 0.05, H - 0.05):
            for k in range(8):
                a, b = res("arch", 2.0)
                cxm = 0.5 * (col_x[k] + col_x[k + 1])
                R = 0.5 * (col_x[k + 1] - col_x[k])
                al = np.linspace(0.0, math.pi, a + 1)[:, None]
                be = np.linspace(0.0, 2 * math.pi, b + 1)[None, :]
                al, be = np.broadcast_arrays(al, be)
                rr = R + 0.28 * np.cos(be)
                ybase = ytop - 0.55 * (R + 0.28)
                P = np.stack([cxm + rr * np.cos(al), ybase + 0.55 * rr * np.sin(al), sgn * 5.0 + 0.28 * np.sin(be)], axis=-1)
                UV = np.stack([al / math.pi * 4.0, be / (2 * math.pi) * 2.0], axis=-1)
                parts.append(grid_mesh(P, UV, mid["arch"]))
    # curtains: 16 finely tessellated sinusoidal sheets in the upper storey
    k = 0
    for sgn in (-1.0, 1.0):
        for j in range(8):
            a, b = res("curtain", 1.0)
            xs = np.linspace(col_x[j] + 0.5, col_x[j + 1] - 0.5, a + 1)[:, None]
            yy = np.linspace(6.6, 11.6, b + 1)[None, :]
            fall = (11.6 - yy) / 5.0
            phase = rng.random() * 6.28
            zz = sgn * 5.3 + (0.22 * np.sin(9.0 * xs + phase) + 0.06 * np.sin(31.0 * xs + 2 * phase)) * (0.25 + fall) + 0.05 * np.sin(4 * yy + phase)
            P = np.stack([np.broadcast_to(xs, zz.shape), np.broadcast_to(yy, zz.shape), zz], axis=-1)
            UV = np.stack([np.broadcast_to((xs - xs.min()) / 0.5, zz.shape), np.broadcast_to((yy - 6.6) / 0.7, zz.shape)], axis=-1)
            name = f"fabric_tex_{k % 5}" if k % 2 == 0 else f"fabric_{k % 6}"
            parts.append(grid_mesh(P, UV, mid[name]))
            k += 1
    # hanging lamps
    for i, cx in enumerate(np.linspace(-14.0, 14.0, 6)):
        a, b = res("lamp", 1.0)
        parts.append(_displaced_sphere(a, b, lambda th, ph: 0.3 + 0.0 * th * ph, center=(cx, 4.6, 0.0 if i % 2 else 1.5), uv_scale=1.0,
                                       mat=mid["lamp_warm" if i % 2 == 0 else "lamp_cool"]))
    tris = np.concatenate(parts)
    return tris, mats, textures, SPONZA_CAMERA


def make_scene(kind: str, **kw):
    """kind in {"cornell", "helmet", "dragon", "atrium"} -> (tris, materials, textures, camera_pose)."""
    return {"cornell": cornell_box, "helmet": helmet_scene, "dragon": dragon_scene, "atrium": atrium_scene}[kind](**kw)

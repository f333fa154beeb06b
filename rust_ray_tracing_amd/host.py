"""Host-side mirror of the reference's Renderer / Scene API for the path-tracing hot path.

Names, argument meaning and error behaviour follow the reference
(src/renderer.rs:8-117, src/renderer/backend.rs:6-10, src/scene.rs:12-195):

* ``Renderer.new(options)`` validates like ``Renderer::new`` and returns ``None`` (after logging
  the reference's message) instead of raising;
* ``Renderer.render(scene)`` dispatches on ``options.backend``; the new arm is
  ``RendererBackend.MI355X`` which goes through the C ABI of libmipt.so;
* ``Scene.load(path)`` / ``Scene.set_camera(camera)`` / ``Camera.update_view()`` as in scene.rs.

Everything numerical happens behind the C ABI (include/mipt.h); this module only marshals.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
import sys
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib as L


def log_error(msg: str) -> None:  # log.rs:22-29
    print(f"[ERROR] {msg}", file=sys.stderr)


def log_info(msg: str) -> None:  # log.rs:2-9
    if os.environ.get("MIPT_LOG"):
        print(f"[INFO] {msg}", file=sys.stderr)


class RendererBackend(enum.Enum):  # renderer/backend.rs:6-10 + the new arm
    GPU = "GPU"        # the reference's wgpu backend: not part of this build
    CPU = "CPU"        # the reference's rayon backend: not part of this build (see oracle/ for tests)
    MI355X = "MI355X"  # this build: gfx950 megakernel behind the C ABI


@dataclass
class RendererOptions:  # renderer.rs:96-116
    samples: int = 1
    max_ray_depth: int = 6
    output_image_dimensions: Tuple[int, int] = (1920, 1080)
    output_image_path: Optional[str] = None
    backend: RendererBackend = RendererBackend.MI355X
    is_realtime: bool = False
    # MI355X-path extensions (MiptOptions)
    seed_mode: int = L.SEED_PIXEL_STREAM
    # the CPU backend's own un-culled traversal (ray.rs:69-81): the reference by construction, and what the parity tests compare
    # counter for counter with the oracle.  Production use: TRAVERSAL_CULLED with CULL_MARGIN_SAFE -- same frame, 1.7x faster
    # (the C++ mirror include/mipt_host.hpp and the Rust binding default to it).
    traversal: int = L.TRAVERSAL_REFERENCE
    cull_margin: float = L.CULL_MARGIN_SAFE
    shading: int = L.SHADING_CPU          # SHADING_WGPU: the wgpu shader's material model (rt_compute.wgsl)
    device_id: int = 0


@dataclass
class Camera:  # scene.rs:169-195
    pitch: float = 0.0
    yaw: float = 0.0
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    uniform: np.ndarray = field(default_factory=lambda: np.zeros((), dtype=L.CAMERA))

    def update_view(self) -> None:
        pos = (C.c_float * 3)(*[float(x) for x in self.position])
        out = np.zeros((), dtype=L.CAMERA)
        L.check(L.load().mipt_camera_from_pose(C.byref(pos), C.c_float(self.pitch), C.c_float(self.yaw), L.ptr(out)),
                "mipt_camera_from_pose")
        self.uniform = out


def material_default() -> np.ndarray:  # scene.rs:148-167
    m = np.zeros((), dtype=L.MATERIAL)
    L.load().mipt_material_default(L.ptr(m))
    return m


@dataclass
class Texture:  # texture.rs:3-10
    width: int
    height: int
    hash: int
    pixel_data: np.ndarray          # (height, width, 4) uint8, rows as Texture::load stores them (flipv applied)

    @staticmethod
    def load(path: str) -> Optional["Texture"]:  # texture.rs:13-31 (None + log line when the file is missing / undecodable)
        lib = L.load()
        img = C.c_void_p()
        desc = L.MiptTexture()
        h = C.c_uint32()
        if lib.mipt_texture_load(os.fsencode(path), C.byref(img), C.byref(desc), C.byref(h)) != 0:
            log_error(lib.mipt_last_error().decode())
            return None
        try:
            px = np.ctypeslib.as_array(C.cast(desc.rgba8, C.POINTER(C.c_uint8)), (desc.height, desc.width, 4)).copy()
            return Texture(int(desc.width), int(desc.height), int(h.value), px)
        finally:
            lib.mipt_texture_free(img)


class Scene:  # scene.rs:12-19
    """tris / materials / textures / bvh / camera.  ``materials`` is name -> Material in id order."""

    def __init__(self):
        self.tris: np.ndarray = np.zeros(0, dtype=L.TRIANGLE)
        self.materials: Dict[str, np.ndarray] = {}
        self.textures: List[np.ndarray] = []      # each (h, w, 4) uint8, rows as Texture::load stores them
        self.bvh_nodes: np.ndarray = np.zeros(0, dtype=L.NODE)
        self.camera = Camera()
        self._handle: Optional[C.c_void_p] = None
        self._handle_device = -1

    # -- construction ------------------------------------------------------------------------
    @staticmethod
    def load(path: str, build_bvh: bool = True) -> Optional["Scene"]:  # scene.rs:22-36
        """``build_bvh=False``: OBJ::load + From<OBJ> without the BVH::build at scene.rs:80 (mipt_obj_load_triangles) -- triangles in file
        order for ``upload_from_triangles``, which builds the tree on the GPU."""
        lib = L.load()
        obj = C.c_void_p()
        rc = (lib.mipt_obj_load if build_bvh else lib.mipt_obj_load_triangles)(os.fsencode(path), C.byref(obj))
        if rc != 0:
            log_error(lib.mipt_last_error().decode())
            return None
        try:
            desc = L.MiptSceneDesc()
            names = C.POINTER(C.c_char_p)()
            L.check(lib.mipt_obj_get(obj, C.byref(desc), C.byref(names)), "mipt_obj_get")
            sc = Scene()
            sc.tris = np.ctypeslib.as_array(C.cast(desc.tris, C.POINTER(C.c_uint8)), (desc.n_tris * 112,)).copy().view(L.TRIANGLE)
            if desc.n_nodes:
                sc.bvh_nodes = np.ctypeslib.as_array(C.cast(desc.nodes, C.POINTER(C.c_uint8)), (desc.n_nodes * 32,)).copy().view(L.NODE)
            mats = np.ctypeslib.as_array(C.cast(desc.materials, C.POINTER(C.c_uint8)), (desc.n_materials * 80,)).copy().view(L.MATERIAL)
            for i in range(desc.n_materials):
                sc.materials[names[i].decode()] = mats[i].copy()
            for i in range(desc.n_textures):
                t = desc.textures[i]
                px = np.ctypeslib.as_array(C.cast(t.rgba8, C.POINTER(C.c_uint8)), (t.height, t.width, 4)).copy()
                sc.textures.append(px)
            return sc
        finally:
            lib.mipt_obj_free(obj)

    @staticmethod
    def from_arrays(tris: np.ndarray, materials, textures=(), build_bvh: bool = True, threads: int = 0) -> "Scene":
        """impl From<OBJ> for Scene (scene.rs:44-85) for already-expanded triangles."""
        sc = Scene()
        sc.tris = np.ascontiguousarray(tris, dtype=L.TRIANGLE).copy()
        if isinstance(materials, dict):
            sc.materials = {k: np.asarray(v, dtype=L.MATERIAL).reshape(()) for k, v in materials.items()}
        else:
            sc.materials = {f"material_{i}": np.asarray(m, dtype=L.MATERIAL).reshape(()) for i, m in enumerate(materials)}
        sc.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        if build_bvh:
            sc.build_bvh(threads)
        return sc

    def build_bvh(self, threads: int = 0) -> None:  # BVH::build, bvh.rs:13-54
        n = len(self.tris)
        nodes = np.zeros(max(2 * n, 1), dtype=L.NODE)
        count = C.c_uint32(0)
        L.check(L.load().mipt_bvh_build(L.ptr(self.tris), n, L.ptr(nodes), len(nodes), C.byref(count), threads), "mipt_bvh_build")
        self.bvh_nodes = nodes[: count.value].copy()
        self.release()

    def build_bvh_device(self, device_id: int = 0) -> float:
        """BVH::build on the GPU (mipt_bvh_build_device); returns the device build time in ms."""
        n = len(self.tris)
        nodes = np.zeros(max(2 * n, 1), dtype=L.NODE)
        count = C.c_uint32(0)
        ms = C.c_double(0.0)
        L.check(L.load().mipt_bvh_build_device(L.ptr(self.tris), n, L.ptr(nodes), len(nodes), C.byref(count), device_id, C.byref(ms)),
                "mipt_bvh_build_device")
        self.bvh_nodes = nodes[: count.value].copy()
        self.release()
        return ms.value

    def set_camera(self, camera: Camera) -> None:  # scene.rs:38-41
        self.camera = camera
        self.camera.update_view()

    # -- device residency --------------------------------------------------------------------
    def materials_array(self) -> np.ndarray:
        return np.array([m for m in self.materials.values()], dtype=L.MATERIAL) if self.materials else np.zeros(0, dtype=L.MATERIAL)

    def desc(self):
        """MiptSceneDesc over this scene's arrays (keeps the backing arrays alive on the returned object)."""
        mats = np.ascontiguousarray(self.materials_array())
        texs = (L.MiptTexture * max(len(self.textures), 1))()
        for i, t in enumerate(self.textures):
            texs[i].width, texs[i].height, texs[i].rgba8 = t.shape[1], t.shape[0], t.ctypes.data
        d = L.MiptSceneDesc(L.ptr(self.tris), len(self.tris), L.ptr(self.bvh_nodes), len(self.bvh_nodes),
                            L.ptr(mats), len(mats), texs, len(self.textures))
        d._keep = (mats, texs)
        return d

    def upload(self, device_id: int = 0) -> C.c_void_p:
        if self._handle is not None and self._handle_device == device_id:
            return self._handle
        self.release()
        h = C.c_void_p()
        d = self.desc()
        L.check(L.load().mipt_scene_create(C.byref(d), device_id, C.byref(h)), "mipt_scene_create")
        self._handle, self._handle_device = h, device_id
        return h

    def host_layout(self):
        """TEST INFRASTRUCTURE (libmipt_diag.so, tests/cpp/host_layout.cpp): the geometry buffers as rounds 1-3 laid them out on the
        host from this Scene's triangles and nodes -- the byte-for-byte reference of the device layout kernels both entries now use.
        Returns (geom bytes, attr bytes, dict(n_pair_records, max_leaf, root_a, root_n))."""
        diag = L.load_diag()
        d = self.desc()
        sizes, info = (C.c_uint64 * 2)(), (C.c_uint32 * 4)()
        rc = diag.mipt_diag_host_layout(C.byref(d), None, 0, None, 0, C.byref(sizes), C.byref(info))
        if rc:
            raise RuntimeError(f"mipt_diag_host_layout failed with status {rc}")
        geom, attr = np.zeros(sizes[0], dtype=np.uint8), np.zeros(sizes[1], dtype=np.uint8)
        rc = diag.mipt_diag_host_layout(C.byref(d), geom.ctypes.data, sizes[0], attr.ctypes.data, sizes[1], C.byref(sizes), C.byref(info))
        if rc:
            raise RuntimeError(f"mipt_diag_host_layout failed with status {rc}")
        return geom, attr, dict(n_pair_records=int(info[0]), max_leaf=int(info[1]), root_a=int(info[2]), root_n=int(info[3]))

    def host_layout_fingerprint(self):
        """(hash of geom, hash of attr, bytes of geom, bytes of attr) of host_layout(), comparable with mipt_diag_scene_hash / _sizes."""
        geom, attr, _ = self.host_layout()
        diag = L.load_diag()
        hg, ha = C.c_uint64(), C.c_uint64()
        assert diag.mipt_diag_hash_words(geom.ctypes.data, geom.size // 4, C.byref(hg)) == 0
        assert diag.mipt_diag_hash_words(attr.ctypes.data, attr.size // 4, C.byref(ha)) == 0
        return int(hg.value), int(ha.value), int(geom.size), int(attr.size)

    def upload_from_triangles(self, device_id: int = 0, fetch_bvh: bool = False) -> C.c_void_p:
        """BVH::build + upload in ONE call, everything after the triangle copy on the GPU (mipt_scene_create_from_triangles): the
        scene's triangles go up in their current order, the tree is built and laid out in HBM.  With ``fetch_bvh`` the Scene is
        left as BVH::build leaves the reference's (bvh.rs:13-54): ``bvh_nodes`` filled and ``tris`` reordered."""
        self.release()
        h = C.c_void_p()
        d = self.desc()
        lib = L.load()
        L.check(lib.mipt_scene_create_from_triangles(C.byref(d), device_id, C.byref(h)), "mipt_scene_create_from_triangles")
        self._handle, self._handle_device = h, device_id
        if fetch_bvh:
            self._fetch_bvh(h)
        return h

    def _fetch_bvh(self, handle) -> None:
        n = len(self.tris)
        nodes = np.zeros(max(2 * n, 1), dtype=L.NODE)
        order = np.zeros(n, dtype=np.uint32)
        count = C.c_uint32(0)
        L.check(L.load().mipt_scene_get_bvh(handle, L.ptr(nodes), len(nodes), C.byref(count), L.ptr(order)), "mipt_scene_get_bvh")
        self.bvh_nodes = nodes[: count.value].copy()
        self.tris = self.tris[order]

    def info(self, handle=None) -> dict:
        """MiptSceneInfo of the resident scene (sizes + what the setup took)."""
        h = handle if handle is not None else self._handle
        if h is None:
            raise RuntimeError("scene is not resident on a device")
        inf = L.MiptSceneInfo()
        L.check(L.load().mipt_scene_info(h, C.byref(inf)), "mipt_scene_info")
        return inf.as_dict()

    def release(self) -> None:
        if self._handle is not None:
            L.load().mipt_scene_destroy(self._handle)
            self._handle = None
        if getattr(self, "_multi", None) is not None:
            L.load().mipt_multi_destroy(self._multi)
            self._multi = None

    def upload_multi(self, device_ids=None, from_triangles: bool = False, fetch_bvh: bool = False) -> C.c_void_p:
        """One replica + RCCL communicator per device (mipt_multi_create); device_ids None = every visible device.  The scene crosses
        PCIe once (to the first device), the other replicas are device-to-device copies.  ``from_triangles``: the BVH is built on
        that first device (mipt_multi_create_from_triangles); ``fetch_bvh`` then leaves nodes + reordered triangles in this Scene."""
        key = (None if device_ids is None else tuple(device_ids), from_triangles)
        if getattr(self, "_multi", None) is not None and self._multi_key == key:
            return self._multi
        if getattr(self, "_multi", None) is not None:
            L.load().mipt_multi_destroy(self._multi)
            self._multi = None
        h = C.c_void_p()
        d = self.desc()
        ids = None if device_ids is None else (C.c_int * len(device_ids))(*device_ids)
        lib = L.load()
        if from_triangles:
            L.check(lib.mipt_multi_create_from_triangles(C.byref(d), ids, 0 if device_ids is None else len(device_ids), C.byref(h)), "mipt_multi_create_from_triangles")
            if fetch_bvh:
                self._fetch_bvh(lib.mipt_multi_scene(h, 0))
        else:
            L.check(lib.mipt_multi_create(C.byref(d), ids, 0 if device_ids is None else len(device_ids), C.byref(h)), "mipt_multi_create")
        self._multi, self._multi_key = h, key
        return h

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def make_options(width, height, samples, max_ray_depth, seed_mode=L.SEED_PIXEL_STREAM, traversal=L.TRAVERSAL_REFERENCE,
                 flags=0, tile_rank=0, tile_world=0, sample_begin=0, cull_margin=L.CULL_MARGIN_SAFE, shading=0) -> L.MiptOptions:
    o = L.MiptOptions()
    o.width, o.height, o.samples, o.max_ray_depth = width, height, samples, max_ray_depth
    o.seed_mode, o.traversal, o.flags = seed_mode, traversal, flags
    o.tile_rank, o.tile_world, o.sample_begin = tile_rank, tile_world, sample_begin
    o.cull_margin = cull_margin
    o.shading = shading
    return o


class Renderer:  # renderer.rs:8-85
    def __init__(self, options: RendererOptions):
        self.options = options
        self.last_stats: Optional[dict] = None

    @staticmethod
    def new(options: RendererOptions) -> Optional["Renderer"]:  # renderer.rs:14-48
        w, h = options.output_image_dimensions
        if w == 0 or h == 0:
            log_error("Width and height must be greater than 0")
            return None
        if options.max_ray_depth == 0:
            log_error("Max ray depth must be greater than 0")
            return None
        if options.samples == 0:
            log_error("Sample count must be greater than 0")
            return None
        if options.output_image_path is None and not options.is_realtime:
            log_error("Output image path must be Some if realtime mode is disabled")
            return None
        if options.backend != RendererBackend.GPU and options.is_realtime:
            log_error("Only the GPU backend is supported for realtime mode")
            return None
        return Renderer(options)

    def render_buffers(self, scene: Scene, want_hdr: bool = True, want_rgba8: bool = True, flags: int = 0):
        """The backend arm: (Renderer, &Scene) -> pixels.  Returns (hdr float32 [h,w,3] | None,
        rgba8 uint8 [h,w,4] | None, stats dict)."""
        o = self.options
        if o.backend != RendererBackend.MI355X:
            raise NotImplementedError(f"backend {o.backend.name} is not part of this build; use RendererBackend.MI355X")
        w, h = o.output_image_dimensions
        handle = scene.upload(o.device_id)
        opt = make_options(w, h, o.samples, o.max_ray_depth, o.seed_mode, o.traversal, flags, cull_margin=o.cull_margin, shading=o.shading)
        hdr = np.zeros((h, w, 3), dtype=np.float32) if want_hdr else None
        rgba = np.zeros((h, w, 4), dtype=np.uint8) if want_rgba8 else None
        st = L.MiptStats()
        rc = L.load().mipt_render(handle, L.ptr(scene.camera.uniform), C.byref(opt),
                                  L.ptr(hdr) if want_hdr else None, L.ptr(rgba) if want_rgba8 else None, C.byref(st))
        L.check(rc, "mipt_render")
        self.last_stats = st.as_dict()
        return hdr, rgba, self.last_stats

    def render_buffers_multi(self, scene: Scene, mode: int = L.MULTI_TILES, device_ids=None, want_hdr: bool = True,
                             want_rgba8: bool = True, flags: int = 0):
        """The same arm over all GPUs of the node in one call (mipt_render_multi): image tiles + one RCCL gather, or
        sample ranges + one RCCL sum-reduce.  Returns (hdr, rgba8, stats dict)."""
        o = self.options
        w, h = o.output_image_dimensions
        multi = scene.upload_multi(device_ids)
        opt = make_options(w, h, o.samples, o.max_ray_depth, o.seed_mode, o.traversal, flags, cull_margin=o.cull_margin, shading=o.shading)
        hdr = np.zeros((h, w, 3), dtype=np.float32) if want_hdr else None
        rgba = np.zeros((h, w, 4), dtype=np.uint8) if want_rgba8 else None
        st = L.MiptMultiStats()
        rc = L.load().mipt_render_multi(multi, L.ptr(scene.camera.uniform), C.byref(opt), mode,
                                        L.ptr(hdr) if want_hdr else None, L.ptr(rgba) if want_rgba8 else None, C.byref(st))
        L.check(rc, "mipt_render_multi")
        self.last_stats = st.as_dict()
        return hdr, rgba, self.last_stats

    def render(self, scene: Scene) -> bytes:  # renderer.rs:50-85 (offline arm)
        if self.options.is_realtime:
            raise NotImplementedError("the realtime window (gpu/window.rs) is out of scope")
        _, rgba, stats = self.render_buffers(scene, want_hdr=False, want_rgba8=True)
        log_info(f"Rendering took {stats['kernel_ms']:.1f} ms")
        path = self.options.output_image_path
        if path:
            w, h = self.options.output_image_dimensions
            write_ppm(path, rgba[:, :, :3]) if path.endswith(".ppm") else write_png_rgba8(path, rgba)
            log_info(f"Succesfully wrote image data to '{path}'")
        return rgba.tobytes()


def write_ppm(path: str, rgb: np.ndarray) -> None:
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())


def write_png_rgba16(path: str, rgba16: np.ndarray) -> None:
    """RGBA16 PNG (big-endian samples) -- the format Renderer::render saves (renderer.rs:67-73, ColorType::Rgba16)."""
    import struct
    import zlib
    h, w, _ = rgba16.shape
    be = np.ascontiguousarray(rgba16, dtype=np.uint16).astype(">u2")
    raw = b"".join(b"\x00" + be[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_png_rgba8(path: str, rgba: np.ndarray) -> None:
    """Minimal PNG writer (zlib + CRC from the stdlib).  The reference saves through image::save_buffer
    (renderer.rs:67-73) as Rgba16 -- which cannot hold the CPU path's RGBA8 bytes; this
    writes the RGBA8 pixels the CPU path actually produces."""
    import struct
    import zlib
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + np.ascontiguousarray(rgba[y]).tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

"""ctypes binding of libmipt.so (include/mipt.h) -- the C ABI is the product boundary.

The library is built in-tree by ``__graft_entry__.build()`` (``make -C rust_ray_tracing_amd/csrc``).
There is no fallback: if the shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIPT_LIB") or os.path.join(_HERE, "libmipt.so")   # MIPT_LIB: A/B a kernel variant

# ---- numpy views of the reference PODs (reference src/scene.rs:87-146, src/bvh.rs:164-171) ----
VERTEX = np.dtype([("position", "<f4", 3), ("tex_coord_x", "<f4"), ("normal", "<f4", 3), ("tex_coord_y", "<f4")])
TRIANGLE = np.dtype([("vertices", VERTEX, 3), ("material_id", "<u4"), ("_pad", "u1", 12)])
NODE = np.dtype([("bounds_min", "<f4", 3), ("first_tri_or_child", "<u4"), ("bounds_max", "<f4", 3), ("num_tris", "<u4")])
MATERIAL = np.dtype([
    ("base_color", "<f4", 3), ("transmission", "<f4"), ("specular_tint", "<f4", 3), ("ior", "<f4"),
    ("emission", "<f4", 3), ("roughness", "<f4"), ("metallic", "<f4"), ("transparency", "<f4"),
    ("base_color_tex_id", "<u4"), ("transparency_tex_id", "<u4"), ("roughness_tex_id", "<u4"),
    ("metallic_tex_id", "<u4"), ("emission_tex_id", "<u4"), ("normal_tex_id", "<u4")])
CAMERA = np.dtype([("look_at", "<f4", (4, 4)), ("position", "<f4", 3), ("_pad", "<f4")])
assert VERTEX.itemsize == 32 and TRIANGLE.itemsize == 112 and NODE.itemsize == 32
assert MATERIAL.itemsize == 80 and CAMERA.itemsize == 80

NO_TEXTURE = 0xFFFFFFFF

SEED_PIXEL_STREAM, SEED_PER_SAMPLE = 0, 1
TRAVERSAL_REFERENCE, TRAVERSAL_CULLED = 0, 1
SHADING_CPU, SHADING_WGPU = 0, 1
FLAG_COUNT, FLAG_PACKED, FLAG_SUM, FLAG_ACCUM, FLAG_TOUCHED = 1, 2, 4, 8, 16
CULL_MARGIN_SAFE = 0.0078125  # MIPT_CULL_MARGIN_SAFE

OK, ERR_INVALID_ARG, ERR_HIP, ERR_SCENE_LIMIT, ERR_BVH, ERR_IO, ERR_STACK, ERR_RCCL = 0, -1, -2, -3, -4, -5, -6, -7
MULTI_TILES, MULTI_SAMPLES = 0, 1


class MiptTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p)]


class MiptSceneDesc(C.Structure):
    _fields_ = [("tris", C.c_void_p), ("n_tris", C.c_uint32),
                ("nodes", C.c_void_p), ("n_nodes", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("textures", C.POINTER(MiptTexture)), ("n_textures", C.c_uint32)]


class MiptOptions(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("max_ray_depth", C.c_uint32), ("seed_mode", C.c_uint32), ("traversal", C.c_uint32),
                ("flags", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32),
                ("sample_begin", C.c_uint32), ("cull_margin", C.c_float), ("shading", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class MiptStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("rays", C.c_uint64), ("inner_steps", C.c_uint64),
                ("tri_tests", C.c_uint64), ("hits", C.c_uint64), ("texel_fetches", C.c_uint64),
                ("stack_overflows", C.c_uint64), ("tex_clamped", C.c_uint64), ("max_stack", C.c_uint64),
                ("pixels", C.c_uint64), ("diag", C.c_uint64 * 11), ("touched_lines", C.c_uint64 * 2)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("diag", "touched_lines")}
        d["diag"] = list(self.diag)
        d["touched_lines"] = list(self.touched_lines)
        return d


class MiptSceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_nodes", C.c_uint32), ("n_pair_records", C.c_uint32), ("max_leaf", C.c_uint32),
                ("geometry_bytes", C.c_uint64), ("built_on_device", C.c_uint32), ("replica_of_device", C.c_uint32),
                ("upload_ms", C.c_double), ("build_ms", C.c_double), ("layout_ms", C.c_double), ("total_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class MiptMultiStats(C.Structure):
    _fields_ = [("total", MiptStats), ("collective_ms", C.c_double), ("wall_ms", C.c_double),
                ("device_kernel_ms", C.c_double * 8), ("n_devices", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        d = self.total.as_dict()
        d.update(collective_ms=self.collective_ms, wall_ms=self.wall_ms, n_devices=self.n_devices,
                 device_kernel_ms=list(self.device_kernel_ms)[: self.n_devices])
        return d


# every symbol include/mipt.h declares (tests/test_abi.py checks the list against the header)
EXPORTS = [
    "mipt_scene_create", "mipt_scene_destroy", "mipt_render", "mipt_render_device",
    "mipt_packed_pixels", "mipt_unpack_tiles", "mipt_tonemap_device", "mipt_postprocess_device", "mipt_bvh_build", "mipt_bvh_build_device",
    "mipt_camera_from_pose", "mipt_material_default", "mipt_last_error", "mipt_abi_version",
    "mipt_device_count", "mipt_obj_load", "mipt_obj_free", "mipt_obj_get",
    "mipt_texture_load", "mipt_texture_free", "mipt_image_save_png",
    "mipt_multi_create", "mipt_multi_destroy", "mipt_multi_device_count", "mipt_render_multi",
    "mipt_render_multi_device", "mipt_multi_root_device", "mipt_multi_device_stats",
    "mipt_scene_create_from_triangles", "mipt_scene_get_bvh", "mipt_scene_info", "mipt_multi_create_from_triangles", "mipt_multi_scene", "mipt_obj_load_triangles",
]

_lib = None


def load() -> C.CDLL:
    """Load libmipt.so once.  torch (if importable) is imported first so that libmipt binds to
    the HIP runtime torch already loaded (same SONAME libamdhip64.so.7): device pointers of torch
    tensors are then valid in our launches."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C rust_ray_tracing_amd/csrc). "
                          "There is no CPU fallback for the MI355X backend.")
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    _lib = _bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    return _lib


def _bind(lib: C.CDLL) -> C.CDLL:
    """argtypes / restype of every include/mipt.h entry point on a loaded library (the product or a build variant of it)."""
    vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    lib.mipt_scene_create.argtypes = [C.POINTER(MiptSceneDesc), C.c_int, C.POINTER(vp)]
    lib.mipt_scene_create.restype = C.c_int
    lib.mipt_scene_destroy.argtypes = [vp]
    lib.mipt_scene_destroy.restype = None
    lib.mipt_render.argtypes = [vp, vp, C.POINTER(MiptOptions), vp, vp, C.POINTER(MiptStats)]
    lib.mipt_render.restype = C.c_int
    lib.mipt_render_device.argtypes = [vp, vp, C.POINTER(MiptOptions), vp, vp, vp, C.POINTER(MiptStats)]
    lib.mipt_render_device.restype = C.c_int
    lib.mipt_packed_pixels.argtypes = [u32, u32, u32]
    lib.mipt_packed_pixels.restype = u64
    lib.mipt_unpack_tiles.argtypes = [vp, u32, u32, u32, vp, vp]
    lib.mipt_unpack_tiles.restype = C.c_int
    lib.mipt_tonemap_device.argtypes = [vp, u64, f32, vp, vp]
    lib.mipt_tonemap_device.restype = C.c_int
    lib.mipt_postprocess_device.argtypes = [vp, u64, f32, vp, vp]
    lib.mipt_postprocess_device.restype = C.c_int
    lib.mipt_bvh_build.argtypes = [vp, u32, vp, u32, C.POINTER(u32), u32]
    lib.mipt_bvh_build.restype = C.c_int
    lib.mipt_bvh_build_device.argtypes = [vp, u32, vp, u32, C.POINTER(u32), C.c_int, C.POINTER(C.c_double)]
    lib.mipt_bvh_build_device.restype = C.c_int
    lib.mipt_camera_from_pose.argtypes = [C.POINTER(f32 * 3), f32, f32, vp]
    lib.mipt_camera_from_pose.restype = C.c_int
    lib.mipt_material_default.argtypes = [vp]
    lib.mipt_material_default.restype = None
    lib.mipt_last_error.argtypes = []
    lib.mipt_last_error.restype = C.c_char_p
    lib.mipt_abi_version.argtypes = []
    lib.mipt_abi_version.restype = C.c_int
    lib.mipt_device_count.argtypes = []
    lib.mipt_device_count.restype = C.c_int
    lib.mipt_obj_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.mipt_obj_load.restype = C.c_int
    lib.mipt_obj_load_triangles.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.mipt_obj_load_triangles.restype = C.c_int
    lib.mipt_obj_free.argtypes = [vp]
    lib.mipt_obj_free.restype = None
    lib.mipt_obj_get.argtypes = [vp, C.POINTER(MiptSceneDesc), C.POINTER(C.POINTER(C.c_char_p))]
    lib.mipt_obj_get.restype = C.c_int
    lib.mipt_texture_load.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(MiptTexture), C.POINTER(u32)]
    lib.mipt_texture_load.restype = C.c_int
    lib.mipt_texture_free.argtypes = [vp]
    lib.mipt_texture_free.restype = None
    lib.mipt_image_save_png.argtypes = [C.c_char_p, u32, u32, u32, vp]
    lib.mipt_image_save_png.restype = C.c_int
    lib.mipt_multi_create.argtypes = [C.POINTER(MiptSceneDesc), vp, C.c_int, C.POINTER(vp)]
    lib.mipt_multi_create.restype = C.c_int
    lib.mipt_multi_destroy.argtypes = [vp]
    lib.mipt_multi_destroy.restype = None
    lib.mipt_multi_device_count.argtypes = [vp]
    lib.mipt_multi_device_count.restype = C.c_int
    lib.mipt_render_multi.argtypes = [vp, vp, C.POINTER(MiptOptions), u32, vp, vp, C.POINTER(MiptMultiStats)]
    lib.mipt_render_multi.restype = C.c_int
    lib.mipt_render_multi_device.argtypes = [vp, vp, C.POINTER(MiptOptions), u32, vp, vp, C.POINTER(MiptMultiStats)]
    lib.mipt_render_multi_device.restype = C.c_int
    lib.mipt_multi_root_device.argtypes = [vp]
    lib.mipt_multi_root_device.restype = C.c_int
    lib.mipt_multi_device_stats.argtypes = [vp, C.c_int, C.POINTER(MiptStats)]
    lib.mipt_multi_device_stats.restype = C.c_int
    lib.mipt_scene_create_from_triangles.argtypes = [C.POINTER(MiptSceneDesc), C.c_int, C.POINTER(vp)]
    lib.mipt_scene_create_from_triangles.restype = C.c_int
    lib.mipt_scene_get_bvh.argtypes = [vp, vp, u32, C.POINTER(u32), vp]
    lib.mipt_scene_get_bvh.restype = C.c_int
    lib.mipt_scene_info.argtypes = [vp, C.POINTER(MiptSceneInfo)]
    lib.mipt_scene_info.restype = C.c_int
    lib.mipt_multi_create_from_triangles.argtypes = [C.POINTER(MiptSceneDesc), vp, C.c_int, C.POINTER(vp)]
    lib.mipt_multi_create_from_triangles.restype = C.c_int
    lib.mipt_multi_scene.argtypes = [vp, C.c_int]
    lib.mipt_multi_scene.restype = vp
    return lib


MULTITEST_LIB_PATH = os.path.join(_HERE, "libmipt_multitest.so")
_multitest = None


def load_multitest() -> C.CDLL:
    """libmipt_multitest.so: the product's objects with mipt_multi.cpp compiled against the RCCL TEST DOUBLE of
    tests/cpp/rccl_double/ (N logical ranks on one device).  Test infrastructure: lets the n > 1 code of mipt_render_multi run on
    the one-GPU box.  Same C ABI (+ rccl_double_inject); its own error string and device state."""
    global _multitest
    if _multitest is not None:
        return _multitest
    load()
    if not os.path.exists(MULTITEST_LIB_PATH):
        raise ImportError(f"{MULTITEST_LIB_PATH} is missing: run __graft_entry__.build()")
    lib = _bind(C.CDLL(MULTITEST_LIB_PATH))
    lib.rccl_double_inject.argtypes = [C.c_int, C.c_int]
    lib.rccl_double_inject.restype = C.c_longlong
    _multitest = lib
    return lib


DIAG_LIB_PATH = os.path.join(_HERE, "libmipt_diag.so")
DIAG_EXPORTS = ["mipt_debug_eval", "mipt_debug_eval_range", "mipt_diag_last_error",
                "mipt_internal_pair_order", "mipt_internal_pair_order_top", "mipt_internal_tri_slots",
                "mipt_diag_scene_sizes", "mipt_diag_scene_read", "mipt_diag_scene_hash", "mipt_diag_write_obj", "mipt_diag_host_layout", "mipt_diag_hash_words"]
_diag = None


def load_diag() -> C.CDLL:
    """libmipt_diag.so (include/mipt_diag.h): the device-arithmetic probe the GPU known-answer tests use and the
    re-exported layout functions (pair-record order, triangle slots).  Test infrastructure; the product library exports none of it."""
    global _diag
    if _diag is not None:
        return _diag
    load()
    if not os.path.exists(DIAG_LIB_PATH):
        raise ImportError(f"{DIAG_LIB_PATH} is missing: run __graft_entry__.build()")
    lib = C.CDLL(DIAG_LIB_PATH)
    vp = C.c_void_p
    lib.mipt_debug_eval.argtypes = [C.c_int, vp, vp, C.c_uint64, vp]
    lib.mipt_debug_eval.restype = C.c_int
    lib.mipt_debug_eval_range.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_float, vp]
    lib.mipt_debug_eval_range.restype = C.c_int
    lib.mipt_diag_last_error.argtypes = []
    lib.mipt_diag_last_error.restype = C.c_char_p
    u32p = C.POINTER(C.c_uint32)
    lib.mipt_internal_pair_order.argtypes = [vp, C.c_uint32, vp, C.c_uint32, u32p]
    lib.mipt_internal_pair_order.restype = C.c_int
    lib.mipt_internal_pair_order_top.argtypes = []
    lib.mipt_internal_pair_order_top.restype = C.c_uint32
    lib.mipt_internal_tri_slots.argtypes = [vp, C.c_uint32, C.c_uint32, vp, u32p]
    lib.mipt_internal_tri_slots.restype = C.c_int
    lib.mipt_diag_scene_sizes.argtypes = [vp, C.POINTER(C.c_uint64 * 2)]
    lib.mipt_diag_scene_sizes.restype = C.c_int
    lib.mipt_diag_scene_read.argtypes = [vp, C.c_int, vp, C.c_uint64]
    lib.mipt_diag_scene_read.restype = C.c_int
    lib.mipt_diag_scene_hash.argtypes = [vp, C.POINTER(C.c_uint64 * 2)]
    lib.mipt_diag_scene_hash.restype = C.c_int
    lib.mipt_diag_write_obj.argtypes = [C.c_char_p, vp, C.c_uint64, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32]
    lib.mipt_diag_write_obj.restype = C.c_int
    lib.mipt_diag_host_layout.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64 * 2), C.POINTER(C.c_uint32 * 4)]
    lib.mipt_diag_host_layout.restype = C.c_int
    lib.mipt_diag_hash_words.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.mipt_diag_hash_words.restype = C.c_int
    _diag = lib
    return lib


class MiptError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = load().mipt_last_error()
        super().__init__(f"{where} failed with status {code}: {msg.decode() if msg else ''}")


def check(code: int, where: str) -> None:
    if code != 0:
        raise MiptError(code, where)


def ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)

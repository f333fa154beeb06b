"""rust_ray_tracing_amd -- MI355X-native path-tracing backend for the MiksuNy/rust_ray_tracing
Renderer/Scene seam.  The product is libmipt.so (hand-written gfx950 HIP behind the C ABI of
include/mipt.h); this package is the host-side mirror of the reference interface plus the ctypes
binding.  Importing it never falls back to a CPU renderer."""
from ._lib import (CAMERA, CULL_MARGIN_SAFE, FLAG_ACCUM, FLAG_COUNT, FLAG_PACKED, FLAG_SUM, FLAG_TOUCHED, MATERIAL, NODE, NO_TEXTURE, SEED_PER_SAMPLE, SHADING_CPU, SHADING_WGPU,
                   SEED_PIXEL_STREAM, TRAVERSAL_CULLED, TRAVERSAL_REFERENCE, TRIANGLE, VERTEX, MiptError,
                   MiptOptions, MiptStats, load, load_diag)
from .host import Camera, Renderer, RendererBackend, RendererOptions, Scene, Texture, make_options, material_default

__all__ = ["Camera", "Renderer", "RendererBackend", "RendererOptions", "Scene", "Texture", "make_options", "material_default",
           "load", "load_diag", "MiptError", "MiptOptions", "MiptStats", "TRIANGLE", "NODE", "MATERIAL", "CAMERA", "VERTEX",
           "NO_TEXTURE", "SEED_PIXEL_STREAM", "SEED_PER_SAMPLE", "TRAVERSAL_REFERENCE", "TRAVERSAL_CULLED",
           "FLAG_COUNT", "FLAG_PACKED", "FLAG_SUM", "FLAG_ACCUM", "FLAG_TOUCHED", "CULL_MARGIN_SAFE", "SHADING_CPU", "SHADING_WGPU"]

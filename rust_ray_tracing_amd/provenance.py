"""Which sources decide what the trace kernel executes and reads, and their hash.  A committed PMC summary under profiles/ is only
quoted by bench.py for the source it was measured on (tools/pmc_summary.py writes the same hash into its provenance line)."""
from __future__ import annotations

import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

# the kernel and its math; the launch parameters and the host-built device layout (mipt_api.cpp); the record orders (bvh_build.cpp);
# the same layout built on the GPU (scene_device.hip)
KERNEL_SOURCES = ("pt_kernel.hip", "pt_device_math.h", "pt_kernel.h", "glibc_flt32_data.h", "mipt_api.cpp", "bvh_build.cpp", "scene_device.hip")


def kernel_source_sha() -> str:
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

"""Index arithmetic of the multi-GPU shards (pure Python/numpy; mirrors pt_kernel.hip and
mipt_unpack_tiles).  The path shards by image tile (config 4: 8x8-pixel tiles dealt round-robin to
ranks, each rank writing a rank-packed slice, one all-gather per frame) or by sample range
(config 5: every rank renders all pixels for a disjoint range of per-sample seeds, one sum-reduce)."""
from __future__ import annotations

import numpy as np

TILE = 8


def tiles(width: int, height: int):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def packed_pixels(width: int, height: int, world: int) -> int:
    """Slots in one rank's packed slice (= mipt_packed_pixels): ceil(tiles / world) * 64."""
    tx, ty = tiles(width, height)
    return ((tx * ty + world - 1) // world) * TILE * TILE


def slot_pixels(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """pixel index held by each packed slot of `rank` (-1 = padding: ragged tile or missing last tile)."""
    tx, ty = tiles(width, height)
    n_slots = packed_pixels(width, height, world)
    slot = np.arange(n_slots, dtype=np.int64)
    lt, p = slot // 64, slot % 64
    gt = lt * world + rank
    px = (gt % tx) * TILE + (p & 7)
    py = (gt // tx) * TILE + (p >> 3)
    ok = (gt < tx * ty) & (px < width) & (py < height)
    return np.where(ok, py * width + px, -1)


def unpack(packed_all: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """[world, slots, C] rank-major slices (an all-gather's output) -> [height*width, C] frame."""
    packed_all = np.asarray(packed_all)
    out = np.zeros((width * height,) + packed_all.shape[2:], dtype=packed_all.dtype)
    for r in range(world):
        pix = slot_pixels(width, height, r, world)
        m = pix >= 0
        out[pix[m]] = packed_all[r][m]
    return out


def sample_ranges(samples: int, world: int):
    """Contiguous per-rank sample ranges [(first_sample_number, count)], sample numbers start at 1
    (reference src/renderer/backend/gpu.rs:252)."""
    base, rem = divmod(samples, world)
    out, s = [], 1
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, n))
        s += n
    return out

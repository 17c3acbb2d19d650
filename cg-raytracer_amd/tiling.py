"""Image tiling across GPUs: the frame is cut into 8x8-pixel tiles (one wave each), grouped
into 64x64-pixel super-tiles; super-tile i (row-major) belongs to rank i % nranks (interleaved, so
silhouette-heavy regions spread over all ranks), the scene is replicated and no data-path collective exists.  Pure host logic shared by bench.py and the world_size-2 gloo tests;
``owned_mask`` restates what the kernel's tile_pixel() computes."""
from __future__ import annotations

import math

import numpy as np

TILE = 8
SUPER = 64  # super-tile side in pixels (8 x 8 tiles)
BASE_W, BASE_H = 1920, 1080  # BASELINE.json: dragon.obj @1920x1080


def frame_for_world(nranks: int, base_w: int = BASE_W, base_h: int = BASE_H):
    """Weak scaling: per-GPU work is fixed, so the frame grows with the number of GPUs -- both sides by
    sqrt(nranks), rounded to whole tiles.  1 -> 1920x1080, 4 -> 3840x2160 (BASELINE.json config 5)."""
    s = math.sqrt(nranks)
    w = int(round(base_w * s / TILE)) * TILE
    h = int(round(base_h * s / TILE)) * TILE
    return w, h


def tiles(W: int, H: int, rect=None):
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
    return (x1 - x0 + TILE - 1) // TILE, (y1 - y0 + TILE - 1) // TILE


def owned_mask(W: int, H: int, rank: int, nranks: int, rect=None) -> np.ndarray:
    """(H, W) bool: pixels rank `rank` traces."""
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, W, H)
    stx = (x1 - x0 + SUPER - 1) // SUPER
    yy, xx = np.mgrid[0:H, 0:W]
    inside = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1)
    st = ((yy - y0) // SUPER) * stx + (xx - x0) // SUPER
    return inside & (st % nranks == rank)


def owned_pixels(W: int, H: int, rank: int, nranks: int, rect=None) -> int:
    return int(owned_mask(W, H, rank, nranks, rect).sum())

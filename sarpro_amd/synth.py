"""Synthetic dual-pol GRD-like scenes -- bench / test support.

Counter-based and integer-only, so the numpy generator here and the HIP generator
(`sarpro_hip_synth_scene_u16_dev`, csrc/kernels.hip) agree bit for bit:

    h     = splitmix64(seed ^ (band << 60) ^ idx)          idx = row * cols + col
    class = ((row // block) + 3 * (col // block)) % 4      sigma scale {1/4, 1, 2, 4}
    DN    = Q[band][class][h >> 48]                        Rayleigh inverse-CDF tables
    DN    = 20000 + (h & 0x7FFF)   where (h >> 20) % 10000 == 0   (bright targets)
    DN    = 0  inside the two no-data wedges (about 3 % of the scene, same for both bands)

block = ceil(rows / 16) (1250 for the 20000 x 20000 scene).
"""
from __future__ import annotations

import numpy as np

SEED_SCENE_A = 0x535250524F01
SIGMA = (180.0, 70.0)          # VV, VH base Rayleigh sigma in DN
CLASS_SCALE = (0.25, 1.0, 2.0, 4.0)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def q_tables(flat: bool = False) -> np.ndarray:
    """(2, 4, 65536) uint16 inverse-CDF tables.  flat=True: one class (scale 1) everywhere."""
    k = (np.arange(65536, dtype=np.float64) + 0.5) / 65536.0
    base = np.sqrt(-2.0 * np.log1p(-k))
    out = np.empty((2, 4, 65536), dtype=np.uint16)
    for b in range(2):
        for c in range(4):
            s = SIGMA[b] * (1.0 if flat else CLASS_SCALE[c])
            out[b, c] = np.clip(np.rint(base * s), 1, 65535).astype(np.uint16)
    return out


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return z ^ (z >> np.uint64(31))


def scene_u16(rows: int, cols: int, band: int, seed: int = SEED_SCENE_A, q: np.ndarray | None = None,
              row0: int = 0, rows_local: int | None = None) -> np.ndarray:
    """Rows [row0, row0+rows_local) of band `band` of the rows x cols scene, as uint16."""
    if q is None:
        q = q_tables()
    if rows_local is None:
        rows_local = rows - row0
    block = max((rows + 15) // 16, 1)
    r = np.arange(row0, row0 + rows_local, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    idx = r * np.uint64(cols) + c
    key = np.uint64(seed) ^ (np.uint64(band) << np.uint64(60))
    h = splitmix64(idx ^ key)
    cls = ((r // np.uint64(block)) + np.uint64(3) * (c // np.uint64(block))) % np.uint64(4)
    dn = q[band][cls.astype(np.intp), (h >> np.uint64(48)).astype(np.intp)]
    bright = ((h >> np.uint64(20)) % np.uint64(10000)) == 0
    dn = np.where(bright, (np.uint64(20000) + (h & np.uint64(0x7FFF))).astype(np.uint16), dn)
    R, C = np.uint64(rows), np.uint64(cols)
    left = c * R * np.uint64(100) < np.uint64(3) * C * (R - r)
    right = (C - np.uint64(1) - c) * R * np.uint64(100) < np.uint64(3) * C * r
    dn = np.where(left | right, np.uint16(0), dn)
    return np.ascontiguousarray(dn.astype(np.uint16))

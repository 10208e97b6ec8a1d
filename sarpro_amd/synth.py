"""Synthetic dual-pol GRD-like scenes -- bench / test support.

Counter-based and integer-only, so the numpy generator here and the HIP generator
(`sarpro_hip_synth_scene_u16_dev`, csrc/kernels.hip) agree bit for bit:

    h     = splitmix64(seed ^ (band << 60) ^ idx)          idx = row * cols + col
    class = ((row // block) + 3 * (col // block)) % 4      sigma scale {1/4, 1, 2, 4}
    DN    = Q[band][class][h >> 48]                        Rayleigh inverse-CDF tables
    DN    = 20000 + (h & 0x7FFF)   where (h >> 20) % 10000 == 0   (bright targets)
    DN    = 0  inside the two no-data wedges (about 3 % of the scene, same for both bands)

block = ceil(rows / 16) (1250 for the 20000 x 20000 scene).

`flags` (sarpro_hip_synth_scene_u16_dev_ex) vary the STRUCTURE of a scene: NO_WEDGE (no invalid pixel anywhere), NO_BRIGHT,
class_map(m) (0: the map above; 1: (row // block) xor (col // block); 2: diagonal bands ((row + col) // block); 3: one class),
blocks(n) class blocks per side instead of 16.  q_tables(sigma=..., scales=...) vary the amplitude statistics.  BENCH_SCENES is
the set bench.py cycles its timed steps over.
"""
from __future__ import annotations

import numpy as np

SEED_SCENE_A = 0x535250524F01
SIGMA = (180.0, 70.0)          # VV, VH base Rayleigh sigma in DN
CLASS_SCALE = (0.25, 1.0, 2.0, 4.0)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


NO_WEDGE, NO_BRIGHT, NO_BAND2 = 1, 2, 4


def class_map(m: int) -> int:
    return (m & 15) << 4


def blocks(n: int) -> int:
    return (n & 255) << 8


def q_tables(flat: bool = False, sigma=SIGMA, scales=CLASS_SCALE) -> np.ndarray:
    """(2, 4, 65536) uint16 inverse-CDF tables.  flat=True: one class (scale 1) everywhere."""
    k = (np.arange(65536, dtype=np.float64) + 0.5) / 65536.0
    base = np.sqrt(-2.0 * np.log1p(-k))
    out = np.empty((2, 4, 65536), dtype=np.uint16)
    for b in range(2):
        for c in range(4):
            s = sigma[b] * (1.0 if flat else scales[c])
            out[b, c] = np.clip(np.rint(base * s), 1, 65535).astype(np.uint16)
    return out


# The scenes bench.py cycles its timed steps over: (name, seed offset, flags, q_tables keyword arguments, what it is there for).
# Scene A first (the scene of rounds 1-3, so that rounds stay comparable).  Three of them are there to MISS the speculative fused
# route: no invalid pixel (level 0 absent: the u8 rescale of autoscale.rs:348-364 is not provably the identity), amplitude windows
# beyond the fused pass's LDS pool, a band without a valid sample (no level 255).
BENCH_SCENES = (
    ("A", 0, 0, {}, "scene A of rounds 1-3: four sigma classes in 16 x 16 blocks, no-data wedges, bright targets"),
    ("B-xor-map", 1, class_map(1), {"sigma": (140.0, 55.0)}, "other class map and sigma set"),
    ("C-diagonal", 2, class_map(2) | blocks(11), {"sigma": (220.0, 90.0), "scales": (0.5, 1.0, 1.5, 3.0)}, "diagonal class bands, other scales"),
    ("D-no-wedge", 3, NO_WEDGE, {}, "no invalid pixel anywhere: level 0 has to come from the darkest valid samples for the identity of the u8 rescale to be proven"),
    ("E-wide-windows", 4, class_map(1), {"sigma": (420.0, 260.0)}, "p99 windows of VV + VH beyond the fused pass's 3072-entry pool: the pass's WIDE form (DN -> bin byte table)"),
    ("F-flat", 5, class_map(3), {"sigma": (160.0, 60.0)}, "one class everywhere (single-Rayleigh scene: IQR < 5 dB)"),
    ("G-no-bright", 6, class_map(1) | blocks(7) | NO_BRIGHT, {"sigma": (120.0, 80.0), "scales": (0.3, 1.0, 2.5, 5.0)}, "large blocks, no bright targets"),
    ("H-quantised-VH", 7, 0, {"sigma": (180.0, 2.5)}, "VH amplitudes of a few DN: heavily quantised band, a dozen occupied CLAHE bins"),
    ("I-no-VH", 8, NO_BAND2, {}, "VH band without a valid sample (a missing polarisation): level 0 everywhere, its u8 rescale is the identity by max == min (autoscale.rs:356)"),
)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return z ^ (z >> np.uint64(31))


def scene_u16(rows: int, cols: int, band: int, seed: int = SEED_SCENE_A, q: np.ndarray | None = None,
              row0: int = 0, rows_local: int | None = None, flags: int = 0) -> np.ndarray:
    """Rows [row0, row0+rows_local) of band `band` of the rows x cols scene, as uint16."""
    if q is None:
        q = q_tables()
    if rows_local is None:
        rows_local = rows - row0
    per_side = ((flags >> 8) & 255) or 16
    block = max((rows + per_side - 1) // per_side, 1)
    r = np.arange(row0, row0 + rows_local, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    idx = r * np.uint64(cols) + c
    key = np.uint64(seed) ^ (np.uint64(band) << np.uint64(60))
    h = splitmix64(idx ^ key)
    B, m = np.uint64(block), (flags >> 4) & 15
    if m == 0:
        cls = ((r // B) + np.uint64(3) * (c // B)) % np.uint64(4)
    elif m == 1:
        cls = ((r // B) ^ (c // B)) % np.uint64(4)
    elif m == 2:
        cls = ((r + c) // B) % np.uint64(4)
    else:
        cls = np.ones(np.broadcast_shapes(r.shape, c.shape), np.uint64)
    dn = q[band][cls.astype(np.intp), (h >> np.uint64(48)).astype(np.intp)]
    if not flags & NO_BRIGHT:
        bright = ((h >> np.uint64(20)) % np.uint64(10000)) == 0
        dn = np.where(bright, (np.uint64(20000) + (h & np.uint64(0x7FFF))).astype(np.uint16), dn)
    if not flags & NO_WEDGE:
        R, C = np.uint64(rows), np.uint64(cols)
        left = c * R * np.uint64(100) < np.uint64(3) * C * (R - r)
        right = (C - np.uint64(1) - c) * R * np.uint64(100) < np.uint64(3) * C * r
        dn = np.where(left | right, np.uint16(0), dn)
    if flags & NO_BAND2 and band == 1:
        dn = np.zeros_like(dn)
    return np.ascontiguousarray(dn.astype(np.uint16))

"""A SECOND, independent restatement of the reference's raster core -- pure Python, written from the Rust source
(/root/reference/src/core/processing/{pipeline,autoscale,ops,synthetic_rgb}.rs), NOT from oracle/sarpro_oracle.c.

TEST INFRASTRUCTURE.  Its only use: tests/test_pyref_vs_oracle.py compares it with the C oracle on small inputs, so that the
oracle is no longer one author's single transcription of autoscale.rs ("parity unpinned": the reference has no
vectors and cannot be built here; this does not pin it either, it removes the single-transcription risk).

Everything is a per-pixel loop in the reference's own order, on Python floats (IEEE f64) and explicitly rounded f32 values.
libm comes from glibc through ctypes (`log10`, `pow`, `powf`: what Rust's f64::log10 / f64::powf / f32::powf lower to on
Linux) -- never numpy's `power` (88 of the 512 LUT entries differ from glibc's powf in the last ulp).

Rust semantics spelled out here:
  * `x as u8/u16/u32/usize/isize` from a float: truncate toward zero, saturate, NaN -> 0          (as_int)
  * `f32::round` / `f64::round`: half away from zero                                               (rround)
  * `f64::max` / `f64::min`: a NaN operand is ignored                                              (fmax / fmin)
  * `x.clamp(lo, hi)`: `if x < lo { lo } else if x > hi { hi } else { x }` (NaN stays NaN)         (clamp)
  * f32 arithmetic: every operation rounded to f32 (an f64 operation on f32 operands followed by one rounding to f32 is the
    correctly rounded f32 result for + - * /: 53 >= 2 * 24 + 2)                                     (f32)
"""
from __future__ import annotations

import ctypes
import math
import struct

_libm = ctypes.CDLL("libm.so.6")
_libm.log10.restype = ctypes.c_double
_libm.log10.argtypes = [ctypes.c_double]
_libm.pow.restype = ctypes.c_double
_libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
_libm.powf.restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]

STANDARD, ROBUST, ADAPTIVE, EQUALIZED, CLAHE, TAMED, DEFAULT = range(7)  # types.rs:115-123
U8, U16 = 0, 1                                                           # types.rs:170-173
SUM, DIFF, RATIO, NDIFF, LOGRATIO = range(5)                             # types.rs:8-14
F64_EPSILON = 2.220446049250313e-16


def f32(x: float) -> float:
    """x rounded to the nearest f32 (ties to even), returned as a Python float; inf / nan pass through."""
    if x != x or x in (math.inf, -math.inf):
        return x
    try:
        return struct.unpack("<f", struct.pack("<f", x))[0]
    except OverflowError:  # beyond f32's range: rounds to infinity
        return math.copysign(math.inf, x)


def powf(x: float, y: float) -> float:
    return float(_libm.powf(ctypes.c_float(x), ctypes.c_float(y)))


def as_int(x: float, lo: int, hi: int) -> int:
    if x != x:
        return 0
    if x <= lo:
        return lo
    if x >= hi:
        return hi
    return int(x)  # truncates toward zero


def rround(x: float) -> float:
    if x != x or x in (math.inf, -math.inf):
        return x
    f = math.floor(abs(x))
    r = f + 1.0 if abs(x) - f >= 0.5 else f
    return math.copysign(r, x)


def fmax(a: float, b: float) -> float:
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def fmin(a: float, b: float) -> float:
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


def clamp(x: float, lo: float, hi: float) -> float:
    if x < lo:
        return lo
    if x > hi:
        return hi
    return x


# ------------------------------------------------------------------------------------------------ pipeline.rs:8-40
def process_scalar_data_inplace(src):
    """src: flat list of f32 values -> (db list of f64, mask list of bool)"""
    db, mask = [], []
    for v in src:
        magnitude = fmax(float(v), 1e-10)          # (v as f64).max(1e-10)
        db_val = 10.0 * _libm.log10(magnitude)
        db.append(db_val)
        mask.append(db_val > -50.0)
    return db, mask


# ------------------------------------------------------------------------------------------------ autoscale.rs:35-160
class Stats:
    __slots__ = ("valid_count", "min_db", "max_db", "mean_db", "std_db", "median_db", "p01", "p02", "p05", "p10", "p25", "p75",
                 "p90", "p95", "p98", "p99")


def compute_histogram_stats(db, mask) -> Stats:
    s = Stats()
    count = 0
    min_db, max_db = math.inf, -math.inf
    mean = m2 = 0.0
    for v, ok in zip(db, mask):
        if ok:
            count += 1
            if v < min_db:
                min_db = v
            if v > max_db:
                max_db = v
            delta = v - mean
            mean += delta / float(count)
            delta2 = v - mean
            m2 += delta * delta2
    names = ("median_db", "p01", "p02", "p05", "p10", "p25", "p75", "p90", "p95", "p98", "p99")
    if count == 0:
        s.valid_count = 0
        s.min_db = s.max_db = s.mean_db = s.std_db = 0.0
        for n in names:
            setattr(s, n, 0.0)
        return s
    std_db = math.sqrt(m2 / float(count)) if count > 1 else 0.0
    s.valid_count, s.min_db, s.max_db, s.mean_db, s.std_db = count, min_db, max_db, mean, std_db
    if abs(max_db - min_db) < F64_EPSILON:
        for n in ("median_db", "p01", "p02", "p05", "p10", "p25"):
            setattr(s, n, min_db)
        for n in ("p75", "p90", "p95", "p98", "p99"):
            setattr(s, n, max_db)
        return s
    NUM_BINS = 4096
    hist = [0] * NUM_BINS
    span = max_db - min_db
    inv_span = 1.0 / span
    for v, ok in zip(db, mask):
        if not ok:
            continue
        t = clamp((v - min_db) * inv_span, 0.0, 1.0)
        idx = as_int(t * float(NUM_BINS), 0, (1 << 64) - 1)
        if idx >= NUM_BINS:
            idx = NUM_BINS - 1
        hist[idx] += 1

    def estimate_percentile(p: float) -> float:
        n = count
        target = as_int(math.floor(p * float(n)), 0, (1 << 64) - 1)
        if target >= n:
            target = n - 1
        cumsum = 0
        for b, h in enumerate(hist):
            nxt = cumsum + h
            if target < nxt:
                within = target - cumsum if target >= cumsum else 0  # saturating_sub
                frac = float(within) / float(h) if h > 0 else 0.0
                bin_width = span / float(NUM_BINS)
                bin_start = min_db + float(b) * bin_width
                return bin_start + frac * bin_width
            cumsum = nxt
        return max_db

    for n, p in zip(names, (0.5, 0.01, 0.02, 0.05, 0.10, 0.25, 0.75, 0.90, 0.95, 0.98, 0.99)):
        setattr(s, n, estimate_percentile(p))
    return s


# ------------------------------------------------------------------------------------------------ autoscale.rs:220-345
class ClaheUnderflow(Exception):
    """`r1 - r0` (or `c1 - c0`) underflows usize: the reference panics (debug) or wraps (release) for this shape."""


def clahe_equalize_normalized(norm, mask, rows, cols, tiles_x=8, tiles_y=8, clip_limit=2.0, num_bins=256):
    if rows == 0 or cols == 0 or tiles_x == 0 or tiles_y == 0 or num_bins < 2:
        return list(norm)
    tile_h = (rows + tiles_y - 1) // tiles_y
    tile_w = (cols + tiles_x - 1) // tiles_x
    cdfs = [[0.0] * num_bins for _ in range(tiles_x * tiles_y)]
    for ty in range(tiles_y):
        r0 = ty * tile_h
        r1 = min((ty + 1) * tile_h, rows)
        if r1 < r0:
            raise ClaheUnderflow()
        tile_rows = r1 - r0
        for tx in range(tiles_x):
            c0 = tx * tile_w
            c1 = min((tx + 1) * tile_w, cols)
            if c1 < c0:
                raise ClaheUnderflow()
            tile_cols = c1 - c0
            hist = [0] * num_bins
            for r in range(r0, r1):
                for c in range(c0, c1):
                    if mask[r * cols + c]:
                        v = clamp(norm[r * cols + c], 0.0, 1.0)
                        b = as_int(rround(v * (float(num_bins) - 1.0)), -(1 << 63), (1 << 63) - 1)
                        if b < 0:
                            b = 0
                        if b >= num_bins:
                            b = num_bins - 1
                        hist[b] += 1
            avg = float(tile_rows * tile_cols) / float(num_bins)
            clip_threshold = fmax(clip_limit * avg, 1.0)
            excess = 0.0
            for i in range(num_bins):
                if float(hist[i]) > clip_threshold:
                    excess += float(hist[i]) - clip_threshold
                    hist[i] = as_int(clip_threshold, 0, 0xFFFFFFFF)
            add_per_bin = math.floor(excess / float(num_bins))
            remainder = as_int(rround(excess - add_per_bin * float(num_bins)), 0, (1 << 64) - 1)
            for i in range(num_bins):
                hist[i] = as_int(float(hist[i]) + add_per_bin, 0, 0xFFFFFFFF)
            b = 0
            while remainder > 0:
                hist[b] += 1
                b = (b + 1) % num_bins
                remainder -= 1
            total = 0.0
            for x in hist:
                total += float(x)
            total = fmax(total, 1.0)
            acc = 0.0
            cdf = [0.0] * num_bins
            for i in range(num_bins):
                acc += float(hist[i])
                cdf[i] = clamp(acc / total, 0.0, 1.0)
            cdfs[ty * tiles_x + tx] = cdf

    def sample_cdf(r, c, val):
        rf = float(r) / float(tile_h) - 0.5
        cf = float(c) / float(tile_w) - 0.5
        ty = as_int(fmax(math.floor(rf), 0.0), -(1 << 63), (1 << 63) - 1)
        tx = as_int(fmax(math.floor(cf), 0.0), -(1 << 63), (1 << 63) - 1)
        dy = rf - float(ty)
        dx = cf - float(tx)
        ty0 = min(max(ty, 0), tiles_y - 1)
        tx0 = min(max(tx, 0), tiles_x - 1)
        ty1 = min(max(ty + 1, 0), tiles_y - 1)
        tx1 = min(max(tx + 1, 0), tiles_x - 1)
        bin_pos = as_int(rround(clamp(val, 0.0, 1.0) * (float(num_bins) - 1.0)), 0, (1 << 64) - 1)
        cdf00 = cdfs[ty0 * tiles_x + tx0][bin_pos]
        cdf01 = cdfs[ty0 * tiles_x + tx1][bin_pos]
        cdf10 = cdfs[ty1 * tiles_x + tx0][bin_pos]
        cdf11 = cdfs[ty1 * tiles_x + tx1][bin_pos]
        top = cdf00 * (1.0 - dx) + cdf01 * dx
        bottom = cdf10 * (1.0 - dx) + cdf11 * dx
        return top * (1.0 - dy) + bottom * dy

    out = [0.0] * (rows * cols)
    for r in range(rows):
        for c in range(cols):
            if mask[r * cols + c]:
                out[r * cols + c] = sample_cdf(r, c, norm[r * cols + c])
    return out


# ------------------------------------------------------------------------------------------------ autoscale.rs:348-364
def scale_u16_to_u8(data):
    if not data:
        return []
    mn = f32(float(min(data)))
    mx = f32(float(max(data)))
    scale = f32(255.0 / f32(mx - mn)) if mx > mn else 1.0
    out = []
    for x in data:
        val = rround(f32(f32(f32(float(x)) - mn) * scale))
        out.append(as_int(clamp(val, 0.0, 255.0), 0, 255))
    return out


def _max_val(bit_depth):
    return 255.0 if bit_depth == U8 else 65535.0


def _map(db, mask, low_clip, high_clip, rng, gamma, max_val):
    out = []
    for v, ok in zip(db, mask):
        if ok:
            clipped = fmin(fmax(v, low_clip), high_clip)
            normalized = _libm.pow((clipped - low_clip) / rng, gamma)
            out.append(as_int(clamp(normalized * max_val, 0.0, max_val), 0, 65535))
        else:
            out.append(0)
    return out


# ------------------------------------------------------------------------------------------------ autoscale.rs:368-448
def autoscale_db_image(db, mask, bit_depth):
    st = compute_histogram_stats(db, mask)
    if st.valid_count == 0:
        return [0] * len(db), st, None
    min_db, max_db = st.min_db, st.max_db
    max_val = _max_val(bit_depth)
    dynamic_range = max_db - min_db
    iqr = st.p75 - st.p25
    if dynamic_range < 15.0:
        rng_ = fmax(20.0, dynamic_range * 0.8)
        low, high, gamma = st.median_db - rng_ / 2.0, st.median_db + rng_ / 2.0, 1.1
    elif iqr < 5.0:
        outlier_factor = 2.5
        low, high, gamma = st.p25 - outlier_factor * iqr, st.p75 + outlier_factor * iqr, 1.0
    elif dynamic_range > 40.0:
        low = fmax(st.p02, min_db + 0.02 * dynamic_range)
        high = fmin(st.p98, max_db - 0.02 * dynamic_range)
        gamma = 0.9
    else:
        low, high, gamma = st.p02, st.p98, 1.0
    low = fmax(low, min_db)
    high = fmin(high, max_db)
    rng = fmax(high - low, 1.0)
    return _map(db, mask, low, high, rng, gamma, max_val), st, (low, high, gamma)


def approx_eq(a, b):
    return abs(a - b) < 1e-9


# ------------------------------------------------------------------------------------------------ autoscale.rs:452-659
def autoscale_db_image_advanced(db, mask, rows, cols, bit_depth, strategy):
    max_val = _max_val(bit_depth)
    st = compute_histogram_stats(db, mask)
    if st.valid_count == 0:
        return [0] * len(db), st, None
    min_db, max_db = st.min_db, st.max_db
    iqr = st.p75 - st.p25
    if strategy == ROBUST:
        outlier_threshold = 2.5 * iqr
        low = fmax(fmax(st.p25 - outlier_threshold, st.p01), min_db)
        high = fmin(fmin(st.p75 + outlier_threshold, st.p99), max_db)
        gamma = 1.0
    elif strategy == ADAPTIVE:
        skew_factor = (st.mean_db - st.median_db) / fmax(abs(st.std_db), 1.0)
        tail_heaviness = (st.p99 - st.p95) / fmax(st.p95 - st.p75, 1.0)
        if abs(skew_factor) > 0.5:
            low_pct, high_pct, gamma = (0.02, 0.98, 0.9) if skew_factor > 0.0 else (0.05, 0.95, 1.1)
        elif tail_heaviness > 2.0:
            low_pct, high_pct, gamma = 0.10, 0.90, 0.8
        else:
            low_pct, high_pct, gamma = 0.05, 0.95, 1.0
        if approx_eq(low_pct, 0.10): low = st.p10
        elif approx_eq(low_pct, 0.02): low = st.p02
        elif approx_eq(low_pct, 0.05): low = st.p05
        elif approx_eq(low_pct, 0.25): low = st.p25
        elif approx_eq(low_pct, 0.75): low = st.p75
        elif approx_eq(low_pct, 0.95): low = st.p95
        elif approx_eq(low_pct, 0.99): low = st.p99
        else: low = st.p05
        if approx_eq(high_pct, 0.90): high = st.p90
        elif approx_eq(high_pct, 0.98): high = st.p98
        elif approx_eq(high_pct, 0.95): high = st.p95
        elif approx_eq(high_pct, 0.75): high = st.p75
        elif approx_eq(high_pct, 0.99): high = st.p99
        else: high = st.p95
    elif strategy in (EQUALIZED, CLAHE):
        low, high, gamma = st.p01, st.p99, 1.0
    elif strategy == TAMED:
        low, high, gamma = st.p25, st.p99, 1.0
    else:  # Standard | Default
        low, high, gamma = st.p05, st.p95, 1.0
    rng = fmax(high - low, 1.0)
    if strategy == CLAHE:
        norm = []
        for v, ok in zip(db, mask):
            if ok:
                clipped = fmin(fmax(v, low), high)
                norm.append((clipped - low) / rng)
            else:
                norm.append(0.0)
        eq = clahe_equalize_normalized(norm, mask, rows, cols, 8, 8, 2.0, 256)
        out = []
        for n, ok in zip(eq, mask):
            out.append(as_int(clamp(n, 0.0, 1.0) * max_val, 0, 65535) if ok else 0)
        return out, st, (low, high, gamma)
    # (use_local_enhancement is false in every arm: autoscale.rs:613-643 never runs)
    return _map(db, mask, low, high, rng, gamma, max_val), st, (low, high, gamma)


# ------------------------------------------------------------------------------------------------ pipeline.rs:42-67 + autoscale.rs:662-704
def process_scalar_data_pipeline(src, rows, cols, bit_depth, strategy):
    """-> (raster as a flat list of ints, Stats, (low, high, gamma) or None)"""
    db, mask = process_scalar_data_inplace(src)
    if strategy == STANDARD:
        v, st, win = autoscale_db_image(db, mask, bit_depth)
    else:
        v, st, win = autoscale_db_image_advanced(db, mask, rows, cols, bit_depth, strategy)
    if bit_depth == U8:
        v = scale_u16_to_u8(v)
    return v, st, win


# ------------------------------------------------------------------------------------------------ autoscale.rs:710-742
def autoscale_db_image_tamed_synrgb_u8(db, mask, is_copol):
    st = compute_histogram_stats(db, mask)
    if st.valid_count == 0:
        return [0] * len(db)
    low, high = (fmin(st.p02, st.p05), st.p99) if is_copol else (st.p05, st.p99)
    rng = fmax(high - low, 1.0)
    out = []
    for v, ok in zip(db, mask):
        if ok:
            clipped = fmin(fmax(v, low), high)
            normalized = (clipped - low) / rng
            out.append(as_int(clamp(normalized * 255.0, 0.0, 255.0), 0, 255))
        else:
            out.append(0)
    return out


# ------------------------------------------------------------------------------------------------ ops.rs:4-44
def polop(op, a, b):
    """a, b: flat lists of f32 values -> flat list of f32 values (as Python floats)"""
    eps = f32(1e-10)
    out = []
    for x, y in zip(a, b):
        if op == SUM:
            out.append(f32(x + y))
        elif op == DIFF:
            out.append(f32(x - y))
        elif op in (RATIO, LOGRATIO):
            out.append(f32(x / y) if abs(y) > eps else 0.0)
        else:
            d = f32(x + y)
            out.append(f32(f32(x - y) / d) if abs(d) > eps else 0.0)
    return out


# ------------------------------------------------------------------------------------------------ synthetic_rgb.rs:10-67
def create_synthetic_rgb(b1, b2):
    GAMMA_R, GAMMA_G, GAMMA_B, S255, BLUE = f32(0.7), f32(0.9), f32(0.1), 255.0, f32(0.24)
    lut_r, lut_g = [0] * 256, [0] * 256
    for v in range(256):
        vf = f32(float(v) / S255)
        lut_r[v] = as_int(clamp(rround(f32(powf(vf, GAMMA_R) * S255)), 0.0, 255.0), 0, 255)
        lut_g[v] = as_int(clamp(rround(f32(powf(vf, GAMMA_G) * S255)), 0.0, 255.0), 0, 255)
    lut_b = [0] * 65536
    for x1 in range(256):
        for x2 in range(256):
            if x2 == 0:
                blue = 0
            else:
                r, g = float(lut_r[x1]), float(lut_g[x2])
                if g == 0.0:
                    ratio = math.inf if r > 0.0 else math.nan  # r / 0.0 in IEEE arithmetic
                else:
                    ratio = f32(r / g)
                t = f32(f32(powf(ratio, GAMMA_B) * S255) * BLUE)
                blue = as_int(rround(clamp(t, 0.0, 255.0)), 0, 255)
            lut_b[(x1 << 8) | x2] = blue
    out = []
    for v1, v2 in zip(b1, b2):
        out += [lut_r[v1], lut_g[v2], lut_b[(v1 << 8) | v2]]
    return out, (lut_r, lut_g, lut_b)


# ------------------------------------------------------------------------------------------------ synthetic_rgb.rs:88-178
def create_synthetic_rgb_suppressed(b1, b2):
    U32MAX = 0xFFFFFFFF
    histogram = [0] * 256
    for v in b1:
        histogram[v] = min(histogram[v] + 1, U32MAX)
    for v in b2:
        histogram[v] = min(histogram[v] + 1, U32MAX)
    total_count = (len(b1) + len(b2)) & U32MAX           # `as u32` of a usize: wraps
    target_count = as_int(rround(float(total_count) * 0.05), 0, U32MAX)
    cumulative = 0
    floor_value = 0
    for i in range(256):
        cumulative = min(cumulative + histogram[i], U32MAX)
        if cumulative >= target_count:
            floor_value = i
            break
    floor_with_cushion = min(floor_value + 3, 40)
    S255, GR, GG = 255.0, f32(1.15), f32(1.10)
    floor = float(floor_with_cushion)
    denom = fmax(f32(255.0 - floor), 1.0)
    lut_r, lut_g = [0] * 256, [0] * 256
    for v in range(256):
        if v <= floor_with_cushion:
            continue
        shifted = f32(f32(float(v) - floor) / denom)
        lut_r[v] = as_int(clamp(rround(f32(powf(shifted, GR) * S255)), 0.0, 255.0), 0, 255)
        lut_g[v] = as_int(clamp(rround(f32(powf(shifted, GG) * S255)), 0.0, 255.0), 0, 255)
    GB, BLUE, EPS = f32(0.1), f32(0.18), 8.0
    lut_b = [0] * 65536
    for x1 in range(256):
        for x2 in range(256):
            r, g = float(lut_r[x1]), float(lut_g[x2])
            ratio = f32(f32(r + EPS) / f32(g + EPS))
            t = f32(f32(powf(ratio, GB) * S255) * BLUE)
            lut_b[(x1 << 8) | x2] = as_int(rround(clamp(t, 0.0, 255.0)), 0, 255)
    out = []
    for v1, v2 in zip(b1, b2):
        if v1 <= floor_with_cushion and v2 <= floor_with_cushion:
            out += [0, 0, 0]
        else:
            out += [lut_r[v1], lut_g[v2], lut_b[(v1 << 8) | v2]]
    return out, (lut_r, lut_g, lut_b, floor_with_cushion)


# ------------------------------------------------------------------------------------------------ synthetic_rgb.rs:72-79, 182-197
def create_synthetic_rgb_by_mode_and_strategy(mode, strategy, b1, b2):
    if strategy in (TAMED, CLAHE):
        return create_synthetic_rgb_suppressed(b1, b2)[0]
    return create_synthetic_rgb(b1, b2)[0]  # the mode is ignored (all four map to the default composition)


# ------------------------------------------------------------------------------------------------ save.rs:317-367 at native resolution
def dualpol_synrgb(band1, band2, rows, cols, strategy, mode=0):
    """pipeline (U8) per band; Tamed replaces each band's u8 by autoscale_db_image_tamed_synrgb_u8 (save.rs:324-351); then synRGB."""
    u8 = []
    for k, src in enumerate((band1, band2)):
        v, _, _ = process_scalar_data_pipeline(src, rows, cols, U8, strategy)
        if strategy == TAMED:
            db, mask = process_scalar_data_inplace(src)
            v = autoscale_db_image_tamed_synrgb_u8(db, mask, k == 0)
        u8.append(v)
    return create_synthetic_rgb_by_mode_and_strategy(mode, strategy, u8[0], u8[1]), u8[0], u8[1]

"""numpy stand-ins for the HIP kernels, composed with the product's host half (via the C ABI).

TEST INFRASTRUCTURE: lets the CPU-only suite check the host half of the path (statistics from
histograms, tables, CDFs, rescale, compose tables) against the oracle without a GPU, and gives
the multi-rank (gloo) test a per-stripe "kernel".  The product never imports this.
"""
from __future__ import annotations

import numpy as np

import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd


def dn_hist(dn: np.ndarray) -> np.ndarray:
    return np.bincount(dn.ravel(), minlength=65536).astype(np.uint64)


def tile_bin_hists(dn: np.ndarray, binlut: np.ndarray, rows_total: int, cols: int, row0: int = 0) -> np.ndarray:
    """Per-tile 256-bin histograms of the local stripe (rows [row0, row0+dn.shape[0]) of the scene)."""
    th, tw = -(-rows_total // 8), -(-cols // 8)
    out = np.zeros((64, 256), np.uint64)
    bins = binlut[dn]
    valid = dn > 0
    for ty in range(8):
        r0, r1 = max(ty * th, row0) - row0, min((ty + 1) * th, rows_total, row0 + dn.shape[0]) - row0
        if r1 <= r0:
            continue
        for tx in range(8):
            c0, c1 = tx * tw, min((tx + 1) * tw, cols)
            if c1 <= c0:
                continue
            sl = (slice(r0, r1), slice(c0, c1))
            out[ty * 8 + tx] = np.bincount(bins[sl][valid[sl]].ravel(), minlength=256)
    return out


def clahe_apply(dn: np.ndarray, binlut: np.ndarray, cdfs: np.ndarray, rows_total: int, cols: int, max_val: float,
                row0: int = 0) -> np.ndarray:
    th, tw = -(-rows_total // 8), -(-cols // 8)
    r = (np.arange(dn.shape[0]) + row0)[:, None]
    c = np.arange(cols)[None, :]
    rf = r / float(th) - 0.5
    cf = c / float(tw) - 0.5
    ty = np.maximum(np.floor(rf), 0).astype(np.int64)
    tx = np.maximum(np.floor(cf), 0).astype(np.int64)
    dy, dx = rf - ty, cf - tx
    ty0, ty1 = np.clip(ty, 0, 7), np.clip(ty + 1, 0, 7)
    tx0, tx1 = np.clip(tx, 0, 7), np.clip(tx + 1, 0, 7)
    bins = binlut[dn]
    c00, c01 = cdfs[ty0 * 8 + tx0, bins], cdfs[ty0 * 8 + tx1, bins]
    c10, c11 = cdfs[ty1 * 8 + tx0, bins], cdfs[ty1 * 8 + tx1, bins]
    top = c00 * (1.0 - dx) + c01 * dx
    bot = c10 * (1.0 - dx) + c11 * dx
    o = top * (1.0 - dy) + bot * dy
    lv = (np.clip(o, 0.0, 1.0) * max_val).astype(np.uint16)
    return np.where(dn > 0, lv, 0).astype(np.uint16)


def band_levels(dn: np.ndarray, strategy: St, u8: bool, tamed: int = 0, rows_total=None, row0=0, reduce=None):
    """Levels (pre-rescale) of one band + stats, the way the device path computes them.
    `reduce(array)->array` merges integer histograms across ranks (identity for one rank)."""
    reduce = reduce or (lambda a: a)
    rows_total = rows_total or dn.shape[0]
    cols = dn.shape[1]
    st = S.host_stats_from_dn_hist(reduce(dn_hist(dn)))
    S.host_window(st, strategy, tamed)
    if st.valid_count == 0:
        return np.zeros(dn.shape, np.uint16), st
    if strategy != St.Clahe or tamed:
        lut = S.host_level_lut_u16(st, Bd.U8 if u8 else Bd.U16, tamed)
        return lut[dn], st
    binlut = S.host_clahe_bin_lut_u16(st)
    th = reduce(tile_bin_hists(dn, binlut, rows_total, cols, row0))
    cdfs = S.host_clahe_cdfs(th, rows_total, cols)
    return clahe_apply(dn, binlut, cdfs, rows_total, cols, 255.0 if u8 else 65535.0, row0), st


def pipeline(dn: np.ndarray, bit_depth: Bd, strategy: St):
    """process_scalar_data_pipeline for a u16 band -> (raster, stats)."""
    lv, st = band_levels(dn, strategy, bit_depth == Bd.U8)
    if bit_depth == Bd.U16:
        return lv, st
    resc = S.host_u8_rescale_lut(int(lv.min()), int(lv.max())) if lv.size else np.arange(256, dtype=np.uint8)
    return resc[lv], st


def dualpol_synrgb(b1: np.ndarray, b2: np.ndarray, strategy: St, rows_total=None, row0=0, reduce=None):
    """save.rs:317-367 at native resolution -> (rgb, u8_1, u8_2) for the local stripe."""
    reduce = reduce or (lambda a: a)
    rows_total = rows_total or b1.shape[0]
    u8 = []
    hists = []
    for k, dn in enumerate((b1, b2)):
        tamed = (1 if k == 0 else 2) if strategy == St.Tamed else 0
        lv, _ = band_levels(dn, strategy, True, tamed, rows_total, row0, reduce)
        lh = reduce(np.bincount(lv.ravel(), minlength=256).astype(np.uint64))
        if tamed:
            resc = np.arange(256, dtype=np.uint8)
        else:
            nz = np.nonzero(lh)[0]
            resc = S.host_u8_rescale_lut(int(nz[0]), int(nz[-1])) if nz.size else np.arange(256, dtype=np.uint8)
        u8.append(resc[lv])
        fh = np.zeros(256, np.uint64)
        np.add.at(fh, resc, lh)
        hists.append(fh)
    lut_r, lut_g, lut_b, _ = S.host_synrgb_luts(strategy, hists[0] + hists[1], rows_total * b1.shape[1])
    rgb = np.stack([lut_r[u8[0]], lut_g[u8[1]], lut_b[u8[0], u8[1]]], axis=-1)
    if strategy in (St.Tamed, St.Clahe):
        fl = S.host_synrgb_luts(strategy, hists[0] + hists[1], rows_total * b1.shape[1])[3]
        rgb[(u8[0] <= fl) & (u8[1] <= fl)] = 0
    return rgb, u8[0], u8[1]


# ----------------------------------------------------------------------------- f32 flavour
def _steps(thr: np.ndarray, v: np.ndarray) -> np.ndarray:
    """#{k >= 1 : v >= thr[k]} -- what the device's branch-free binary search returns."""
    return np.searchsorted(thr[1:], v, side="right")


def f32_levels(x: np.ndarray, strategy: St, u8: bool, tamed: int = 0, rows_total=None, row0=0, reduce=None, gather=None):
    """Levels of an f32 band (or of a row stripe of it) through the product's threshold tables (numpy search = kernel).
    Stripes: `gather(partial) -> [partials of all ranks]` where the GPU path gathers the 32-byte partials, `reduce(hist)`
    where it all-reduces a u64 histogram (sarpro_hip_stripe_f32_phase1..4)."""
    import oracle  # only for the dB moments (mean/std), which the device computes with its own log10
    reduce = reduce or (lambda a: a)
    gather = gather or (lambda p: [p])
    x = np.ascontiguousarray(x, np.float32)
    rows_total = rows_total or x.shape[0]
    cols = x.shape[1]
    tv = np.float32(S.host_f32_valid_threshold())
    valid = x >= tv  # NaN -> False
    db_all, _ = oracle.db_mask(x) if x.size else (np.zeros(x.shape), None)
    dbv = db_all[valid]
    local = S.F32Partial(int(valid.sum()), float(dbv.sum()), float((dbv * dbv).sum()),
                         float(x[valid].min()) if valid.any() else float("inf"), float(x[valid].max()) if valid.any() else float("-inf"))
    g = S.host_f32_merge_partials(gather(local))
    n = int(g.count)
    if n == 0:
        return np.zeros(x.shape, np.uint16), None
    db_min = 10.0 * np.log10(np.float64(np.float32(g.min_v)))
    db_max = 10.0 * np.log10(np.float64(np.float32(g.max_v)))
    if rows_total == x.shape[0]:
        assert db_min == dbv.min() and db_max == dbv.max()
        mean, std = float(dbv.mean()), float(dbv.std())
    else:
        mean = g.sum_db / n
        std = float(np.sqrt(max(g.sumsq_db / n - mean * mean, 0.0))) if n > 1 else 0.0
    hist = np.zeros(4096, np.uint64)
    if not abs(db_max - db_min) < np.finfo(np.float64).eps:
        thr = S.host_f32_bin4096_thresholds(float(db_min), float(db_max))
        hist = np.bincount(_steps(thr, x[valid]), minlength=4096).astype(np.uint64)
    st = S.host_stats_from_bins4096(n, float(db_min), float(db_max), mean, std, reduce(hist))
    S.host_window(st, strategy, tamed)
    if strategy == St.Clahe and not tamed:
        bins = _steps(S.host_f32_clahe_bin_thresholds(st), x).astype(np.int64)
        th_, tw_ = -(-rows_total // 8), -(-cols // 8)
        tile = np.zeros((64, 256), np.uint64)
        for ty in range(8):
            r0, r1 = max(ty * th_, row0) - row0, min((ty + 1) * th_, rows_total, row0 + x.shape[0]) - row0
            if r1 <= r0:
                continue
            for tx in range(8):
                sl = (slice(r0, r1), slice(tx * tw_, min((tx + 1) * tw_, cols)))
                tile[ty * 8 + tx] = np.bincount(bins[sl][valid[sl]].ravel(), minlength=256)
        cdfs = S.host_clahe_cdfs(reduce(tile), rows_total, cols)
        # reuse the u16 blend helper: feed it "bins" through an identity-like table
        lut = np.arange(65536, dtype=np.int64) % 256
        fake = np.where(valid, bins, 0).astype(np.uint16)
        lv = clahe_apply(np.where(valid, fake + 256 * 1, 0).astype(np.uint16), lut.astype(np.uint8), cdfs, rows_total, cols,
                         255.0 if u8 else 65535.0, row0)
        return np.where(valid, lv, 0).astype(np.uint16), st
    thr = S.host_f32_level_thresholds(st, Bd.U8 if (u8 or tamed) else Bd.U16)
    lv = _steps(thr, x)
    return np.where(valid, lv, 0).astype(np.uint16), st


def f32_pipeline(x: np.ndarray, bit_depth: Bd, strategy: St, rows_total=None, row0=0, reduce=None, gather=None):
    lv, st = f32_levels(x, strategy, bit_depth == Bd.U8, 0, rows_total, row0, reduce, gather)
    if bit_depth == Bd.U16 or st is None:
        return lv, st
    reduce = reduce or (lambda a: a)
    lh = reduce(np.bincount(lv.ravel(), minlength=256).astype(np.uint64))  # the level histogram the GPU path all-reduces
    nz = np.nonzero(lh)[0]
    resc = S.host_u8_rescale_lut(int(nz[0]), int(nz[-1]))
    return resc[lv], st

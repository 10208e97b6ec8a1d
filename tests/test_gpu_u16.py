"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle, u16 (integer-DN) flavour.

Bar: bit-exact for every u8/u16 raster.  Sizes are small enough that the oracle runs in
seconds.  Mirrors how the reference's own tests would read: call the reference-named
function, compare the raster.
"""
import numpy as np
import pytest

import oracle
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SyntheticRgbMode as Mode
from sarpro_amd import SarproHipError, synth
from sarpro_amd import _lib
import sarpro_amd as S

pytestmark = pytest.mark.gpu

SHAPES = [(257, 300), (64, 64), (100, 333), (512, 640), (7, 8), (43, 1000)]


def scene(rows, cols, band, **kw):
    return synth.scene_u16(rows, cols, band, **kw)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("strategy", list(St))
@pytest.mark.parametrize("bit_depth", list(Bd))
def test_pipeline_u16_matches_oracle(ctx, shape, strategy, bit_depth):
    rows, cols = shape
    for band in (0, 1):
        dn = scene(rows, cols, band)
        u8, u16, st = ctx.process_scalar_data_pipeline(dn, bit_depth, strategy, want_stats=True)
        rc, ref, so = oracle.pipeline(dn.astype(np.float32), int(bit_depth), int(strategy), want_stats=True)
        assert rc == 0
        got = u8 if bit_depth == Bd.U8 else u16
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref), f"{(got != ref).sum()} px differ"
        # statistics: exact except Welford mean/std (summed per DN instead)
        for k in ("valid_count", "min_db", "max_db", "median_db", "p01", "p02", "p05", "p10", "p25", "p75",
                  "p90", "p95", "p98", "p99", "low_clip", "high_clip", "gamma"):
            assert getattr(st, k) == getattr(so, k), k
        assert abs(st.mean_db - so.mean_db) <= 1e-9 * max(1.0, abs(so.mean_db))
        assert abs(st.std_db - so.std_db) <= 1e-9 * max(1.0, abs(so.std_db))


@pytest.mark.parametrize("shape", [(257, 300), (96, 1031), (512, 640)])
@pytest.mark.parametrize("strategy", list(St))
def test_dualpol_synrgb_u16_matches_oracle(ctx, shape, strategy):
    rows, cols = shape
    b1, b2 = scene(rows, cols, 0), scene(rows, cols, 1)
    rgb, u1, u2 = ctx.dualpol_synrgb(b1, b2, strategy, Mode.Default, want_u8=True)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1.astype(np.float32), b2.astype(np.float32), int(strategy))
    assert rc == 0
    assert np.array_equal(u1, r1), f"band1 {(u1 != r1).sum()} px differ"
    assert np.array_equal(u2, r2), f"band2 {(u2 != r2).sum()} px differ"
    assert np.array_equal(rgb, rrgb), f"rgb {(rgb != rrgb).any(axis=2).sum()} px differ"
    # without the per-band outputs the composition must be the same
    rgb2 = ctx.dualpol_synrgb(b1, b2, strategy, Mode.SarUrban)  # mode is ignored by the reference (synthetic_rgb.rs:72-79)
    assert np.array_equal(rgb2, rrgb)


@pytest.mark.parametrize("is_copol", [True, False])
def test_tamed_synrgb_u8(ctx, is_copol):
    dn = scene(200, 264, 0 if is_copol else 1)
    got = ctx.autoscale_db_image_tamed_synrgb_u8(dn, is_copol)
    ref = oracle.tamed_synrgb_u8(dn.astype(np.float32), is_copol)
    assert np.array_equal(got, ref)


def test_flat_scene_takes_iqr_arm(ctx):
    # single-class Rayleigh: IQR ~3.4 dB < 5 -> Standard's `iqr < 5` arm (autoscale.rs:409-413)
    q = synth.q_tables(flat=True)
    dn = synth.scene_u16(300, 300, 0, q=q)
    u8, _, st = ctx.process_scalar_data_pipeline(dn, Bd.U8, St.Standard, want_stats=True)
    rc, ref, so = oracle.pipeline(dn.astype(np.float32), 0, 0, want_stats=True)
    assert (st.p75 - st.p25) < 5.0 and (st.max_db - st.min_db) >= 15.0 and st.gamma == 1.0
    assert np.array_equal(u8, ref)


@pytest.mark.parametrize("strategy", list(St))
@pytest.mark.parametrize("bit_depth", list(Bd))
def test_degenerate_inputs(ctx, strategy, bit_depth):
    for dn in (np.zeros((50, 70), np.uint16),              # no valid pixel -> zero raster (autoscale.rs:376-378)
               np.full((50, 70), 1234, np.uint16),         # constant -> degenerate stats branch (:81-100)
               np.where(np.arange(3500).reshape(50, 70) % 2 == 0, 7, 65535).astype(np.uint16)):
        u8, u16 = ctx.process_scalar_data_pipeline(dn, bit_depth, strategy)
        rc, ref = oracle.pipeline(dn.astype(np.float32), int(bit_depth), int(strategy))
        assert rc == 0
        assert np.array_equal(u8 if bit_depth == Bd.U8 else u16, ref)


def test_low_contrast_arm(ctx):
    # dynamic range < 15 dB -> Standard's median-window arm with gamma 1.1 (autoscale.rs:404-408)
    rng = np.random.default_rng(5)
    dn = rng.integers(1000, 4000, size=(120, 168)).astype(np.uint16)
    u16 = ctx.process_scalar_data_pipeline(dn, Bd.U16, St.Standard)[1]
    rc, ref, so = oracle.pipeline(dn.astype(np.float32), 1, 0, want_stats=True)
    assert so.gamma == 1.1
    assert np.array_equal(u16, ref)


def test_uniform_dn_exercises_global_atomics_and_big_window(ctx):
    # DN uniform over the whole u16 range: the histogram's bright tail path (DN >= LDS window)
    # and a table window too large for LDS (gathered from global memory instead)
    rng = np.random.default_rng(11)
    dn = rng.integers(0, 65536, size=(256, 512)).astype(np.uint16)
    for strategy in (St.Clahe, St.Robust, St.Adaptive):
        for bd in Bd:
            u8, u16 = ctx.process_scalar_data_pipeline(dn, bd, strategy)
            rc, ref = oracle.pipeline(dn.astype(np.float32), int(bd), int(strategy))
            assert np.array_equal(u8 if bd == Bd.U8 else u16, ref)


def test_clahe_unsupported_shapes(ctx):
    # shapes where the reference's tile arithmetic underflows (autoscale.rs:250,254): it panics,
    # the oracle and the library both refuse
    for rows, cols in ((9, 64), (64, 13), (3, 3), (34, 100)):
        assert not oracle.clahe_shape_ok(rows, cols)
        dn = scene(rows, cols, 0)
        with pytest.raises(SarproHipError) as ei:
            ctx.process_scalar_data_pipeline(dn, Bd.U8, St.Clahe)
        assert ei.value.code == _lib.ERR_UNSUPPORTED_SHAPE
        rc, _ = oracle.pipeline(dn.astype(np.float32), 0, int(St.Clahe))
        assert rc == oracle.ERR_UNSUPPORTED_SHAPE
        # the other strategies do not tile and must still work
        u8, _ = ctx.process_scalar_data_pipeline(dn, Bd.U8, St.Robust)
        assert np.array_equal(u8, oracle.pipeline(dn.astype(np.float32), 0, int(St.Robust))[1])


def test_empty_raster(ctx):
    dn = np.zeros((0, 0), np.uint16)
    u8, _ = ctx.process_scalar_data_pipeline(dn, Bd.U8, St.Clahe)
    assert u8.shape == (0, 0)


@pytest.mark.parametrize("op", range(5))
def test_polops_match_oracle(ctx, op):
    rng = np.random.default_rng(op)
    a = (rng.standard_normal((123, 457)) * 300).astype(np.float32)
    b = (rng.standard_normal((123, 457)) * 300).astype(np.float32)
    # zero handling / tiny denominators / specials
    b.ravel()[:7] = [0.0, 1e-10, -1e-10, 1.0000001e-10, np.inf, np.nan, -0.0]
    a.ravel()[7:10] = [np.inf, -np.inf, np.nan]
    b.ravel()[10:13] = -a.ravel()[10:13]  # a + b == 0 for the normalized difference
    fn = [ctx.sum_arrays, ctx.difference_arrays, ctx.ratio_arrays, ctx.normalized_diff_arrays, ctx.log_ratio_arrays][op]
    got = fn(a, b)
    ref = oracle.polop(op, a, b)
    # bit-exact wherever the result is a number; NaN sign/payload is not defined by IEEE-754 for
    # invalid operations (x86 yields 0xFFC00000, gfx950 0x7FC00000), so NaNs match by position
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])


@pytest.mark.parametrize("strategy", [St.Robust, St.Clahe, St.Tamed])
@pytest.mark.parametrize("n", [1, 4095, 4096, 100003])
def test_synrgb_matches_oracle(ctx, strategy, n):
    rng = np.random.default_rng(n)
    b1 = rng.integers(0, 256, n).astype(np.uint8)
    b2 = rng.integers(0, 256, n).astype(np.uint8)
    if n > 100:
        b1[: n // 3] = 0  # a dark wedge moves the suppressed floor (synthetic_rgb.rs:100-113)
        b2[: n // 3] = 0
    got = ctx.create_synthetic_rgb_by_mode_and_strategy(Mode.Default, strategy, b1, b2)
    ref = oracle.synrgb(0, int(strategy), b1, b2)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("shape", [(257, 300), (512, 640), (43, 1000)])
def test_device_chain_equals_host_orchestrated_path(ctx, shape, monkeypatch):
    """CLAHE u8 runs as a device-resident chain (statistics, bins, CDFs, tables computed by kernels);
    SARPRO_HIP_NO_CHAIN=1 forces the host-orchestrated phases.  Both must give the oracle's raster."""
    rows, cols = shape
    b1, b2 = scene(rows, cols, 0), scene(rows, cols, 1)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1.astype(np.float32), b2.astype(np.float32), int(St.Clahe))
    for no_chain in ("0", "1"):
        monkeypatch.setenv("SARPRO_HIP_NO_CHAIN", no_chain)
        rgb, u1, u2, st = ctx.dualpol_synrgb(b1, b2, St.Clahe, want_u8=True, want_stats=True)
        assert np.array_equal(rgb, rrgb) and np.array_equal(u1, r1) and np.array_equal(u2, r2)
        u8, _, s1 = ctx.process_scalar_data_pipeline(b1, Bd.U8, St.Clahe, want_stats=True)
        assert np.array_equal(u8, r1)
        _, u16, s16 = ctx.process_scalar_data_pipeline(b1, Bd.U16, St.Clahe, want_stats=True)  # u16 output: the chain up to the exact f64 blend
        rc16, ref16 = oracle.pipeline(b1.astype(np.float32), 1, int(St.Clahe))
        assert rc16 == 0 and np.array_equal(u16, ref16) and s16.p99 == s1.p99
        rc1, _, so = oracle.pipeline(b1.astype(np.float32), 0, int(St.Clahe), want_stats=True)
        for k in ("valid_count", "min_db", "max_db", "median_db", "p01", "p10", "p25", "p75", "p99", "low_clip", "high_clip"):
            assert getattr(s1, k) == getattr(so, k) == getattr(st[0], k), (k, no_chain)


def test_device_chain_non_identity_rescale(ctx):
    # a band whose CLAHE levels do not span 0..255 (no invalid pixel, narrow data): the chain must
    # apply scale_u16_to_u8 (autoscale.rs:348-364) through its device-built map
    rng = np.random.default_rng(2)
    dn = rng.integers(900, 1100, size=(96, 128)).astype(np.uint16)
    dn2 = rng.integers(300, 5000, size=(96, 128)).astype(np.uint16)
    u8, _ = ctx.process_scalar_data_pipeline(dn, Bd.U8, St.Clahe)
    rc, ref = oracle.pipeline(dn.astype(np.float32), 0, int(St.Clahe))
    assert rc == 0 and np.array_equal(u8, ref)
    rgb, u1, u2 = ctx.dualpol_synrgb(dn, dn2, St.Clahe, want_u8=True)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(dn.astype(np.float32), dn2.astype(np.float32), int(St.Clahe))
    assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, rrgb)


@pytest.mark.parametrize("kind", ["two_values", "high_plateau", "natural", "constant"])
def test_partial_level_histogram_and_its_recount_fallback(ctx, kind, monkeypatch):
    """The chain's apply kernel only counts levels < 64 one by one (levels above in bulk at the lane's highest level);
    k_level_hist_guard proves that enough or flags the band for a recount of its level raster.  Rasters whose CLAHE
    levels all lie above 27 take the recount; SARPRO_HIP_FULL_LEVEL_HIST=1 is the full histogram.  All equal the oracle."""
    rng = np.random.default_rng(11)
    rows, cols = 264, 512
    if kind == "two_values":      # two populated bins: levels ~127 and 255 only
        b1 = rng.choice(np.array([120, 4000], np.uint16), size=(rows, cols))
        b2 = rng.choice(np.array([300, 900], np.uint16), size=(rows, cols))
    elif kind == "high_plateau":  # 80 % of the pixels in the lowest bin: its CDF foot is already at ~0.8
        b1 = np.where(rng.random((rows, cols)) < 0.8, 50, rng.integers(51, 6000, (rows, cols))).astype(np.uint16)
        b2 = np.where(rng.random((rows, cols)) < 0.5, 70, rng.integers(71, 3000, (rows, cols))).astype(np.uint16)
    elif kind == "constant":
        b1 = np.full((rows, cols), 777, np.uint16)
        b2 = np.full((rows, cols), 12, np.uint16)
    else:
        b1, b2 = scene(rows, cols, 0), scene(rows, cols, 1)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1.astype(np.float32), b2.astype(np.float32), int(St.Clahe))
    assert rc == 0
    for full in ("", "1"):
        if full:
            monkeypatch.setenv("SARPRO_HIP_FULL_LEVEL_HIST", full)
        else:
            monkeypatch.delenv("SARPRO_HIP_FULL_LEVEL_HIST", raising=False)
        rgb, u1, u2 = ctx.dualpol_synrgb(b1, b2, St.Clahe, want_u8=True)
        assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, rrgb), (kind, full)


@pytest.mark.parametrize("strategy", [St.Standard, St.Robust, St.Adaptive, St.Equalized, St.Tamed, St.Default])
def test_single_band_percentile_chain_equals_host_orchestrated_path(strategy, monkeypatch):
    """Single-band u8 of the percentile strategies runs as a device chain too (statistics, window, level table, rescale
    by kernels, then one table pass); SARPRO_HIP_NO_CHAIN=1 is the host-orchestrated route.  Also the tamed-synRGB entry."""
    rows, cols = 300, 520
    b1 = scene(rows, cols, 0)
    rc, ref = oracle.pipeline(b1.astype(np.float32), 0, int(strategy))
    assert rc == 0
    tam = oracle.tamed_synrgb_u8(b1.astype(np.float32), True)
    with S.Context(0, timing=True) as c:
        for no_chain in ("0", "1"):
            monkeypatch.setenv("SARPRO_HIP_NO_CHAIN", no_chain)
            u8, _, st = c.process_scalar_data_pipeline(b1, Bd.U8, strategy, want_stats=True)
            assert np.array_equal(u8, ref), (strategy, no_chain)
            names = [n for n, _ in c.last_kernel_times()]
            assert ("chain_stats" in names) == (no_chain == "0")
            assert np.array_equal(c.autoscale_db_image_tamed_synrgb_u8(b1, True), tam)
            # u16 output: the chain builds the 65535-level table of every DN on the device (gamma != 1 through the device
            # pow with a certification margin; SARPRO_HIP_FORCE_UNCERTAIN exercises the rerun on the host route)
            rc16, ref16 = oracle.pipeline(b1.astype(np.float32), 1, int(strategy))
            for force in (False, True):
                if force:
                    monkeypatch.setenv("SARPRO_HIP_FORCE_UNCERTAIN", "1")
                else:
                    monkeypatch.delenv("SARPRO_HIP_FORCE_UNCERTAIN", raising=False)
                _, u16, _ = c.process_scalar_data_pipeline(b1, Bd.U16, strategy, want_stats=True)
                assert rc16 == 0 and np.array_equal(u16, ref16), (strategy, no_chain, force)
                names = [n for n, _ in c.last_kernel_times()]
                assert ("chain_stats" in names) == (no_chain == "0")
                assert ("host:phase1_launch" in names) == (no_chain == "1" or force)
            monkeypatch.delenv("SARPRO_HIP_FORCE_UNCERTAIN", raising=False)


def test_tile_histograms_left_clean_by_the_previous_scene():
    """The chains' last reader of the tile histograms zeroes them and the next scene skips its fill: scenes of different
    footprints, strategies and band counts on ONE context, with a failing call and a host-route call in between."""
    import os
    rng = np.random.default_rng(77)
    def band(rows, cols):
        return np.clip(rng.rayleigh(rng.uniform(50, 2000), (rows, cols)), 0, 65535).astype(np.uint16)
    with S.Context(0) as c:
        seq = [(St.Clahe, 300, 500, 2), (St.Robust, 120, 90, 2), (St.Clahe, 90, 130, 2), (St.Clahe, 700, 900, 2), (St.Adaptive, 700, 900, 2),
               (St.Clahe, 64, 64, 1), (St.Tamed, 300, 500, 2), (St.Clahe, 300, 500, 2)]
        for k, (strategy, rows, cols, nb) in enumerate(seq):
            b = [band(rows, cols) for _ in range(2)]
            if k == 3:  # a call that fails after validation, then one through the host-orchestrated phases
                with pytest.raises(S.SarproHipError):
                    c.dualpol_synrgb(band(5, 300), band(5, 300), St.Clahe)
                os.environ["SARPRO_HIP_NO_CHAIN"] = "1"
                try:
                    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(St.Clahe))
                    assert rc == 0 and np.array_equal(c.dualpol_synrgb(b[0], b[1], St.Clahe), rrgb)
                finally:
                    os.environ.pop("SARPRO_HIP_NO_CHAIN")
            if nb == 2:
                rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
                assert rc == 0 and np.array_equal(c.dualpol_synrgb(b[0], b[1], strategy), rrgb), (k, strategy)
            else:
                rc, ref = oracle.pipeline(b[0].astype(np.float32), int(Bd.U8), int(strategy))
                assert rc == 0 and np.array_equal(c.process_scalar_data_pipeline(b[0], Bd.U8, strategy)[0], ref), (k, strategy)


@pytest.mark.parametrize("kind", ["uniform", "bright_plateau", "half_dark"])
def test_piece_histogram_tail_queue_and_its_overflow_route(kind):
    """The piece histogram (whole dual-pol scenes, csrc/piece_kernels.hip) keeps DN < 8192 in LDS bins and queues brighter samples in
    LDS (1024 entries per workgroup and tile visit) for the tile's global histogram; a lane that finds the queue full recounts its
    tail samples in a second loop behind the row loop.  Rasters that are mostly brighter than the LDS bins drive every workgroup
    through that route, across several tiles and pieces; statistics and rasters must be the oracle's."""
    rng = np.random.default_rng(17)
    rows, cols = 1500, 2100
    if kind == "uniform":            # 87 % of the samples beyond the LDS bins, every 65536-bin counter in use
        b = [rng.integers(1, 65536, (rows, cols)).astype(np.uint16) for _ in range(2)]
    elif kind == "bright_plateau":   # every sample in the tail, a few thousand distinct DNs, an invalid stripe
        b = [(20000 + rng.integers(0, 3000, (rows, cols))).astype(np.uint16) for _ in range(2)]
        b[0][:, :97] = 0
    else:                            # rows alternate between dark (LDS bins) and bright (queue): the queue fills and drains per tile
        b = []
        for k in range(2):
            x = rng.integers(1, 4000, (rows, cols)).astype(np.uint16)
            x[::2] = (9000 + rng.integers(0, 50000, (rows // 2 + rows % 2, cols))).astype(np.uint16)
            b.append(x)
    for strategy in (St.Clahe, St.Robust):
        rc, ref, r1, r2 = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
        assert rc == 0
        with S.Context(0, timing=True) as c:
            rgb, u1, u2 = c.dualpol_synrgb(b[0], b[1], strategy, want_u8=True)
            names = [n for n, _ in c.last_kernel_times()]
        assert "dn_hist_u16" in names
        assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, ref), (kind, strategy)

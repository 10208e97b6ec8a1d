"""The zone route of the f32 flavour (f32_path.cpp): percentiles without the 4096-bin sweep -- a row sample places zones
around the percentiles the strategy reads, the min / max pass keeps the samples inside them, a count against the few
thresholds that fall inside a zone gives the exact bin, count below and count in the bin (autoscale.rs:120-140).  The raster
must be the oracle's; the route must actually run where expected and step aside where it cannot answer."""
import numpy as np
import pytest

import f32data
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ZONE_STRATEGIES = [St.Standard, St.Robust, St.Equalized, St.Clahe, St.Tamed, St.Default]


def names(ctx):
    return [n for n, _ in ctx.last_kernel_times()]


@pytest.fixture(autouse=True)
def _no_direct_route(monkeypatch):
    """These rasters are small: without this they would take the small-scene direct route, not the zone route under test."""
    monkeypatch.setenv("SARPRO_HIP_F32_DIRECT", "0")


@pytest.mark.parametrize("strategy", ZONE_STRATEGIES)
@pytest.mark.parametrize("bd", list(Bd))
@pytest.mark.parametrize("scene", ["ratio", "resampled", "nasty"])
def test_zone_route_equals_oracle(strategy, bd, scene, monkeypatch):
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "force")
    x = {"ratio": lambda: f32data.ratio_scene(403, 520), "resampled": lambda: f32data.resampled_scene(300, 421),
         "nasty": lambda: f32data.nasty_scene(257, 389)}[scene]()
    rc, ref = oracle.pipeline(x, int(bd), int(strategy))
    assert rc == 0
    with S.Context(0, timing=True) as c:
        got = c.process_scalar_data_pipeline(x, bd, strategy)
        out = got[0] if bd == Bd.U8 else got[1]
        assert np.array_equal(out, ref)
        if scene == "ratio":  # a well-behaved scene: the route must have answered, not stepped aside
            assert "f32_zone_count" in names(c) and "f32_hist4096" not in names(c), names(c)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Standard])
def test_zone_route_steps_aside(strategy, monkeypatch):
    """No room in the side buffer: the first kept sample overflows, the 4096-bin sweep answers instead."""
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "tiny")
    x = f32data.ratio_scene(403, 520)
    rc, ref = oracle.pipeline(x, int(Bd.U8), int(strategy))
    with S.Context(0, timing=True) as c:
        got = c.process_scalar_data_pipeline(x, Bd.U8, strategy)
        assert np.array_equal(got[0], ref)
        assert "f32_prepass_zones" in names(c) and "f32_hist4096" in names(c)


def test_zone_route_not_taken_when_statistics_are_wanted_or_adaptive(monkeypatch):
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "force")
    x = f32data.ratio_scene(403, 520)
    with S.Context(0, timing=True) as c:
        out8, _, st = c.process_scalar_data_pipeline(x, Bd.U8, St.Clahe, want_stats=True)
        assert "f32_hist4096" in names(c) and "f32_zone_count" not in names(c)
        rc, ref, so = oracle.pipeline(x, int(Bd.U8), int(St.Clahe), want_stats=True)
        assert np.array_equal(out8, ref) and st.p01 == so.p01 and st.p99 == so.p99
        c.process_scalar_data_pipeline(x, Bd.U8, St.Adaptive)
        assert "f32_hist4096" in names(c)


def test_zone_route_heavy_ties_and_degenerate_scenes(monkeypatch):
    """Few distinct values (a zone then holds a large share of the scene), a constant scene, no valid sample."""
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "force")
    rng = np.random.default_rng(3)
    rows, cols = 300, 400
    cases = [rng.choice(np.array([0.5, 1.0, 1.0, 1.0, 2.0, 3.0, 40.0], np.float32), (rows, cols)),
             np.full((rows, cols), 2.5, np.float32), np.zeros((rows, cols), np.float32),
             np.where(rng.random((rows, cols)) < 0.98, 0.0, rng.gamma(2.0, 3.0, (rows, cols))).astype(np.float32)]
    with S.Context(0, timing=True) as c:
        for x in cases:
            for strategy in (St.Clahe, St.Robust, St.Default):
                rc, ref = oracle.pipeline(x, int(Bd.U8), int(strategy))
                assert rc == 0
                assert np.array_equal(c.process_scalar_data_pipeline(x, Bd.U8, strategy)[0], ref)


@pytest.mark.parametrize("op", [Op.LogRatio, Op.NDiff, Op.Sum])
@pytest.mark.parametrize("strategy", [St.Clahe, St.Standard])
def test_zone_route_with_the_pol_op_computed_on_the_fly(op, strategy, monkeypatch):
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "force")
    rows, cols = 411, 536
    a, b = synth.scene_u16(rows, cols, 0), synth.scene_u16(rows, cols, 1)
    r = oracle.polop(int(op), a.astype(np.float32), b.astype(np.float32))
    with S.Context(0, timing=True) as c:
        for bd in Bd:
            rc, ref = oracle.pipeline(r, int(bd), int(strategy))
            got = c.polop_autoscale_band(op, a, b, bd, strategy)
            assert np.array_equal(got[0] if bd == Bd.U8 else got[1], ref), (op, strategy, bd)
        assert "f32_prepass_zones" in names(c)


@pytest.mark.parametrize("side", [3000, 20000])
def test_zone_route_default_on_for_large_rasters_and_equal_to_the_sweep(side, monkeypatch):
    """No switch: the route is the default -- a row SAMPLE places the zones (every 4th row at 3000^2, every 31st at
    BASELINE.json's 20000^2) -- and its raster equals the 4096-bin route's, pixel for pixel."""
    rows, cols, pitch = side, side, (side + 63) // 64 * 64
    q = synth.q_tables()
    with S.Context(0, timing=True) as c:
        d = [torch.zeros((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + 2, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
        for strategy, bd, dt in ((St.Clahe, Bd.U16, torch.int16), (St.Robust, Bd.U8, torch.uint8), (St.Standard, Bd.U16, torch.int16)):
            outs = []
            for env in (None, "0"):
                if env:
                    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", env)
                else:
                    monkeypatch.delenv("SARPRO_HIP_F32_ZONES", raising=False)
                o = torch.zeros((rows, pitch), dtype=dt, device="cuda")
                torch.cuda.synchronize()
                c.dev_polop_autoscale_band(Op.LogRatio, d[0].data_ptr(), d[1].data_ptr(), True, rows, cols, pitch, strategy, bd, o.data_ptr(), pitch,
                                           want_stats=False)
                assert ("f32_zone_count" in names(c)) == (env is None), (strategy, names(c))
                outs.append(o[:, :cols].clone())
            assert torch.equal(outs[0], outs[1]), strategy


@pytest.mark.parametrize("strategy", [St.Tamed, St.Clahe, St.Robust])
def test_zone_route_inside_the_dual_pol_f32_product(strategy, monkeypatch):
    """save.rs:317-367 for f32 bands: both pipelines (and, for Tamed, the band-specific re-autoscale of autoscale.rs:710-741,
    whose windows read p02 / p05 / p99) take the zone route; the RGB is the oracle's."""
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "force")
    rows, cols = 333, 417
    b = [f32data.resampled_scene(rows, cols, k) for k in (0, 1)]
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0], b[1], int(strategy))
    assert rc == 0
    with S.Context(0, timing=True) as c:
        rgb = c.dualpol_synrgb(b[0], b[1], strategy)
        assert np.array_equal(rgb, ref)
        assert "f32_prepass_zones" in names(c)


@pytest.mark.parametrize("strategy,bd", [(St.Clahe, Bd.U16), (St.Robust, Bd.U16), (St.Standard, Bd.U8), (St.Tamed, Bd.U8)])
@pytest.mark.parametrize("switch", ["SARPRO_HIP_NO_MAILBOX", "SARPRO_HIP_F32_LEVEL_GENERAL"])
def test_round3_cross_check_switches(strategy, bd, switch, monkeypatch):
    """The host turns through the pinned mailbox (post / prep kernels) against copy and fill commands with stream waits, and
    the folded gamma = 1 level line against the reference's expression term by term: same raster, same statistics, and the
    oracle's raster.  1500 x 1700: the zone sweep runs more than 63 turns per lane (its packed counters are unpacked in between)."""
    x = f32data.ratio_scene(1500, 1700)
    rc, ref = oracle.pipeline(x, int(bd), int(strategy))
    assert rc == 0
    outs = []
    with S.Context(0, timing=True) as c:
        for on in (False, True):
            if on:
                monkeypatch.setenv(switch, "1")
            else:
                monkeypatch.delenv(switch, raising=False)
            got = c.process_scalar_data_pipeline(x, bd, strategy)
            outs.append(got[0] if bd == Bd.U8 else got[1])
    assert np.array_equal(outs[0], outs[1]), (strategy, bd, switch)
    assert np.array_equal(outs[0], ref), (strategy, bd, switch)


def test_zone_sweep_counts_every_valid_sample_whatever_the_table_says(monkeypatch):
    """The sweep's class table decides only WHERE a sample is counted (a gap's counter or the side buffer): the valid count
    it reports must be the scene's, also when the selection stepped aside (table of one gap) and for values far outside the
    table's 64 octaves (clamped to its ends)."""
    rng = np.random.default_rng(5)
    x = np.exp(rng.normal(0.0, 1.0, (900, 1100))).astype(np.float32)
    x[::7, ::5] = 0.0           # invalid
    x[3, 3] = 1e-30             # far below the table
    x[5, 8] = 1e30              # far above it
    for strategy in (St.Robust, St.Clahe):
        rc, ref = oracle.pipeline(x, int(Bd.U8), int(strategy))
        assert rc == 0
        with S.Context(0, timing=True) as c:
            got = c.process_scalar_data_pipeline(x, Bd.U8, strategy)
            assert np.array_equal(got[0], ref), strategy
            assert "f32_prepass_zones" in names(c)


@pytest.mark.parametrize("big", [(5000, 4000, 4032), (20000, 20000, 20032)])
def test_f32_clahe_u8_speculative_blend_equals_the_f64_blend_and_the_oracle(big, monkeypatch):
    """u8 CLAHE of f32 samples: the f32 blend with a margin (f32_kernels.hip e') against the reference's f64 sequence for every
    sample (SARPRO_HIP_NO_SPEC=1) on a 5000 x 4000 pol-op raster and on BASELINE.json's 20000 x 20000, and against the oracle on a
    1500 x 1700 one."""
    x = f32data.ratio_scene(1500, 1700)
    rc, ref = oracle.pipeline(x, int(Bd.U8), int(St.Clahe))
    assert rc == 0
    with S.Context(0, timing=True) as c:
        got = c.process_scalar_data_pipeline(x, Bd.U8, St.Clahe)
        assert np.array_equal(got[0], ref)
        assert "f32_clahe_apply" in names(c)
    rows, cols, pitch = big
    q = synth.q_tables()
    with S.Context(0) as c:
        d = [torch.zeros((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + 5, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
        outs = []
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("SARPRO_HIP_NO_SPEC", env)
            else:
                monkeypatch.delenv("SARPRO_HIP_NO_SPEC", raising=False)
            o = torch.zeros((rows, pitch), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            c.dev_polop_autoscale_band(Op.LogRatio, d[0].data_ptr(), d[1].data_ptr(), True, rows, cols, pitch, St.Clahe, Bd.U8, o.data_ptr(), pitch,
                                       want_stats=False)
            outs.append(o[:, :cols].clone())
        assert torch.equal(outs[0], outs[1])

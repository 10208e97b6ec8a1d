"""Fused pol-op -> autoscale (sarpro_hip_polop_autoscale_band_*): ops.rs:4-44 computed inside every pass of the f32
flavour, against oracle.polop + oracle.pipeline -- 5 operations x 7 strategies x 2 bit depths, f32 and u16 inputs, host and
device entry points, and against the unfused route (polop_f32 + autoscale_band_f32)."""
import numpy as np
import pytest

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def ref(op, a, b, bd, strategy):
    r = oracle.polop(int(op), a.astype(np.float32), b.astype(np.float32))
    rc, out = oracle.pipeline(r, int(bd), int(strategy))
    assert rc == 0
    return out


@pytest.mark.parametrize("op", list(Op))
@pytest.mark.parametrize("strategy", list(St))
@pytest.mark.parametrize("bd", list(Bd))
def test_fused_polop_autoscale_u16_inputs(ctx, op, strategy, bd):
    rows, cols = 301, 428
    a, b = synth.scene_u16(rows, cols, 0), synth.scene_u16(rows, cols, 1)
    got = ctx.polop_autoscale_band(op, a, b, bd, strategy)
    out = got[0] if bd == Bd.U8 else got[1]
    assert np.array_equal(out, ref(op, a, b, bd, strategy))


@pytest.mark.parametrize("op", list(Op))
@pytest.mark.parametrize("strategy", [St.Standard, St.Adaptive, St.Clahe, St.Tamed])
@pytest.mark.parametrize("bd", list(Bd))
def test_fused_polop_autoscale_f32_inputs_with_specials(ctx, op, strategy, bd):
    """f32 bands with the values the guards of ops.rs exist for: zeros, denormal-small divisors, negatives, equal magnitudes
    of opposite sign (n-diff denominator 0)."""
    rows, cols = 157, 273                      # odd pitch: the scalar kernels
    rng = np.random.default_rng(11)
    a = (rng.gamma(2.0, 150.0, (rows, cols))).astype(np.float32)
    b = (rng.gamma(2.0, 60.0, (rows, cols))).astype(np.float32)
    b[::7, ::5] = 0.0; b[1::11, 2::3] = 1e-11; a[3::13, ::4] = 0.0; b[5::17, 1::6] = -b[5::17, 1::6]
    a[2::19, 3::7] = -b[2::19, 3::7]
    got = ctx.polop_autoscale_band(op, a, b, bd, strategy)
    out = got[0] if bd == Bd.U8 else got[1]
    assert np.array_equal(out, ref(op, a, b, bd, strategy))


@pytest.mark.parametrize("op", [Op.LogRatio, Op.NDiff, Op.Sum])
@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust])
def test_fused_polop_dev_equals_unfused_route(ctx, op, strategy):
    rows, cols, pitch = 1200, 1736, 1792
    q = synth.q_tables()
    d = [torch.zeros((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(2):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + 5, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
    f = []
    for k in range(2):
        x = d[k].to(torch.float32)
        x[x < 0] += 65536.0
        f.append(x.contiguous())
    ratio = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda")
    ctx.dev_polop_f32(op, f[0].data_ptr(), f[1].data_ptr(), rows * pitch, ratio.data_ptr())
    for bd, dt in ((Bd.U8, torch.uint8), (Bd.U16, torch.int16)):
        want = torch.zeros((rows, pitch), dtype=dt, device="cuda")
        ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, pitch, strategy, bd, want.data_ptr(), pitch)
        for u16_in, src in ((True, d), (False, f)):
            got = torch.zeros((rows, pitch), dtype=dt, device="cuda")
            ctx.dev_polop_autoscale_band(op, src[0].data_ptr(), src[1].data_ptr(), u16_in, rows, cols, pitch, strategy, bd, got.data_ptr(), pitch)
            assert torch.equal(got[:, :cols], want[:, :cols]), (bd, u16_in)


def test_fast_division_of_the_u16_kernels_is_the_ieee_division_for_every_pair(ctx):
    """ops.rs:10-33 on u16 DN: the kernels divide with the Newton core of the IEEE division (no rescaling frame).  All 2^32 pairs,
    a / b and (a - b) / (a + b), against the compiler's correctly rounded division: no pair may differ."""
    assert ctx.selftest_polop_division() == 0


@pytest.mark.parametrize("op", [Op.LogRatio, Op.NDiff, Op.Diff])
@pytest.mark.parametrize("cols,pitch", [(1000, 1024), (1003, 1008), (1003, 1004)])
def test_zone_sweep_with_eight_or_four_samples_per_lane(op, cols, pitch, monkeypatch):
    """u16 operands, device-resident: the zone sweep reads eight samples per lane when the pitch allows (pitch % 8 == 0), four
    otherwise or with SARPRO_HIP_F32_NO_VEC8=1; ragged rows (cols % 8 != 0); every form gives the oracle's raster."""
    rows = 700
    a, b = synth.scene_u16(rows, cols, 0), synth.scene_u16(rows, cols, 1)
    want = ref(op, a, b, Bd.U16, St.Robust)
    monkeypatch.setenv("SARPRO_HIP_F32_DIRECT", "0")  # (small raster: keep it on the zone route)
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "force")
    with S.Context(0, timing=True) as c:
        d = []
        for x in (a, b):
            t = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
            t[:, :cols] = torch.from_numpy(x.view(np.int16)).cuda()
            d.append(t)
        for no8 in (False, True):
            if no8:
                monkeypatch.setenv("SARPRO_HIP_F32_NO_VEC8", "1")
            else:
                monkeypatch.delenv("SARPRO_HIP_F32_NO_VEC8", raising=False)
            o = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
            c.dev_polop_autoscale_band(op, d[0].data_ptr(), d[1].data_ptr(), True, rows, cols, pitch, St.Robust, Bd.U16, o.data_ptr(), pitch, want_stats=False)
            assert "f32_prepass_zones" in [n for n, _ in c.last_kernel_times()]
            assert np.array_equal(o[:, :cols].cpu().numpy().view(np.uint16), want), (op, cols, pitch, no8)

"""Randomised GPU parity: many small odd shapes and data distributions through the C ABI vs the oracle
(bit-exact).  Seeds are fixed so a failure is reproducible."""
import numpy as np
import pytest

import oracle
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SarproHipError
from sarpro_amd import _lib

pytestmark = pytest.mark.gpu


def random_band(rng, rows, cols):
    kind = rng.integers(0, 6)
    if kind == 0:   # Rayleigh speckle, random scale
        a = rng.rayleigh(rng.uniform(5, 3000), (rows, cols))
    elif kind == 1:  # uniform over a random window
        lo = rng.integers(0, 60000); a = rng.integers(lo, min(lo + rng.integers(1, 6000), 65536), (rows, cols))
    elif kind == 2:  # few distinct values
        a = rng.choice(rng.integers(0, 65536, 4), (rows, cols))
    elif kind == 3:  # smooth ramp + speckle (tile statistics differ strongly)
        y, x = np.mgrid[0:rows, 0:cols]
        a = (x * 40.0 / max(cols, 1) + y * 25.0 / max(rows, 1) + 1) ** 2 * rng.rayleigh(1.0, (rows, cols))
    elif kind == 4:  # mostly invalid
        a = np.where(rng.random((rows, cols)) < 0.9, 0, rng.integers(1, 5000, (rows, cols)))
    else:            # full range
        a = rng.integers(0, 65536, (rows, cols))
    a = np.clip(a, 0, 65535).astype(np.uint16)
    if rng.random() < 0.3:
        a[: rows // 3, : cols // 2] = 0
    return a


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_single_band(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    rows, cols = int(rng.integers(1, 180)), int(rng.integers(1, 260))
    dn = random_band(rng, rows, cols)
    for strategy in St:
        for bd in Bd:
            rc, ref = oracle.pipeline(dn.astype(np.float32), int(bd), int(strategy))
            if rc == oracle.ERR_UNSUPPORTED_SHAPE:
                with pytest.raises(SarproHipError) as ei:
                    ctx.process_scalar_data_pipeline(dn, bd, strategy)
                assert ei.value.code == _lib.ERR_UNSUPPORTED_SHAPE
                continue
            assert rc == 0
            u8, u16 = ctx.process_scalar_data_pipeline(dn, bd, strategy)
            got = u8 if bd == Bd.U8 else u16
            assert np.array_equal(got, ref), (seed, rows, cols, strategy, bd, int((got != ref).sum()))
            # the f32 flavour must agree on the same (integral) samples
            f8, f16 = ctx.process_scalar_data_pipeline(dn.astype(np.float32), bd, strategy)
            assert np.array_equal(f8 if bd == Bd.U8 else f16, ref), ("f32", seed, rows, cols, strategy, bd)


@pytest.mark.parametrize("seed", range(25))
def test_random_shapes_dualpol(ctx, seed):
    rng = np.random.default_rng(5000 + seed)
    rows, cols = int(rng.integers(42, 200)), int(rng.integers(42, 300))
    b1, b2 = random_band(rng, rows, cols), random_band(rng, rows, cols)
    for strategy in St:
        rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1.astype(np.float32), b2.astype(np.float32), int(strategy))
        assert rc == 0
        rgb, u1, u2 = ctx.dualpol_synrgb(b1, b2, strategy, want_u8=True)
        assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, rrgb), (seed, rows, cols, strategy)
        assert np.array_equal(ctx.dualpol_synrgb(b1, b2, strategy), rrgb), (seed, rows, cols, strategy, "no per-band outputs")


@pytest.mark.parametrize("seed", range(8))
def test_random_medium_shapes_clahe(ctx, seed):
    """Shapes with several 512-column strips and several row items per interpolation cell (the apply kernel's edge lanes,
    scratch-line stores and item boundaries at arbitrary columns), widths of every residue mod 8."""
    rng = np.random.default_rng(7000 + seed)
    rows, cols = int(rng.integers(600, 2600)), int(rng.integers(600, 2600)) | (seed & 7)
    b1, b2 = random_band(rng, rows, cols), random_band(rng, rows, cols)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1.astype(np.float32), b2.astype(np.float32), int(St.Clahe))
    assert rc == 0
    rgb, u1, u2 = ctx.dualpol_synrgb(b1, b2, St.Clahe, want_u8=True)
    assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, rrgb), (seed, rows, cols)
    rc, ref = oracle.pipeline(b1.astype(np.float32), int(Bd.U16), int(St.Clahe))
    assert rc == 0 and np.array_equal(ctx.process_scalar_data_pipeline(b1, Bd.U16, St.Clahe)[1], ref), (seed, rows, cols, "u16")


@pytest.mark.parametrize("seed", range(15))
def test_random_f32_bands(ctx, seed):
    rng = np.random.default_rng(9000 + seed)
    rows, cols = int(rng.integers(42, 150)), int(rng.integers(42, 220))
    x = np.exp(rng.normal(rng.uniform(-3, 6), rng.uniform(0.2, 3.0), (rows, cols))).astype(np.float32)
    x[rng.random((rows, cols)) < 0.05] = 0.0
    x[rng.random((rows, cols)) < 0.02] *= -1.0
    for strategy in St:
        for bd in Bd:
            rc, ref = oracle.pipeline(x, int(bd), int(strategy))
            assert rc == 0
            u8, u16 = ctx.process_scalar_data_pipeline(x, bd, strategy)
            assert np.array_equal(u8 if bd == Bd.U8 else u16, ref), (seed, rows, cols, strategy, bd)

"""GPU parity against the committed golden vectors (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_core_v1.npz")


def test_hip_path_reproduces_golden_vectors(ctx):
    g = np.load(GOLD)
    ins = {"band0": g["in_u16_band0"], "band1": g["in_u16_band1"], "flat": g["in_u16_flat"],
           "ratio": g["in_f32_ratio"], "resampled": g["in_f32_resampled"]}
    for name, x in ins.items():
        for s in St:
            for bd in Bd:
                u8, u16 = ctx.process_scalar_data_pipeline(x, bd, s)
                assert np.array_equal(u8 if bd == Bd.U8 else u16, g[f"out_{name}_s{int(s)}_b{int(bd)}"]), (name, s, bd)
    for s in St:
        rgb, u1, u2 = ctx.dualpol_synrgb(ins["band0"], ins["band1"], s, want_u8=True)
        assert np.array_equal(rgb, g[f"rgb_s{int(s)}"]) and np.array_equal(u1, g[f"rgb_u1_s{int(s)}"]) and np.array_equal(u2, g[f"rgb_u2_s{int(s)}"])
    a, b = ins["band0"].astype(np.float32), ins["band1"].astype(np.float32)
    fns = [ctx.sum_arrays, ctx.difference_arrays, ctx.ratio_arrays, ctx.normalized_diff_arrays, ctx.log_ratio_arrays]
    for op, fn in enumerate(fns):
        assert np.array_equal(fn(a, b), g[f"polop_{op}"])
    assert np.array_equal(ctx.autoscale_db_image_tamed_synrgb_u8(ins["band0"], True), g["tamed_copol"])
    assert np.array_equal(ctx.autoscale_db_image_tamed_synrgb_u8(ins["band1"], False), g["tamed_crosspol"])


GOLD2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_core_v2_f32flow.npz")


def test_hip_path_reproduces_the_f32_flow_golden_vectors(ctx):
    """The reference's default flow (bands resampled on read -> per-band u8 -> resize -> pad -> synRGB), api/mod.rs:404-437 and
    save.rs:317-367 variants, every strategy."""
    g = np.load(GOLD2)
    b1, b2 = g["in_f32_band0"], g["in_f32_band1"]
    for s in St:
        for plain in (0, 1):
            for ti, (target, pad) in enumerate(((40, True), (None, False))):
                rgb, _ = ctx.dualpol_synrgb_resized_f32(b1, b2, s, target, pad, plain_pipeline=bool(plain))
                assert np.array_equal(rgb, g[f"rgb_s{int(s)}_plain{plain}_t{ti}"]), (s, plain, ti)

"""BASELINE.json `configs[0..4]`, one parity test each (the bench line is configs' headline; these are the others, at the
full size where the oracle finishes in seconds and at a reduced size otherwise).  Everything through the C ABI, bit-exact
against the oracle."""
import numpy as np
import pytest

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, SyntheticRgbMode as Mode, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_config0_single_band_2048_standard_u8(ctx):
    """configs[0]: single VV band 2048 x 2048 u16 -> u8, Standard autoscale (the grayscale JPEG's raster), FULL size."""
    dn = synth.scene_u16(2048, 2048, 0)
    u8, _ = ctx.process_scalar_data_pipeline(dn, Bd.U8, St.Standard)
    rc, ref = oracle.pipeline(dn.astype(np.float32), 0, int(St.Standard))
    assert rc == 0 and np.array_equal(u8, ref)
    u8f, _ = ctx.process_scalar_data_pipeline(dn.astype(np.float32), Bd.U8, St.Standard)  # the f32 flavour GDAL would hand over
    assert np.array_equal(u8f, ref)


def test_config1_dualpol_robust_resized_synrgb(ctx):
    """configs[1]: dual-pol -> Robust autoscale -> Lanczos3 to target (2048 in the config, 256 here) -> pad -> synRGB."""
    rows, cols = 1500, 1130
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rgb, m = ctx.dualpol_synrgb_resized(b[0], b[1], St.Robust, 256, True)
    u8 = [oracle.resize_image_data_with_meta(oracle.pipeline(x.astype(np.float32), 0, int(St.Robust))[1], 256, True)[0] for x in b]
    ref = oracle.synrgb(0, int(St.Robust), u8[0], u8[1])
    assert rgb.shape == (256, 256, 3) and np.array_equal(rgb, ref)


def test_config2_clahe_u16_bands_and_log_ratio_band(ctx):
    """configs[2]: full-resolution dual-pol, CLAHE autoscale, u16 rasters of VV, VH and of the log-ratio pol-op band."""
    rows, cols = 640, 777
    vv, vh = (synth.scene_u16(rows, cols, k) for k in (0, 1))
    for band in (vv, vh):
        _, u16 = ctx.process_scalar_data_pipeline(band, Bd.U16, St.Clahe)
        rc, ref = oracle.pipeline(band.astype(np.float32), 1, int(St.Clahe))
        assert rc == 0 and np.array_equal(u16, ref)
    a, b = vv.astype(np.float32), vh.astype(np.float32)
    ratio = ctx.log_ratio_arrays(a, b)
    ref_ratio = oracle.polop(int(Op.LogRatio), a, b)
    assert np.array_equal(ratio.view(np.uint32), ref_ratio.view(np.uint32))
    _, u16r = ctx.process_scalar_data_pipeline(ratio, Bd.U16, St.Clahe)
    rc, refr = oracle.pipeline(ref_ratio, 1, int(St.Clahe))
    assert rc == 0 and np.array_equal(u16r, refr)


def test_config3_scene_row_striped_over_8_ranks(ctx):
    """configs[3]: one scene row-striped over 8 ranks (one context per rank on the same device; the integer reductions
    between the phases are summed where RCCL would all-reduce them) == the oracle's unstriped raster."""
    from test_gpu_dev import run_striped
    rows, cols = 400, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(St.Clahe))
    r0s, nrs = S.host_stripe_plan(rows, 8)
    assert list(nrs) == [50] * 8  # whole CLAHE tile rows per rank
    got = run_striped(b, rows, cols, St.Clahe, list(zip(r0s, nrs)))
    assert rc == 0 and np.array_equal(got, rrgb)


def test_config4_batch_of_scenes_resized_padded_synrgb(ctx):
    """configs[4]: a batch of scenes, one after the other on the listed devices, Lanczos3 to target (1024 in the config,
    128 here) + pad + synRGB, with BatchReport semantics (api/mod.rs:474-536)."""
    shapes = [(300, 420), (257, 333), (410, 512), (384, 384), (199, 640), (333, 222)]
    scenes = [tuple(synth.scene_u16(r, c, k, seed=synth.SEED_SCENE_A + i) for k in (0, 1)) for i, (r, c) in enumerate(shapes)]
    outs, rep, st, rc = S.batch_dualpol_synrgb_resized([0], scenes, St.Default, 128, True)
    assert rc == 0 and rep.processed == len(shapes) and rep.errors == 0 and rep.skipped == 0
    for (b1, b2), got in zip(scenes, outs):
        u8 = [oracle.resize_image_data_with_meta(oracle.pipeline(x.astype(np.float32), 0, int(St.Default))[1], 128, True)[0] for x in (b1, b2)]
        assert np.array_equal(got, oracle.synrgb(0, int(St.Default), u8[0], u8[1]))

"""GPU parity of the resize + pad row against the oracle's restatement.
Dimension / padding rules are exact restatements of resize.rs:6-30 and padding.rs:5-49.
The Lanczos3 arithmetic restates the third-party fast_image_resize crate (version not pinned by the
reference, source absent): parity with the crate itself is UNPINNED; these tests pin the HIP kernels
to the oracle bit for bit, and the oracle to a float Lanczos3 reference within the fixed-point error."""
import numpy as np
import pytest

import oracle
from sarpro_amd import AutoscaleStrategy as St, resize_output_dims, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,target,pad", [((300, 420), 140, True), ((300, 420), 140, False), ((420, 300), 128, True),
                                               ((257, 257), 64, True), ((100, 333), 333, True), ((100, 333), None, True),
                                               ((100, 333), None, False), ((90, 120), 500, False), ((513, 1030), 77, True)])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_resize_pad_matches_oracle(ctx, shape, target, pad, dtype):
    rng = np.random.default_rng(1)
    rows, cols = shape
    hi = 256 if dtype == np.uint8 else 65536
    a = (np.clip(rng.normal(0.4, 0.25, shape), 0, 1) * (hi - 1)).astype(dtype)
    a[: rows // 5] = 0
    got, m = ctx.resize_image_data_with_meta(a, target, pad)
    ref, mo = oracle.resize_image_data_with_meta(a, target, pad)
    assert got.shape == ref.shape == tuple(reversed(resize_output_dims(cols, rows, target, pad)))
    assert np.array_equal(got, ref), f"{(got != ref).sum()} px differ"
    assert (m.final_cols, m.final_rows, m.pad_left, m.pad_top) == (mo["final_cols"], mo["final_rows"], mo["pad_left"], mo["pad_top"])
    assert m.scale_x == mo["scale_x"] and m.scale_y == mo["scale_y"]


@pytest.mark.parametrize("strategy", [St.Robust, St.Clahe, St.Tamed])
@pytest.mark.parametrize("target,pad", [(128, True), (100, False), (None, True)])
def test_dualpol_resized_flow_matches_oracle(ctx, strategy, target, pad):
    """BASELINE config 2 / 5 flow: autoscale at native resolution -> Lanczos3 -> pad -> synRGB (save.rs:317-367)."""
    rows, cols = 384, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rgb, m = ctx.dualpol_synrgb_resized(b[0], b[1], strategy, target, pad)
    u8 = []
    for k in (0, 1):
        x = b[k].astype(np.float32)
        u = oracle.tamed_synrgb_u8(x, k == 0) if strategy == St.Tamed else oracle.pipeline(x, 0, int(strategy))[1]
        u8.append(oracle.resize_image_data_with_meta(u, target, pad)[0])
    ref = oracle.synrgb(0, int(strategy), u8[0], u8[1])
    assert rgb.shape == ref.shape and np.array_equal(rgb, ref)


@pytest.mark.parametrize("strategy", [St.Standard, St.Clahe])
@pytest.mark.parametrize("bit_depth", [0, 1])
@pytest.mark.parametrize("kind", ["u16", "f32"])
def test_save_processed_image_raster(ctx, strategy, bit_depth, kind):
    """save_processed_image (save.rs:23-170) minus the writer: pipeline -> resize -> pad."""
    import f32data
    from sarpro_amd import BitDepth
    x = synth.scene_u16(300, 410, 0) if kind == "u16" else f32data.resampled_scene(300, 410)
    got, m = ctx.save_processed_image_raster(x, BitDepth(bit_depth), strategy, 128, True)
    rc, full = oracle.pipeline(x.astype(np.float32), bit_depth, int(strategy))
    ref, mo = oracle.resize_image_data_with_meta(full, 128, True)
    assert rc == 0 and np.array_equal(got, ref)
    assert (m.pad_left, m.pad_top, m.scale_x, m.scale_y) == (mo["pad_left"], mo["pad_top"], mo["scale_x"], mo["scale_y"])


def test_batch_driver_counts_and_continues_on_error(ctx):
    """Batch semantics of api/mod.rs:474-536: independent scenes over worker contexts, failures counted."""
    import sarpro_amd as S
    scenes = []
    for k in range(5):
        scenes.append((synth.scene_u16(200, 260, 0, seed=synth.SEED_SCENE_A + k), synth.scene_u16(200, 260, 1, seed=synth.SEED_SCENE_A + k)))
    bad = (synth.scene_u16(9, 64, 0), synth.scene_u16(9, 64, 1))  # CLAHE tile underflow: the reference panics on it
    scenes.insert(2, bad)
    outs, rep, st, rc = S.batch_dualpol_synrgb_resized([0, 0], scenes, St.Clahe, 96, True, continue_on_error=True)
    assert rc == 0 and (rep.processed, rep.errors, rep.skipped) == (5, 1, 0)
    assert st[2] == S._lib.ERR_UNSUPPORTED_SHAPE and outs[2] is None
    for i, (b1, b2) in enumerate(scenes):
        if i == 2:
            continue
        u8 = [oracle.resize_image_data_with_meta(oracle.pipeline(b.astype(np.float32), 0, int(St.Clahe))[1], 96, True)[0] for b in (b1, b2)]
        assert np.array_equal(outs[i], oracle.synrgb(0, int(St.Clahe), u8[0], u8[1])), i
    # stop-on-error: one worker, the failing scene first -> nothing else is attempted
    outs, rep, st, rc = S.batch_dualpol_synrgb_resized([0], [bad] + scenes[:2], St.Clahe, 96, True, continue_on_error=False)
    assert rc == S._lib.ERR_UNSUPPORTED_SHAPE and (rep.processed, rep.errors, rep.skipped) == (0, 1, 2)

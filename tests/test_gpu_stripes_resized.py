"""Row stripes of one scene -> the resized, padded product (sarpro_hip_stripe_run_resized_u16; SURVEY.md 8e's halo item), run as
N ranks = N threads + contexts on ONE GPU through the in-process communicator.

Each rank holds a stripe of both DN rasters and produces a contiguous range of the final raster's rows: the per-band levels with every
global quantity all-reduced, the horizontal Lanczos3 pass on its rows, the rows its vertical windows need from the neighbours through
ONE small all-reduce of boundary zones, the vertical pass for the output rows whose window centres it holds, padding, and the
composition with the suppressed floor taken from the whole padded product's histogram (summed over the ranks).  The assembled raster
must be the CPU oracle's (autoscale -> resize -> pad -> synRGB, save.rs:317-367) bit for bit -- and so the one-piece device flow's."""
import numpy as np
import pytest
import torch

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
from test_gpu_multirank_local import run_ranks, to_dev

pytestmark = pytest.mark.gpu


def oracle_product(b, strategy, target, pad):
    u8 = []
    for k in (0, 1):
        x = b[k].astype(np.float32)
        u = oracle.tamed_synrgb_u8(x, k == 0) if strategy == St.Tamed else oracle.pipeline(x, 0, int(strategy))[1]
        u8.append(oracle.resize_image_data_with_meta(u, target, pad)[0])
    return oracle.synrgb(0, int(strategy), u8[0], u8[1])


def run_striped(b, splits, strategy, target, pad, pitch=None, attrs=None):
    rows, cols = b[0].shape
    pitch = pitch or (cols + 63) // 64 * 64
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.int16) for x in b] for r0, nr in splits]
    want = [S.host_stripe_resized_rows(rows, cols, r0, nr, target, pad) for r0, nr in splits]
    fc, fr = want[0][2], want[0][3]
    slices = [torch.zeros((max(w[1], 1) * fc * 3,), dtype=torch.uint8, device="cuda") for w in want]

    def body(c, k, r0, nr):
        return c.stripe_run_resized_u16(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, strategy, Mode.Default, target, pad, slices[k].data_ptr())
    out, names = run_ranks(splits, body, attrs)
    got = np.full((fr, fc, 3), 0xEE, np.uint8)
    covered = np.zeros(fr, np.int32)
    for (o0, on, m), w, t in zip(out, want, slices):
        assert (o0, on) == (w[0], w[1]) and (m.final_cols, m.final_rows) == (fc, fr)
        got[o0:o0 + on] = t.cpu().numpy()[: on * fc * 3].reshape(on, fc, 3)
        covered[o0:o0 + on] += 1
    assert (covered == 1).all(), covered  # the ranks' ranges tile the product
    return got, names, out


SPLITS = {"ragged+empty": [(0, 5), (5, 0), (5, 301), (306, 78)], "thin": [(0, 200), (200, 3), (203, 2), (205, 179)]}


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Tamed])
@pytest.mark.parametrize("ranks", [2, 3, 8, "ragged+empty", "thin"])
@pytest.mark.parametrize("target,pad", [(128, True), (100, False)])
def test_striped_resized_product_matches_oracle(strategy, ranks, target, pad):
    rows, cols = 384, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    ref = oracle_product(b, strategy, target, pad)
    splits = SPLITS.get(ranks) or list(zip(*S.host_stripe_plan(rows, ranks)))
    got, names, _ = run_striped(b, splits, strategy, target, pad)
    assert got.shape == ref.shape and np.array_equal(got, ref), (strategy, ranks, int((got != ref).any(axis=2).sum()))
    for nm in names:
        assert "allreduce_stripe_geometry" in nm and "allreduce_resize_halo" in nm, nm


@pytest.mark.parametrize("shape,target,pad", [((384, 520), None, True), ((384, 520), None, False), ((520, 384), 64, True), ((300, 1100), 256, True),
                                              ((1100, 300), 90, False), ((403, 520), 700, True), ((264, 264), 264, True)])
def test_striped_resized_product_shapes(shape, target, pad):
    """No resize at all (target None / equal to the long side: the stripes' rows ARE the product's, only padded), a tall scene (the
    padding is left and right, every rank's rows carry it), a wide one (padding above and below: the first and the last rank's),
    an upscale (six-tap windows), an odd pitch (the stripes are staged)."""
    rows, cols = shape
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    ref = oracle_product(b, St.Clahe, target, pad)
    splits = list(zip(*S.host_stripe_plan(rows, 3)))
    got, _, _ = run_striped(b, splits, St.Clahe, target, pad, pitch=cols + 7 if target == 256 else None)
    assert got.shape == ref.shape and np.array_equal(got, ref), (shape, target, pad, int((got != ref).any(axis=2).sum()))


def test_striped_resized_product_equals_the_one_piece_flow_at_scale():
    """25 MP -> 1024^2 on 8 ranks against the one-piece device flow (which the oracle tests pin): windows of ~30 rows across
    stripe boundaries, every kernel at a size where its grids wrap."""
    rows, cols, target = 5000, 5056, 1024
    pitch = cols
    q = synth.q_tables()
    with S.Context(0) as c:
        band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + 3, k, q, rows, cols, 0, rows, band[k].data_ptr(), pitch)
        fc, fr = S.resize_output_dims(cols, rows, target, True)
        one = torch.zeros((fr * fc * 3,), dtype=torch.uint8, device="cuda")
        c.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, target, True, one.data_ptr())
    splits = list(zip(*S.host_stripe_plan(rows, 8)))
    want = [S.host_stripe_resized_rows(rows, cols, r0, nr, target, True) for r0, nr in splits]
    slices = [torch.zeros((max(w[1], 1) * fc * 3,), dtype=torch.uint8, device="cuda") for w in want]

    def body(c, k, r0, nr):
        return c.stripe_run_resized_u16(band[0][r0:].data_ptr(), band[1][r0:].data_ptr(), rows, cols, r0, nr, pitch, St.Clahe, Mode.Default, target, True,
                                        slices[k].data_ptr())
    out, _ = run_ranks(splits, body)
    got = torch.cat([t[: on * fc * 3] for (o0, on, _), t in zip(out, slices)])
    assert [o[0] for o in out] == [w[0] for w in want] and sum(o[1] for o in out) == fr
    assert torch.equal(got, one)


def test_stripes_that_do_not_tile_the_scene_are_refused_on_every_rank():
    rows, cols = 384, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    splits = [(0, 100), (110, 274)]  # a gap of ten rows
    d = [[to_dev(x[r0:r0 + nr], 576, torch.int16) for x in b] for r0, nr in splits]
    sl = [torch.zeros((128 * 128 * 3,), dtype=torch.uint8, device="cuda") for _ in splits]
    errs = []

    def body(c, k, r0, nr):
        try:
            c.stripe_run_resized_u16(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, 576, St.Robust, Mode.Default, 128, True, sl[k].data_ptr())
        except S.SarproHipError as e:
            errs.append(str(e))
    run_ranks(splits, body)
    assert len(errs) == 2 and all("tile the scene" in e for e in errs), errs


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust])
def test_striped_resized_over_the_rccl_communicator_with_one_rank(strategy):
    """The same entry point over RCCL (ncclAllReduce on the context's stream; one rank here: the only GPU of the box): the whole scene is
    the stripe, no boundary zone exists, the geometry and floor-histogram all-reduces still run."""
    rows, cols, pitch = 384, 520, 576
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    ref = oracle_product(b, strategy, 128, True)
    with S.Context(0, timing=True) as c:
        c.comm_init(1, 0, S.comm_unique_id())
        d = [to_dev(x, pitch, torch.int16) for x in b]
        o0, on, fc, fr = S.host_stripe_resized_rows(rows, cols, 0, rows, 128, True)
        assert (o0, on, fc, fr) == (0, 128, 128, 128)
        rgb = torch.zeros((on * fc * 3,), dtype=torch.uint8, device="cuda")
        r0, n, m = c.stripe_run_resized_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, 0, rows, pitch, strategy, Mode.Default, 128, True, rgb.data_ptr())
        names = [x for x, _ in c.last_kernel_times()]
    assert (r0, n) == (0, 128) and "allreduce_stripe_geometry" in names and "allreduce_resize_halo" not in names
    assert np.array_equal(rgb.cpu().numpy().reshape(fr, fc, 3), ref)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Standard, St.Tamed])
@pytest.mark.parametrize("ranks", [3, "ragged+empty"])
def test_striped_resized_product_of_f32_bands(strategy, ranks):
    """sarpro_hip_stripe_run_resized_f32: the reference's default flow hands the raster core non-integer f32 bands (resampled on read,
    api/mod.rs:404-437: both through process_scalar_data_pipeline, resize -> pad -> synRGB); the levels come from the striped f32 chain."""
    import f32data
    rows, cols, target, pad = 384, 520, 128, True
    b = [f32data.resampled_scene(rows, cols, k) for k in (0, 1)]
    u8 = [oracle.resize_image_data_with_meta(oracle.pipeline(x, 0, int(strategy))[1], target, pad)[0] for x in b]
    ref = oracle.synrgb(0, int(strategy), u8[0], u8[1])
    splits = SPLITS.get(ranks) or list(zip(*S.host_stripe_plan(rows, ranks)))
    pitch = 576
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.float32) for x in b] for r0, nr in splits]
    want = [S.host_stripe_resized_rows(rows, cols, r0, nr, target, pad) for r0, nr in splits]
    fc, fr = want[0][2], want[0][3]
    sl = [torch.zeros((max(w[1], 1) * fc * 3,), dtype=torch.uint8, device="cuda") for w in want]
    out, _ = run_ranks(splits, lambda c, k, r0, nr: c.stripe_run_resized_f32(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, strategy, Mode.Default,
                                                                           target, pad, sl[k].data_ptr()))
    got = np.concatenate([t.cpu().numpy()[: on * fc * 3].reshape(on, fc, 3) for (o0, on, _), t in zip(out, sl)], axis=0)
    assert got.shape == ref.shape and np.array_equal(got, ref), (strategy, ranks, int((got != ref).any(axis=2).sum()))


@pytest.mark.parametrize("ranks", [2, 3, 8, "ragged+empty"])
def test_striped_resized_product_of_f32_bands_under_tamed_takes_the_band_specific_autoscale(ranks):
    """save.rs:317-367 for Tamed: each band's u8 raster comes from autoscale_db_image_tamed_synrgb_u8 (autoscale.rs:710-742: co-pol
    min(p02, p05) .. p99, cross-pol p05 .. p99, no u8 rescale), not from the pipeline's Tamed arm -- the reference's own headline flow
    (README.md:8,63).  Round 6: the striped f32 form (refused until now) -- the windows come from the same all-reduced 4096 bins."""
    import f32data
    rows, cols, target, pad = 384, 520, 128, True
    b = [f32data.resampled_scene(rows, cols, k) for k in (0, 1)]
    u8 = [oracle.resize_image_data_with_meta(oracle.tamed_synrgb_u8(x, k == 0), target, pad)[0] for k, x in enumerate(b)]
    ref = oracle.synrgb(0, int(St.Tamed), u8[0], u8[1])
    plain = oracle.synrgb(0, int(St.Tamed), *[oracle.resize_image_data_with_meta(oracle.pipeline(x, 0, int(St.Tamed))[1], target, pad)[0] for x in b])
    assert not np.array_equal(ref, plain)  # (the two flows differ on this scene: the test can tell them apart)
    splits = SPLITS.get(ranks) or list(zip(*S.host_stripe_plan(rows, ranks)))
    pitch = 576
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.float32) for x in b] for r0, nr in splits]
    want = [S.host_stripe_resized_rows(rows, cols, r0, nr, target, pad) for r0, nr in splits]
    fc, fr = want[0][2], want[0][3]
    sl = [torch.zeros((max(w[1], 1) * fc * 3,), dtype=torch.uint8, device="cuda") for w in want]
    out, _ = run_ranks(splits, lambda c, k, r0, nr: c.stripe_run_resized_f32(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, St.Tamed, Mode.Default,
                                                                           target, pad, sl[k].data_ptr(), plain_pipeline=False))
    got = np.concatenate([t.cpu().numpy()[: on * fc * 3].reshape(on, fc, 3) for (o0, on, _), t in zip(out, sl)], axis=0)
    assert got.shape == ref.shape and np.array_equal(got, ref), (ranks, int((got != ref).any(axis=2).sum()))


def test_ranks_with_different_raster_layouts_produce_the_same_product():
    """One rank's stripe at an odd pitch, one at an unaligned base, one in the aligned form: the unaligned ones are staged, so all three
    take the same chain and meet in the same collectives (a rank alone on another route would leave its peers waiting)."""
    rows, cols, target = 384, 520, 128
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    ref = oracle_product(b, St.Clahe, target, True)
    splits = list(zip(*S.host_stripe_plan(rows, 3)))
    pitches = [cols + 7, 576, 576]
    d, ptrs = [], []
    for k, (r0, nr) in enumerate(splits):
        row = []
        for x in b:
            t = torch.zeros((nr * pitches[k] + 8,), dtype=torch.int16, device="cuda")
            off = 3 if k == 1 else 0  # rank 1: base pointer 6 bytes off a 16-byte boundary
            t[off:off + nr * pitches[k]].view(nr, pitches[k])[:, :cols] = torch.from_numpy(np.ascontiguousarray(x[r0:r0 + nr].view(np.int16))).cuda()
            row.append((t, t.data_ptr() + 2 * off))
        d.append(row)
    want = [S.host_stripe_resized_rows(rows, cols, r0, nr, target, True) for r0, nr in splits]
    fc, fr = want[0][2], want[0][3]
    sl = [torch.zeros((max(w[1], 1) * fc * 3,), dtype=torch.uint8, device="cuda") for w in want]
    out, _ = run_ranks(splits, lambda c, k, r0, nr: c.stripe_run_resized_u16(d[k][0][1], d[k][1][1], rows, cols, r0, nr, pitches[k], St.Clahe, Mode.Default, target, True,
                                                                           sl[k].data_ptr()))
    got = np.concatenate([t.cpu().numpy()[: on * fc * 3].reshape(on, fc, 3) for (o0, on, _), t in zip(out, sl)], axis=0)
    assert np.array_equal(got, ref)


def test_striped_resized_argument_errors_and_geometry_of_one_rank():
    """One rank holding the whole scene: the entry point is the one-piece flow plus two trivial all-reduces; bad arguments are refused
    before any collective (a stripe outside the scene, a missing communicator, a target that collapses a dimension)."""
    rows, cols, pitch = 384, 520, 576
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    d = [to_dev(x, pitch, torch.int16) for x in b]
    rgb = torch.zeros((128 * 128 * 3,), dtype=torch.uint8, device="cuda")
    with S.Context(0) as c:
        with pytest.raises(S.SarproHipError, match="communicator"):
            c.stripe_run_resized_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, 0, rows, pitch, St.Clahe, Mode.Default, 128, True, rgb.data_ptr())
    group = S.LocalGroup(1)
    try:
        with S.Context(0) as c:
            c.comm_init_local(group, 0)
            with pytest.raises(S.SarproHipError, match="outside the scene"):
                c.stripe_run_resized_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, 10, rows, pitch, St.Clahe, Mode.Default, 128, True, rgb.data_ptr())
            with pytest.raises(S.SarproHipError):
                c.stripe_run_resized_u16(0, 0, rows, cols, 0, rows, pitch, St.Clahe, Mode.Default, 128, True, rgb.data_ptr())
    finally:
        group.close()
    # (a group is aborted by a failing rank: a fresh one for the good call)
    group = S.LocalGroup(1)
    try:
        with S.Context(0) as c:
            c.comm_init_local(group, 0)
            o0, on, m = c.stripe_run_resized_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, 0, rows, pitch, St.Clahe, Mode.Default, 128, True, rgb.data_ptr())
            assert (o0, on, m.final_cols, m.final_rows) == (0, 128, 128, 128)
            assert np.array_equal(rgb.cpu().numpy().reshape(128, 128, 3), oracle_product(b, St.Clahe, 128, True))
    finally:
        group.close()
    assert S.host_stripe_resized_rows(rows, cols, 0, rows, 128, True) == (0, 128, 128, 128)
    with pytest.raises(S.SarproHipError):
        S.host_stripe_resized_rows(rows, cols, 300, 200, 128, True)

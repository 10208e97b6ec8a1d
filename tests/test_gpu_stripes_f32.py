"""Row stripes of the f32 flavour (sarpro_hip_stripe_*_f32; SURVEY 8e row 1, autoscale.rs:35-117): a scene processed as
2 / 3 / 8 / ragged stripes gives the raster of the one-piece call and of the oracle, bit for bit -- f32 bands and
polarisation operations computed on the fly, all strategies that differ in their phases, both depths."""
import numpy as np
import pytest

import f32data
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, synth
from test_gpu_dev import reduce_device_buffers

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def to_dev(x, pitch, dtype):
    t = torch.zeros((max(x.shape[0], 1), pitch), dtype=dtype, device="cuda")
    if x.shape[0]:
        src = x.view(np.int16) if x.dtype == np.uint16 else x
        t[: x.shape[0], : x.shape[1]] = torch.from_numpy(np.ascontiguousarray(src)).cuda()
    return t


def run_striped(bands, op, rows, cols, strategy, bd, splits, pitch):
    """bands: [x] (f32 band) or [a, b] with op; one context per stripe, same GPU; reductions in the test."""
    ctxs = [S.Context(0) for _ in splits]
    keep, stripes, outs = [], [], []
    u16_in = bands[0].dtype == np.uint16
    odt = torch.uint8 if bd == Bd.U8 else torch.int16
    for c, (r0, nr) in zip(ctxs, splits):
        d = [to_dev(x[r0:r0 + nr], pitch, torch.int16 if u16_in else torch.float32) for x in bands]
        o = torch.zeros((max(nr, 1), pitch), dtype=odt, device="cuda")
        torch.cuda.synchronize()
        if op is None:
            stripes.append(c.stripe_begin_f32(d[0].data_ptr(), rows, cols, r0, nr, pitch, strategy, bd, o.data_ptr(), pitch))
        else:
            stripes.append(c.stripe_begin_polop(op, d[0].data_ptr(), d[1].data_ptr(), u16_in, rows, cols, r0, nr, pitch, strategy, bd, o.data_ptr(), pitch))
        keep.append(d); outs.append(o)
    merged = S.host_f32_merge_partials([s.phase1() for s in stripes])
    reduce_device_buffers([s.phase2(merged) for s in stripes])
    reduce_device_buffers([s.phase3() for s in stripes])
    reduce_device_buffers([s.phase4() for s in stripes])
    stats = [s.phase5() for s in stripes]
    res = []
    for o, (_, nr) in zip(outs, splits):
        a = o.cpu().numpy()[:nr, :cols]
        res.append(a.view(np.uint16) if bd == Bd.U16 else a)
    for s in stripes:
        s.end()
    for c in ctxs:
        c.close()
    return np.concatenate(res, axis=0), stats


SPLITS_RAGGED = [(0, 5), (5, 0), (5, 301), (306, 97)]


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Standard, St.Tamed])
@pytest.mark.parametrize("bd", list(Bd))
def test_f32_stripes_equal_oracle(strategy, bd):
    rows, cols = 403, 520
    x = f32data.ratio_scene(rows, cols)
    rc, ref, so = oracle.pipeline(x, int(bd), int(strategy), want_stats=True)
    assert rc == 0
    for pitch in (576, 521):
        for n in (2, 3, 8):
            r0, nr = S.host_stripe_plan(rows, n)
            got, st = run_striped([x], None, rows, cols, strategy, bd, list(zip(r0, nr)), pitch)
            assert np.array_equal(got, ref), (n, pitch)
            for s in st:  # every rank ends with the scene's statistics
                for f in ("valid_count", "min_db", "max_db", "median_db", "p01", "p99", "low_clip", "high_clip", "gamma"):
                    assert getattr(s, f) == getattr(so, f), f
    got, _ = run_striped([x], None, rows, cols, strategy, bd, SPLITS_RAGGED, 576)
    assert np.array_equal(got, ref)


def test_f32_stripes_adaptive_window_from_merged_moments():
    """Adaptive reads mean / std of dB: f64 sums whose order follows the partition (documented in the header).  The raster
    still equals the oracle's on this scene; the moments agree to rounding."""
    rows, cols = 403, 520
    x = f32data.ratio_scene(rows, cols)
    rc, ref, so = oracle.pipeline(x, int(Bd.U8), int(St.Adaptive), want_stats=True)
    r0, nr = S.host_stripe_plan(rows, 3)
    got, st = run_striped([x], None, rows, cols, St.Adaptive, Bd.U8, list(zip(r0, nr)), 576)
    assert np.array_equal(got, ref)
    assert abs(st[0].mean_db - so.mean_db) < 1e-9 and abs(st[0].std_db - so.std_db) < 1e-9


@pytest.mark.parametrize("op", [Op.LogRatio, Op.NDiff, Op.Ratio])
@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust])
def test_polop_stripes_equal_oracle(op, strategy):
    rows, cols = 317, 444
    a, b = synth.scene_u16(rows, cols, 0), synth.scene_u16(rows, cols, 1)
    r = oracle.polop(int(op), a.astype(np.float32), b.astype(np.float32))
    for bd in Bd:
        rc, ref = oracle.pipeline(r, int(bd), int(strategy))
        assert rc == 0
        r0, nr = S.host_stripe_plan(rows, 3)
        got, _ = run_striped([a, b], op, rows, cols, strategy, bd, list(zip(r0, nr)), 448)
        assert np.array_equal(got, ref), (bd, "u16")
        got, _ = run_striped([a.astype(np.float32), b.astype(np.float32)], op, rows, cols, strategy, bd, SPLITS_RAGGED[:2] + [(5, 312)], 448)
        assert np.array_equal(got, ref), (bd, "f32")


def test_f32_stripes_scene_without_valid_samples_and_constant_scene():
    rows, cols = 120, 200
    for x in (np.zeros((rows, cols), np.float32), np.full((rows, cols), 3.5, np.float32)):
        for strategy in (St.Clahe, St.Robust):
            rc, ref = oracle.pipeline(x, int(Bd.U8), int(strategy))
            assert rc == 0
            got, _ = run_striped([x], None, rows, cols, strategy, Bd.U8, [(0, 60), (60, 60)], 256)
            assert np.array_equal(got, ref)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust])
@pytest.mark.parametrize("bd", list(Bd))
def test_stripe_run_f32_over_library_communicator(strategy, bd):
    """sarpro_hip_stripe_run_f32 / _polop: the partials travel as an all-reduce(sum) of a buffer that is zero outside the
    rank's slot, then the three histogram all-reduces (RCCL on the context's stream).  One rank is all a 1-GPU box offers."""
    rows, cols, pitch = 300, 392, 448
    x = f32data.ratio_scene(rows, cols)
    rc, ref = oracle.pipeline(x, int(bd), int(strategy))
    a, b = synth.scene_u16(rows, cols, 0), synth.scene_u16(rows, cols, 1)
    rc2, ref2 = oracle.pipeline(oracle.polop(int(Op.LogRatio), a.astype(np.float32), b.astype(np.float32)), int(bd), int(strategy))
    assert rc == 0 and rc2 == 0
    odt = torch.uint8 if bd == Bd.U8 else torch.int16
    view = (lambda t: t.cpu().numpy()[:, :cols]) if bd == Bd.U8 else (lambda t: t.cpu().numpy().view(np.uint16)[:, :cols])
    with S.Context(0) as c:
        c.comm_init(1, 0, S.comm_unique_id())
        d = to_dev(x, pitch, torch.float32)
        o = torch.zeros((rows, pitch), dtype=odt, device="cuda")
        torch.cuda.synchronize()
        st = c.stripe_run_f32(d.data_ptr(), rows, cols, 0, rows, pitch, strategy, bd, o.data_ptr(), pitch)
        assert np.array_equal(view(o), ref) and st.valid_count == int((x >= S.host_f32_valid_threshold()).sum())
        da, db = to_dev(a, pitch, torch.int16), to_dev(b, pitch, torch.int16)
        torch.cuda.synchronize()
        c.stripe_run_polop(Op.LogRatio, da.data_ptr(), db.data_ptr(), True, rows, cols, 0, rows, pitch, strategy, bd, o.data_ptr(), pitch)
        assert np.array_equal(view(o), ref2)
        # a rank with an empty stripe still joins every reduction
        st = c.stripe_run_f32(0, rows, cols, rows, 0, pitch, strategy, bd, 0, pitch)
        assert st.valid_count == 0

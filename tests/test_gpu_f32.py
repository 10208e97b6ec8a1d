"""GPU parity, f32-input flavour (arbitrary float samples), through the C ABI vs the oracle.
Bit-exact for every u8/u16 raster; the f64 dB buffer within 1 ulp (tolerance written below)."""
import numpy as np
import pytest

import f32data
import oracle
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SarproHipError, SyntheticRgbMode as Mode, resize_output_dims, synth
from sarpro_amd import _lib
import sarpro_amd as S

pytestmark = pytest.mark.gpu

CASES = [("ratio", lambda: f32data.ratio_scene(257, 300)),
         ("resampled", lambda: f32data.resampled_scene(128, 203)),
         ("nasty", lambda: f32data.nasty_scene(190, 331)),
         ("integral", lambda: synth.scene_u16(100, 264, 0).astype(np.float32))]


@pytest.mark.parametrize("name,make", CASES)
@pytest.mark.parametrize("strategy", list(St))
@pytest.mark.parametrize("bit_depth", list(Bd))
def test_pipeline_f32_matches_oracle(ctx, name, make, strategy, bit_depth):
    x = make()
    u8, u16, st = ctx.process_scalar_data_pipeline(x, bit_depth, strategy, want_stats=True)
    rc, ref, so = oracle.pipeline(x, int(bit_depth), int(strategy), want_stats=True)
    assert rc == 0
    got = u8 if bit_depth == Bd.U8 else u16
    assert np.array_equal(got, ref), f"{(got != ref).sum()} px differ"
    for k in ("valid_count", "min_db", "max_db", "median_db", "p01", "p02", "p05", "p10", "p25", "p75", "p90",
              "p95", "p98", "p99", "low_clip", "high_clip", "gamma"):
        assert getattr(st, k) == getattr(so, k), k
    # mean/std: device log10 + tree sums vs Welford: informational statistics, 1e-9 relative
    assert abs(st.mean_db - so.mean_db) <= 1e-9 * max(1.0, abs(so.mean_db))
    assert abs(st.std_db - so.std_db) <= 1e-9 * max(1.0, abs(so.std_db))


def test_integral_f32_equals_u16_flavour(ctx):
    dn = synth.scene_u16(200, 328, 1)
    for strategy in (St.Clahe, St.Standard, St.Adaptive):
        for bd in Bd:
            a = ctx.process_scalar_data_pipeline(dn, bd, strategy)
            b = ctx.process_scalar_data_pipeline(dn.astype(np.float32), bd, strategy)
            k = 0 if bd == Bd.U8 else 1
            assert np.array_equal(a[k], b[k])


def test_db_mask_f32(ctx):
    x = f32data.nasty_scene(120, 200, seed=3)
    x.ravel()[:4] = [1.0, 65535.0, 0.0, 2.0]
    db, mask = ctx.process_scalar_data_inplace(x)
    rdb, rmask = oracle.db_mask(x)
    assert np.array_equal(mask, rmask.astype(bool))          # exact
    # tolerance: 2 ulp(f64) -- the device's f64 log10 and glibc's are each within ~1 ulp of the true
    # value; the north-star bar (1 ulp on an f32 dB buffer) is checked on the f32-rounded values below
    ulp = np.spacing(np.abs(rdb))
    assert np.all(np.abs(db - rdb) <= 2 * ulp), float(np.max(np.abs(db - rdb) / ulp))
    assert db.ravel()[0] == 0.0 and db.ravel()[2] == -100.0  # SURVEY 8c known answers
    assert abs(db.ravel()[1] - 48.164733037652496) <= np.spacing(48.164733037652496)
    # as an f32 buffer the two agree to 1 ulp(f32)
    assert np.all(np.abs(db.astype(np.float32) - rdb.astype(np.float32)) <= np.spacing(np.abs(rdb.astype(np.float32))))


@pytest.mark.parametrize("is_copol", [True, False])
def test_tamed_synrgb_u8_f32(ctx, is_copol):
    x = f32data.resampled_scene(150, 170, 0 if is_copol else 1)
    assert np.array_equal(ctx.autoscale_db_image_tamed_synrgb_u8(x, is_copol), oracle.tamed_synrgb_u8(x, is_copol))


@pytest.mark.parametrize("strategy", list(St))
def test_dualpol_synrgb_f32_matches_oracle(ctx, strategy):
    b1, b2 = f32data.resampled_scene(140, 230, 0), f32data.resampled_scene(140, 230, 1)
    rgb, u1, u2 = ctx.dualpol_synrgb(b1, b2, strategy, want_u8=True)
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b1, b2, int(strategy))
    assert rc == 0
    assert np.array_equal(u1, r1) and np.array_equal(u2, r2) and np.array_equal(rgb, rrgb)


def test_polop_then_clahe_u16(ctx):
    # BASELINE config 3(ii): ratio_arrays(VV, VH) -> CLAHE, U16
    a = synth.scene_u16(256, 320, 0).astype(np.float32)
    b = synth.scene_u16(256, 320, 1).astype(np.float32)
    ratio = ctx.log_ratio_arrays(a, b)
    assert np.array_equal(ratio, oracle.polop(4, a, b))
    _, u16 = ctx.process_scalar_data_pipeline(ratio, Bd.U16, St.Clahe)
    rc, ref = oracle.pipeline(ratio, 1, int(St.Clahe))
    assert rc == 0 and np.array_equal(u16, ref)


def test_f32_degenerate_and_errors(ctx):
    for x in (np.zeros((40, 50), np.float32), np.full((40, 50), -3.0, np.float32), np.full((40, 50), np.nan, np.float32),
              np.full((40, 50), 7.25, np.float32)):
        for strategy in (St.Standard, St.Clahe, St.Tamed):
            for bd in Bd:
                u8, u16 = ctx.process_scalar_data_pipeline(x, bd, strategy)
                rc, ref = oracle.pipeline(x, int(bd), int(strategy))
                assert rc == 0 and np.array_equal(u8 if bd == Bd.U8 else u16, ref)
    with pytest.raises(SarproHipError) as ei:
        ctx.process_scalar_data_pipeline(np.ones((9, 64), np.float32), Bd.U8, St.Clahe)
    assert ei.value.code == _lib.ERR_UNSUPPORTED_SHAPE


@pytest.mark.parametrize("strategy", [St.Robust, St.Standard, St.Adaptive, St.Default])
def test_u16_levels_without_the_table_equal_the_table_route(strategy, monkeypatch):
    """65535 levels: by default the samples within 1e-6 of a level boundary are queued and settled on the host with the
    reference's own arithmetic (no 65535-entry table); SARPRO_HIP_F32_LEVEL_TABLE=1 builds the table and searches it.  Same
    raster, and the oracle's.  A boundary-heavy raster (few distinct values, each ON a level boundary of some window) too."""
    import f32data
    rng = np.random.default_rng(8)
    for x in (f32data.ratio_scene(320, 410), f32data.nasty_scene(200, 333),
              rng.choice(np.float32([0.001, 0.01, 0.1, 1.0, 10.0, 100.0]), (256, 256))):
        rc, ref = oracle.pipeline(x, int(Bd.U16), int(strategy))
        assert rc == 0
        with S.Context(0, timing=True) as c:
            monkeypatch.delenv("SARPRO_HIP_F32_LEVEL_TABLE", raising=False)
            a = c.process_scalar_data_pipeline(x, Bd.U16, strategy)[1]
            monkeypatch.setenv("SARPRO_HIP_F32_LEVEL_TABLE", "1")
            b = c.process_scalar_data_pipeline(x, Bd.U16, strategy)[1]
        assert np.array_equal(a, ref) and np.array_equal(b, ref)


def test_clahe_cdfs_on_the_device_equal_the_host_twin(monkeypatch):
    """The f32 flavour's CLAHE CDFs come from the u16 chain's kernel (no host turn); SARPRO_HIP_F32_HOST_CDFS=1 is the host twin."""
    import f32data
    monkeypatch.delenv("SARPRO_HIP_F32_HOST_CDFS", raising=False)
    x = f32data.ratio_scene(333, 417)
    rc, ref = oracle.pipeline(x, int(Bd.U16), int(St.Clahe))
    with S.Context(0, timing=True) as c:
        a = c.process_scalar_data_pipeline(x, Bd.U16, St.Clahe)[1]
        assert "chain_cdfs" in [n for n, _ in c.last_kernel_times()]
        monkeypatch.setenv("SARPRO_HIP_F32_HOST_CDFS", "1")
        b = c.process_scalar_data_pipeline(x, Bd.U16, St.Clahe)[1]
        assert "chain_cdfs" not in [n for n, _ in c.last_kernel_times()]
    assert np.array_equal(a, ref) and np.array_equal(b, ref)


# ---------------------------------------------------------------------------- dual-pol f32 as a first-class path (api/mod.rs:374-449)
def _api_flow_oracle(b1, b2, strategy, target, pad, mode=0, plain=True):
    """api/mod.rs:404-437 (plain: both bands through process_scalar_data_pipeline) or save.rs:317-351 (Tamed re-autoscale), then
    resize -> pad per band and the composition on the final rasters, step by step through the oracle."""
    us = []
    for k, b in enumerate((b1, b2)):
        if not plain and strategy == St.Tamed:
            u = oracle.tamed_synrgb_u8(b, k == 0)
        else:
            rc, u = oracle.pipeline(b, 0, int(strategy))
            assert rc == 0
        u, _ = oracle.resize_image_data_with_meta(u, target, pad)
        us.append(u)
    return oracle.synrgb(mode, int(strategy), us[0], us[1]), us


@pytest.mark.parametrize("strategy", [St.Standard, St.Robust, St.Clahe, St.Tamed, St.Default])
@pytest.mark.parametrize("target,pad", [(96, True), (150, False), (None, True)])
@pytest.mark.parametrize("plain", [True, False])
def test_dualpol_f32_resized_flow_matches_oracle(ctx, strategy, target, pad, plain):
    """The reference's default flow: bands resampled on read (non-integer f32), per-band u8, resize, pad, synRGB."""
    from f32data import resampled_scene
    rows, cols = 200, 264
    b1, b2 = resampled_scene(rows, cols, 0), resampled_scene(rows, cols, 1)
    ref, _ = _api_flow_oracle(b1, b2, strategy, target, pad, plain=plain)
    rgb, m = ctx.dualpol_synrgb_resized_f32(b1, b2, strategy, target, pad, plain_pipeline=plain)
    assert rgb.shape == ref.shape and np.array_equal(rgb, ref), (strategy, target, pad, plain, int((rgb != ref).sum()))


@pytest.mark.parametrize("strategy", [St.Standard, St.Clahe, St.Tamed])
def test_dualpol_f32_device_resident_entry_points(strategy):
    import torch
    from f32data import resampled_scene
    rows, cols = 264, 392
    pitch = 448
    b = [resampled_scene(rows, cols, k) for k in (0, 1)]
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b[0], b[1], int(strategy))
    assert rc == 0
    with S.Context(0, timing=True) as c:
        d = []
        for x in b:
            t = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda")
            t[:, :cols] = torch.from_numpy(x).cuda()
            d.append(t)
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        u = [torch.zeros((rows, pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
        st = c.dev_dualpol_synrgb_f32(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch,
                                      u[0].data_ptr(), u[1].data_ptr(), pitch, want_stats=True)
        assert np.array_equal(rgb.cpu().numpy().reshape(rows, pitch, 3)[:, :cols], rrgb)
        assert np.array_equal(u[0].cpu().numpy()[:, :cols], r1) and np.array_equal(u[1].cpu().numpy()[:, :cols], r2)
        assert st[0].valid_count == int((b[0] > 1e-5).sum())
        names = [n for n, _ in c.last_kernel_times()]
        assert "compose_u8" in names and len(names) > 4  # the composite call reports every kernel of both bands
        # resized, device in / device out, api/mod.rs flow
        ref, _ = _api_flow_oracle(b[0], b[1], strategy, 100, True, plain=True)
        fc, fr = resize_output_dims(cols, rows, 100, True)
        out = torch.zeros((fr * fc * 3,), dtype=torch.uint8, device="cuda")
        c.dev_dualpol_synrgb_resized_f32(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, strategy, 100, True, out.data_ptr(), plain_pipeline=True)
        assert np.array_equal(out.cpu().numpy().reshape(fr, fc, 3), ref)


@pytest.mark.parametrize("strategy", [St.Default, St.Clahe, St.Tamed, St.Robust])
@pytest.mark.parametrize("twin", [True, False])
def test_dualpol_f32_second_band_on_the_twin_context(strategy, twin, monkeypatch):
    """Device-resident bands on a context WITHOUT the timing table: the second band runs on the context's twin (own stream and
    workspaces) from a helper thread while the caller's thread runs the first; SARPRO_HIP_NO_BAND_TWIN=1: one after the other.
    Native and resized flows, repeated calls on one context (the twin and its thread persist), the oracle's rasters."""
    import torch
    from f32data import resampled_scene
    if not twin:
        monkeypatch.setenv("SARPRO_HIP_NO_BAND_TWIN", "1")
    with S.Context(0) as c:
        for rows, cols, pitch in ((264, 392, 448), (300, 200, 256)):
            b = [resampled_scene(rows, cols, k) for k in (0, 1)]
            rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b[0], b[1], int(strategy))
            assert rc == 0
            d = []
            for x in b:
                t = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda")
                t[:, :cols] = torch.from_numpy(x).cuda()
                d.append(t)
            rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
            u = [torch.zeros((rows, pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()
            for _ in range(3):
                st = c.dev_dualpol_synrgb_f32(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch,
                                              u[0].data_ptr(), u[1].data_ptr(), pitch, want_stats=True)
                assert np.array_equal(rgb.cpu().numpy().reshape(rows, pitch, 3)[:, :cols], rrgb)
                assert np.array_equal(u[0].cpu().numpy()[:, :cols], r1) and np.array_equal(u[1].cpu().numpy()[:, :cols], r2)
                assert st[0].valid_count == int((b[0] > 1e-5).sum()) and st[1].valid_count == int((b[1] > 1e-5).sum())
            for plain in (True, False):
                ref, _ = _api_flow_oracle(b[0], b[1], strategy, 100, True, plain=plain)
                fc, fr = resize_output_dims(cols, rows, 100, True)
                out = torch.zeros((fr * fc * 3,), dtype=torch.uint8, device="cuda")
                c.dev_dualpol_synrgb_resized_f32(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, strategy, 100, True, out.data_ptr(),
                                                 plain_pipeline=plain)
                assert np.array_equal(out.cpu().numpy().reshape(fr, fc, 3), ref), (strategy, plain)


def test_batch_driver_with_f32_scenes():
    from f32data import resampled_scene
    scenes = [(resampled_scene(120 + 8 * i, 160, 0), resampled_scene(120 + 8 * i, 160, 1)) for i in range(5)]
    scenes[2] = (scenes[2][0], scenes[2][1][:, :100])  # a band of the wrong shape: null pointer -> this scene fails, the batch goes on
    outs, rep, st, rc = S.batch_dualpol_synrgb_resized_f32([0, 0], scenes, St.Robust, 64, True, plain_pipeline=True)
    assert rc == 0 and (rep.processed, rep.errors, rep.skipped) == (4, 1, 0) and st[2] != 0
    for i, (sc, o) in enumerate(zip(scenes, outs)):
        if i == 2:
            assert o is None
            continue
        ref, _ = _api_flow_oracle(sc[0], sc[1], St.Robust, 64, True, plain=True)
        assert np.array_equal(o, ref), i


@pytest.mark.parametrize("lanes", [1, 3, 4])
@pytest.mark.parametrize("strategy,plain", [(St.Clahe, True), (St.Tamed, False), (St.Robust, True)])
def test_resident_batch_of_f32_scenes_over_lanes(lanes, strategy, plain):
    """sarpro_hip_batch_dualpol_synrgb_resized_f32_dev (round 6): the reference's default flow (api/mod.rs:404-437; without the flag
    save.rs:317-367) over a directory's scenes (api/mod.rs:474-536) for bands resident in device memory, `lanes` lanes with a host thread
    each -- the host turns of the f32 chain overlap.  Twelve different scenes, twice in a row on the same context (warm lanes): every
    raster == the oracle's flow; a null band is counted and does not stop the others."""
    import torch
    from f32data import resampled_scene
    rows, cols, pitch, target = 200, 264, 320, 96
    host = [(resampled_scene(rows, cols, 0, seed=100 + i), resampled_scene(rows, cols, 1, seed=200 + i)) for i in range(12)]
    refs = [_api_flow_oracle(a, b, strategy, target, True, plain=plain)[0] for a, b in host]
    fc, fr = resize_output_dims(cols, rows, target, True)
    dev = []
    for a, b in host:
        pair = []
        for x in (a, b):
            t = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda")
            t[:, :cols] = torch.from_numpy(x).cuda()
            pair.append(t)
        dev.append(pair)
    outs = [torch.zeros((fr * fc * 3,), dtype=torch.uint8, device="cuda") for _ in host]
    torch.cuda.synchronize()
    with S.Context(0) as c:
        batch = [(d[0].data_ptr(), d[1].data_ptr(), o.data_ptr()) for d, o in zip(dev, outs)]
        for _ in range(2):
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            rep, st = c.dev_batch_dualpol_synrgb_resized_f32(batch, rows, cols, pitch, strategy, target, True, plain_pipeline=plain, lanes=lanes)
            assert rep == {"processed": 12, "skipped": 0, "errors": 0, "rc": 0} and not any(st)
            for i, o in enumerate(outs):
                assert np.array_equal(o.cpu().numpy().reshape(fr, fc, 3), refs[i]), (i, strategy, lanes)
        bad = list(batch)
        bad[5] = (0, bad[5][1], bad[5][2])
        rep, st = c.dev_batch_dualpol_synrgb_resized_f32(bad, rows, cols, pitch, strategy, target, True, plain_pipeline=plain, lanes=lanes, check=False)
        assert rep["processed"] == 11 and rep["errors"] == 1 and rep["rc"] == 0 and st[5] != 0 and sum(1 for x in st if x) == 1


# ---------------------------------------------------------------------------- small-scene direct route vs the threshold / zone routes
@pytest.mark.parametrize("name,make", CASES)
@pytest.mark.parametrize("strategy", list(St))  # (CLAHE takes the direct route for its statistics only)
@pytest.mark.parametrize("bit_depth", list(Bd))
@pytest.mark.parametrize("route", ["0", "1", "tinyqueue"])
def test_f32_direct_route_and_its_twins_match_oracle(name, make, strategy, bit_depth, route, monkeypatch):
    """Scenes up to 16 MP take the direct route (SARPRO_HIP_F32_DIRECT=1 forces it, 0 forbids it): 4096 bins and levels evaluated in f64 on the
    device, samples within 1e-6 of a boundary queued for the host's glibc -- no threshold tables.  A queue that overflows hands the band
    back to the threshold route.  Whatever the route: the oracle's raster and statistics."""
    if route == "tinyqueue":
        monkeypatch.setenv("SARPRO_HIP_F32_DIRECT", "1")
        monkeypatch.setenv("SARPRO_HIP_F32_DIRECT_QCAP", "0")
    else:
        monkeypatch.setenv("SARPRO_HIP_F32_DIRECT", route)
    x = make()
    with S.Context(0, timing=True) as c:
        u8, u16, st = c.process_scalar_data_pipeline(x, bit_depth, strategy, want_stats=True)
        names = [n for n, _ in c.last_kernel_times()]
    rc, ref, so = oracle.pipeline(x, int(bit_depth), int(strategy), want_stats=True)
    assert rc == 0
    got = u8 if bit_depth == Bd.U8 else u16
    assert np.array_equal(got, ref), (name, strategy, bit_depth, route, int((got != ref).sum()))
    assert ("f32_hist4096_direct" in names) == (route != "0"), names
    for k in ("valid_count", "min_db", "max_db", "median_db", "p01", "p02", "p05", "p25", "p75", "p95", "p98", "p99", "low_clip", "high_clip", "gamma"):
        assert getattr(st, k) == getattr(so, k), (k, route)


def test_f32_direct_route_with_many_samples_on_bin_boundaries(monkeypatch):
    """Quantised data: thousands of samples share each value, among them the minimum, the maximum and values that sit on a bin edge."""
    monkeypatch.setenv("SARPRO_HIP_F32_DIRECT", "1")
    rng = np.random.default_rng(4)
    vals = np.float32(10.0) ** (np.arange(0, 4097, 64, dtype=np.float64) / 4096.0 * 3.0).astype(np.float32)  # dB evenly over 30 dB: many land on edges
    x = rng.choice(vals, size=(300, 520)).astype(np.float32)
    x[:20] = 0.0
    with S.Context(0) as c:
        for s in (St.Standard, St.Robust, St.Adaptive, St.Tamed, St.Clahe):
            for bd in Bd:
                u8, u16 = c.process_scalar_data_pipeline(x, bd, s)
                rc, ref = oracle.pipeline(x, int(bd), int(s))
                assert rc == 0 and np.array_equal(u8 if bd == Bd.U8 else u16, ref), (s, bd)


@pytest.mark.parametrize("bit_depth", list(Bd))
def test_f32_level_queue_overflow_takes_the_table_route_and_still_returns_complete(bit_depth, monkeypatch):
    """The f64 level kernel queues the samples within 1e-6 of a level boundary for the host's glibc; a queue that overflows sends
    the band to the threshold-table route, which enqueues a copy and a second level kernel -- and the call must not return before
    those are done (`idle` stayed set and the u16 form skipped its only synchronisation).  Samples that sit exactly
    on level thresholds + a zero-capacity queue (SARPRO_HIP_F32_LEVEL_QCAP=0) force the route; the raster is read back on torch's
    stream right after the call, without any other synchronisation."""
    torch = pytest.importorskip("torch")
    rows, cols = 700, 1040
    x = f32data.resampled_scene(rows, cols).copy()
    with S.Context(0) as c:
        _, _, st = c.process_scalar_data_pipeline(x, bit_depth, St.Robust, want_stats=True)
    thr = S.host_f32_level_thresholds(st, bit_depth)
    inner = thr[np.isfinite(thr) & (thr > 0)][5:-5].astype(np.float64)
    # a threshold is the first f32 sample AT OR ABOVE a level boundary, up to one f32 step (1e-3 of a u16 level) beyond it: keep those
    # that land within 5e-7 levels of the boundary (the kernel queues what lies within 1e-6)
    nlev = 255.0 if bit_depth == Bd.U8 else 65535.0
    y = (np.clip(10.0 * np.log10(inner), st.low_clip, st.high_clip) - st.low_clip) / max(st.high_clip - st.low_clip, 1.0) * nlev
    picks = inner[np.abs(y - np.rint(y)) < 5e-7].astype(np.float32)[:60]
    assert len(picks) >= 3, len(picks)
    flat = x.ravel()
    order = np.argsort(flat)
    for t in picks:  # each threshold value replaces the sample closest to it: the distribution (and with it the window) stays put
        i = order[min(np.searchsorted(flat[order], t), flat.size - 1)]
        flat[i] = t
    rc, ref, so = oracle.pipeline(x, int(bit_depth), int(St.Robust), want_stats=True)
    assert rc == 0 and so.low_clip == st.low_clip and so.high_clip == st.high_clip  # the inserted samples ARE thresholds of this raster
    monkeypatch.setenv("SARPRO_HIP_F32_DIRECT", "1" if bit_depth == Bd.U8 else "0")  # (u8 takes the f64 level kernel on the direct route only)
    monkeypatch.setenv("SARPRO_HIP_F32_LEVEL_QCAP", "0")
    d_in = torch.from_numpy(x).cuda()
    dt = torch.uint8 if bit_depth == Bd.U8 else torch.int16
    with S.Context(0, timing=True) as c:
        for _ in range(3):
            d_out = torch.zeros((rows, cols), dtype=dt, device="cuda")
            torch.cuda.synchronize()
            c.dev_autoscale_band_f32(d_in.data_ptr(), rows, cols, cols, St.Robust, bit_depth, d_out.data_ptr(), cols, want_stats=False)
            got = d_out.cpu().numpy()  # torch's stream: nothing orders it behind the library's but the call having returned complete
            names = [n for n, _ in c.last_kernel_times()]
            assert names.count("f32_level") == 2, names  # queue overflow -> table route
            got = got.view(np.uint16) if bit_depth == Bd.U16 else got
            assert np.array_equal(got, ref), int((got != ref).sum())

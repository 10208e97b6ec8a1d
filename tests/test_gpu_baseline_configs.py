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


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs 2 and 3 at their FULL size (400 MP).  The oracle needs minutes per band for the per-pixel
# path there, so -- as test_full_size_400mp_parity_by_decomposition does for the headline -- every stage is re-derived
# independently: DN histograms with torch.bincount, statistics / windows / tables with the host half (itself checked
# against the oracle on the CPU), ALL pixels against the table, sampled pixels through the oracle's own formula, and the
# size-reducing tail (resize + pad + composition) through the oracle itself, which is fast on a 2048 x 2048 raster.
# ------------------------------------------------------------------------------------------------------------------
def _scene_400mp(ctx):
    rows = cols = 20000
    pitch = 20032
    q = synth.q_tables()
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in (0, 1):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, band[k].data_ptr(), pitch)
    return rows, cols, pitch, band


def _dn_hist(band, cols):
    dn = band[:, :cols].to(torch.int32) & 0xFFFF
    return dn, torch.bincount(dn.flatten(), minlength=65536).cpu().numpy().astype(np.uint64)


def test_config1_full_size_400mp_robust_to_2048_padded_synrgb(ctx):
    """configs[1] at 400 MP: Robust autoscale of both bands (every pixel against the host-built DN -> u8 table, statistics
    against the histogram route), then Lanczos3 to 2048^2 + pad + default synRGB against the ORACLE run on the u8 rasters."""
    rows, cols, pitch, band = _scene_400mp(ctx)
    u8 = [torch.empty((rows, pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
    rgb_full = torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    stats = ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, Mode.Default,
                                       rgb_full.data_ptr(), pitch, u8[0].data_ptr(), u8[1].data_ptr(), pitch)
    host_u8 = []
    for k in (0, 1):
        dn, hist = _dn_hist(band[k], cols)
        st = S.host_stats_from_dn_hist(hist)
        S.host_window(st, St.Robust)
        for name in ("valid_count", "min_db", "max_db", "median_db", "p01", "p25", "p75", "p99", "low_clip", "high_clip", "gamma"):
            assert getattr(st, name) == getattr(stats[k], name), name
        lut16 = S.host_level_lut_u16(st, Bd.U8)                      # level of every DN at max_val 255 (autoscale.rs:647-655)
        lv_hist = np.bincount(lut16, weights=hist.astype(np.float64), minlength=256)
        occ = np.nonzero(lv_hist)[0]
        resc = S.host_u8_rescale_lut(int(occ[0]), int(occ[-1]))       # scale_u16_to_u8 over the whole raster (autoscale.rs:348-364)
        table = torch.from_numpy(resc[lut16].astype(np.uint8)).cuda()
        assert torch.equal(u8[k][:, :cols], table[dn.long()]), f"band {k}: u8 raster differs from the table route"
        # sampled pixels through the oracle's per-pixel formula (map_window), independent of the table builder
        rng = np.random.default_rng(k)
        rs, cs = rng.integers(0, rows, 50000), rng.integers(0, cols, 50000)
        d = dn[torch.from_numpy(rs).cuda(), torch.from_numpy(cs).cuda()].cpu().numpy().astype(np.float64)
        db = 10.0 * np.log10(np.maximum(d, 1e-10))
        rng_db = max(st.high_clip - st.low_clip, 1.0)
        lvl = np.where(db > -50.0, np.clip(((np.minimum(np.maximum(db, st.low_clip), st.high_clip) - st.low_clip) / rng_db) * 255.0, 0.0, 255.0).astype(np.uint16), 0)
        got = u8[k][torch.from_numpy(rs).cuda(), torch.from_numpy(cs).cuda()].cpu().numpy()
        assert np.array_equal(got, resc[lvl])
        host_u8.append(u8[k][:, :cols].cpu().numpy())
    del rgb_full, u8
    # the flow of the config: bands -> Robust -> resize 2048 -> pad -> synRGB, in ONE library call from host rasters
    b = [x[:, :cols].cpu().numpy().view(np.uint16) for x in band]
    rgb, m = ctx.dualpol_synrgb_resized(b[0], b[1], St.Robust, 2048, True)
    small = [oracle.resize_image_data_with_meta(x, 2048, True)[0] for x in host_u8]
    ref = oracle.synrgb(0, int(St.Robust), small[0], small[1])
    assert rgb.shape == (2048, 2048, 3) and np.array_equal(rgb, ref)


def test_config2_full_size_400mp_clahe_u16_and_log_ratio_band(ctx):
    """configs[2] at 400 MP: (i) CLAHE with u16 output per band -- statistics, bins and CDFs by the histogram route, 100k
    sampled pixels through the oracle's f64 formula; (ii) the log-ratio pol-op over all pixels (IEEE f32 divide, torch) and
    CLAHE u16 of that f32 band against the same decomposition with the f32 thresholds of the host half."""
    rows, cols, pitch, band = _scene_400mp(ctx)
    th = tw = 2500
    rng = np.random.default_rng(7)
    rs, cs = rng.integers(0, rows, 100000), rng.integers(0, cols, 100000)
    rs[:3000] = rng.integers(0, 1250, 3000); cs[3000:6000] = rng.integers(0, 1250, 3000)
    rs[6000:8000] = rows - 1 - rng.integers(0, 3, 2000); cs[8000:10000] = 2500 * rng.integers(1, 8, 2000) + rng.integers(-2, 2, 2000)
    rst, cst = torch.from_numpy(rs).cuda(), torch.from_numpy(cs).cuda()

    def blend_levels(bins_at, cdfs, valid):
        rf = rs / float(th) - 0.5; cf = cs / float(tw) - 0.5
        ty = np.maximum(np.floor(rf), 0).astype(np.int64); tx = np.maximum(np.floor(cf), 0).astype(np.int64)
        dy = rf - ty; dx = cf - tx
        ty0, ty1, tx0, tx1 = np.clip(ty, 0, 7), np.clip(ty + 1, 0, 7), np.clip(tx, 0, 7), np.clip(tx + 1, 0, 7)
        top = cdfs[ty0 * 8 + tx0, bins_at] * (1.0 - dx) + cdfs[ty0 * 8 + tx1, bins_at] * dx
        bot = cdfs[ty1 * 8 + tx0, bins_at] * (1.0 - dx) + cdfs[ty1 * 8 + tx1, bins_at] * dx
        return np.where(valid, (np.clip(top * (1.0 - dy) + bot * dy, 0.0, 1.0) * 65535.0).astype(np.uint16), 0)

    out16 = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
    for k in (0, 1):  # (i)
        st_dev = ctx.dev_autoscale_band_u16(band[k].data_ptr(), rows, cols, pitch, St.Clahe, Bd.U16, out16.data_ptr(), pitch)
        dn, hist = _dn_hist(band[k], cols)
        st = S.host_stats_from_dn_hist(hist)
        S.host_window(st, St.Clahe)
        if st_dev is not None:
            for name in ("valid_count", "p01", "p99", "low_clip", "high_clip"):
                assert getattr(st, name) == getattr(st_dev, name), name
        from test_gpu_dev import clahe_bin_of_every_dn  # plain-Python restatement of pipeline.rs:18-23 + autoscale.rs:583-591, 261-264
        binlut = clahe_bin_of_every_dn(st.low_clip, st.high_clip)
        assert np.array_equal(binlut, S.host_clahe_bin_lut_u16(st))
        bl = torch.from_numpy(binlut.astype(np.int64)).cuda()
        tile_h = np.zeros((64, 256), np.uint64)
        for ty in range(8):
            for tx in range(8):
                blk = dn[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
                tile_h[ty * 8 + tx] = torch.bincount(bl[blk.long()][blk > 0].flatten(), minlength=256).cpu().numpy()
        cdfs = S.host_clahe_cdfs(tile_h, rows, cols)
        d = dn[rst, cst].cpu().numpy()
        want = blend_levels(binlut[d], cdfs, d > 0)
        got = out16[rst, cst].cpu().numpy().view(np.uint16)
        assert np.array_equal(got, want), f"band {k}: {(got != want).sum()} of {len(want)} sampled pixels differ"
        assert int((out16[:, :cols][dn == 0] != 0).sum().item()) == 0  # invalid pixels are 0 everywhere
    # (ii) log-ratio band: a / b where |b| > 1e-10 (f32 literal), else 0 (ops.rs:35-44); correctly rounded f32 divide
    f = []
    for k in (0, 1):
        x = band[k][:, :cols].to(torch.float32).contiguous()
        x[x < 0] += 65536.0
        f.append(x)
    ratio = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
    ctx.dev_polop_f32(Op.LogRatio, f[0].data_ptr(), f[1].data_ptr(), rows * cols, ratio.data_ptr())
    want_ratio = torch.where(f[1].abs() > np.float32(1e-10), f[0] / f[1], torch.zeros_like(f[0]))
    assert torch.equal(ratio.view(torch.int32), want_ratio.view(torch.int32))
    del want_ratio, f
    ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Clahe, Bd.U16, out16.data_ptr(), pitch)
    # decomposition of the f32 flavour at full size (host half: thresholds found with glibc log10, host_logic.cpp):
    # validity and min / max over all samples, the 4096-bin histogram by counting thresholds <= x (torch.bucketize),
    # percentiles from it, the CLAHE bin of every sample the same way, per-tile bin histograms, CDFs, sampled blend
    import math
    vthr = S.host_f32_valid_threshold()
    valid = ratio >= vthr
    count = int(valid.sum().item())
    vmin = float(ratio[valid].min().item()); vmax = float(ratio[valid].max().item())
    min_db, max_db = 10.0 * math.log10(max(vmin, 1e-10)), 10.0 * math.log10(max(vmax, 1e-10))
    thr = torch.from_numpy(S.host_f32_bin4096_thresholds(min_db, max_db)[1:]).cuda()
    h4096 = np.zeros(4096, np.uint64)
    for r0 in range(0, rows, 2500):
        blk = ratio[r0:r0 + 2500]
        idx = torch.bucketize(blk[valid[r0:r0 + 2500]], thr, right=True)
        h4096 += torch.bincount(idx, minlength=4096).cpu().numpy().astype(np.uint64)
    st = S.host_stats_from_bins4096(count, min_db, max_db, 0.0, 0.0, h4096)  # mean / std feed nothing on the CLAHE path
    S.host_window(st, St.Clahe)
    bthr = torch.from_numpy(S.host_f32_clahe_bin_thresholds(st)[1:]).cuda()
    tile_h = np.zeros((64, 256), np.uint64)
    for ty in range(8):
        for tx in range(8):
            blk = ratio[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            v = valid[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile_h[ty * 8 + tx] = torch.bincount(torch.bucketize(blk[v], bthr, right=True), minlength=256).cpu().numpy()
    cdfs = S.host_clahe_cdfs(tile_h, rows, cols)
    r_s = ratio[rst, cst]
    bins_s = torch.bucketize(r_s, bthr, right=True).cpu().numpy()
    want = blend_levels(bins_s, cdfs, valid[rst, cst].cpu().numpy())
    got = out16[rst, cst].cpu().numpy().view(np.uint16)
    assert np.array_equal(got, want), f"log-ratio band: {(got != want).sum()} of {len(want)} sampled pixels differ"
    assert int((out16[:, :cols][~valid] != 0).sum().item()) == 0

"""EVERY pixel of every full-size BASELINE.json configuration against the ORACLE ITSELF (oracle/sarpro_oracle.c, one thread --
the reference's behaviour).  Nothing of the product stands on the expected side here: the oracle gets the scene's samples as the
`Array2<f32>` the reference's readers hand over, runs the reference's per-pixel loops (pipeline.rs:42-67, autoscale.rs:452-659,
synthetic_rgb.rs:88-178, ops.rs:35-44) and its raster is compared with the GPU's with `==` over all 4 * 10^8 pixels.

The single-thread oracle needs 7-15 s per 400 MP band (bench.py times the whole headline at ~28 s), so the module costs about
two minutes of host time; the decomposition tests of test_gpu_dev.py / test_gpu_baseline_configs.py stay as diagnostics (they say
WHICH stage differs when one of these fails)."""
import numpy as np
import pytest

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, SyntheticRgbMode as Mode, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROWS = COLS = 20000
PITCH = 20032


@pytest.fixture(scope="module")
def scene():
    """The bench generator's scene A at the metric's size: device-resident u16 bands + the same samples as host f32 arrays
    (what the oracle is fed).  The device generator is checked against the numpy one in test_gpu_u16.py."""
    q = synth.q_tables()
    c = S.Context(0, timing=True)
    band = [torch.empty((ROWS, PITCH), dtype=torch.int16, device="cuda") for _ in range(2)]
    host = []
    for k in (0, 1):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, ROWS, COLS, 0, ROWS, band[k].data_ptr(), PITCH)
        host.append(band[k][:, :COLS].contiguous().cpu().numpy().view(np.uint16).astype(np.float32))
    yield c, band, host
    c.close()


@pytest.fixture(scope="module")
def headline_ref(scene):
    """The oracle's raster of the headline (CLAHE u8 x 2 + suppressed synRGB of scene A at 20000 x 20000): ~27 s, computed once for
    the tests that compare against it."""
    _, _, host = scene
    rc, ref, r1, r2 = oracle.dualpol_synrgb(host[0], host[1], int(St.Clahe))
    assert rc == 0
    return ref, r1, r2


def _same(dev_t, ref: np.ndarray, what: str):
    """dev_t: device tensor view [rows, cols(, 3)] ; ref: the oracle's raster.  Compared on the device, slab by slab."""
    step = 2500
    bad = 0
    for r0 in range(0, ref.shape[0], step):
        want = torch.from_numpy(ref[r0:r0 + step]).cuda()
        got = dev_t[r0:r0 + step]
        if want.dtype == torch.uint16:
            want = want.view(torch.int16)
        bad += int((got != want).sum().item())
    assert bad == 0, f"{what}: {bad} of {ref.size} raster entries differ from the oracle"


def test_headline_400mp_clahe_synrgb_equals_oracle_every_pixel(scene, headline_ref):
    """The metric's workload: calibrate + CLAHE u8 x 2 + suppressed synRGB (save.rs:317-367) at 20000 x 20000, both GPU routes
    (the fused CLAHE -> RGB pass that bench.py times, and the apply + compose route with its per-band u8 rasters) == oracle."""
    c, band, host = scene
    ref, r1, r2 = headline_ref
    rgb = torch.zeros((ROWS, PITCH * 3), dtype=torch.uint8, device="cuda")
    c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), ROWS, COLS, PITCH, St.Clahe, Mode.Default, rgb.data_ptr(), PITCH,
                             want_stats=False)
    names = [n for n, _ in c.last_kernel_times()]
    assert "clahe_rgb_fused" in names, names  # the route of the bench line
    rep = c.spec_report()
    assert rep["spec_ok"] == 1 and rep["verdict"] == 0, rep  # the prediction stood: the raster below is the fused pass's own
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref, "fused route RGB")
    # a floor predicted one level off: refuted, the second fused pass with the floor the counts point to stands (round 6); two off: the exact kernels
    for force, want in (("mispredict", "retried"), ("mispredict2", "refuted")):
        rgb.zero_()
        c.set_attr("SPEC_FORCE", force)
        c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), ROWS, COLS, PITCH, St.Clahe, Mode.Default, rgb.data_ptr(), PITCH,
                                 want_stats=False)
        c.set_attr("SPEC_FORCE", None)
        rep2 = c.spec_report()
        assert rep2["outcome"] == want and rep2["retried"] == 1 and rep2["floor_first"] == rep["floor_pred"] + (1 if force == "mispredict" else 2), rep2
        _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref, f"fused route RGB, SPEC_FORCE = {force} ({want})")
    rgb.zero_()
    u8 = [torch.zeros((ROWS, PITCH), dtype=torch.uint8, device="cuda") for _ in range(2)]
    c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), ROWS, COLS, PITCH, St.Clahe, Mode.Default, rgb.data_ptr(), PITCH,
                             u8[0].data_ptr(), u8[1].data_ptr(), PITCH)
    names = [n for n, _ in c.last_kernel_times()]
    assert "clahe_rgb_fused" not in names and "clahe_apply_u8_spec" in names, names
    _same(u8[0][:, :COLS], r1, "CLAHE u8 band 1")
    _same(u8[1][:, :COLS], r2, "CLAHE u8 band 2")
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref, "apply + compose route RGB")


def test_400mp_crop_without_level_0_takes_the_predicted_rescale_and_equals_oracle_every_pixel(scene):
    """A 400 MP raster without an invalid pixel whose bands hold no level 0 (scene A's DNs coarsened to steps of 256 and lifted off
    zero): no identity proof exists, the fused pass predicts each band's lowest level, folds the u8 rescale (autoscale.rs:348-364) into
    its tables and verifies both predictions while it composes (round 5, spec_ok = 2).  Its raster and the gated exact kernels' (the
    prediction switched off: the route round 5 found a lost-LDS-add bug in, at 36 MP and up) against the oracle, every pixel."""
    c, band, host = scene
    b = []
    for k in (0, 1):
        t = band[k].to(torch.int32) & 0xFFFF
        b.append((((t >> 8) << 8) + 77 + 40 * k).clamp_(max=32767).to(torch.int16))
        del t
    hb = [x[:, :COLS].contiguous().cpu().numpy().view(np.uint16).astype(np.float32) for x in b]
    rc, ref, r1, r2 = oracle.dualpol_synrgb(hb[0], hb[1], int(St.Clahe))
    assert rc == 0 and int(r1.min()) == 0 and int(r1.max()) == 255  # (the FINAL bands span 0..255: the rescale did its work)
    del hb
    rgb = torch.zeros((ROWS, PITCH * 3), dtype=torch.uint8, device="cuda")
    c.dev_dualpol_synrgb_u16(b[0].data_ptr(), b[1].data_ptr(), ROWS, COLS, PITCH, St.Clahe, Mode.Default, rgb.data_ptr(), PITCH, want_stats=False)
    rep = c.spec_report()
    assert rep["spec_ok"] == 2 and rep["verdict"] == 0 and rep["n_below_min"] == 0 and min(rep["min_pred"]) > 0, rep
    lv = np.concatenate([r1.ravel(), r2.ravel()])
    f = rep["floor_pred"]
    assert rep["n_lt"][0] == int((lv < f).sum()) and (f == 37 or rep["n_lt"][1] == int((lv <= f).sum())), rep
    del lv
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref, "fused route with the predicted rescale")
    rgb.zero_()
    c.set_attr("NO_SPEC_RESCALE", 1)
    try:
        c.dev_dualpol_synrgb_u16(b[0].data_ptr(), b[1].data_ptr(), ROWS, COLS, PITCH, St.Clahe, Mode.Default, rgb.data_ptr(), PITCH, want_stats=False)
        rep = c.spec_report()
        assert rep["spec_ok"] == 0 and rep["verdict"] == 1, rep
        cr = c.chain_report()
        assert [int(cr["level_hist"][k][1:].sum()) for k in range(2)] == [ROWS * COLS] * 2  # every band-pixel counted: bin 0 = pixels - others = 0
    finally:
        c.set_attr("NO_SPEC_RESCALE", None)
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref, "gated exact kernels")


def test_config1_400mp_robust_u8_and_resized_synrgb_equal_oracle_every_pixel(scene):
    """configs[1]: Robust autoscale of both 400 MP bands -> u8 (every pixel == oracle.pipeline), the native-resolution default
    synRGB of those rasters (every pixel), and the config's flow -- Lanczos3 to 2048^2 + pad + synRGB -- from the ORACLE's rasters."""
    c, band, host = scene
    ref8 = []
    for k in (0, 1):
        rc, r = oracle.pipeline(host[k], 0, int(St.Robust))
        assert rc == 0
        ref8.append(r)
    rgb = torch.zeros((ROWS, PITCH * 3), dtype=torch.uint8, device="cuda")
    u8 = [torch.zeros((ROWS, PITCH), dtype=torch.uint8, device="cuda") for _ in range(2)]
    c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), ROWS, COLS, PITCH, St.Robust, Mode.Default, rgb.data_ptr(), PITCH,
                             u8[0].data_ptr(), u8[1].data_ptr(), PITCH)
    for k in (0, 1):
        _same(u8[k][:, :COLS], ref8[k], f"Robust u8 band {k + 1}")
    del u8
    ref_rgb = oracle.synrgb(0, int(St.Robust), ref8[0], ref8[1])
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref_rgb, "Robust native-resolution synRGB")
    # the fused calibrate -> stretch -> compose pass (no per-band rasters requested) gives the same RGB
    rgb.zero_()
    c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), ROWS, COLS, PITCH, St.Robust, Mode.Default, rgb.data_ptr(), PITCH,
                             want_stats=False)
    assert "lut_compose_u16" in [n for n, _ in c.last_kernel_times()]
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref_rgb, "fused percentile pass RGB")
    del rgb, ref_rgb
    small = [oracle.resize_image_data_with_meta(x, 2048, True)[0] for x in ref8]
    want = oracle.synrgb(0, int(St.Robust), small[0], small[1])
    fc, fr = S.resize_output_dims(COLS, ROWS, 2048, True)
    out = torch.zeros((fr * fc * 3,), dtype=torch.uint8, device="cuda")
    c.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), ROWS, COLS, PITCH, St.Robust, 2048, True, out.data_ptr())
    assert (fr, fc) == (2048, 2048) and np.array_equal(out.cpu().numpy().reshape(fr, fc, 3), want)


def test_config2_400mp_clahe_u16_bands_and_log_ratio_band_equal_oracle_every_pixel(scene):
    """configs[2]: CLAHE with u16 output of VV, of VH and of the log-ratio pol-op band (ops.rs:35-44 -> pipeline.rs:42-67),
    every pixel == oracle.  The pol-op band runs both ways: computed inside the f32 flavour's passes from the u16 DN
    (sarpro_hip_polop_autoscale_band_u16_dev) and as a materialised f32 raster (polop_f32_dev -> autoscale_band_f32_dev)."""
    c, band, host = scene
    out16 = torch.zeros((ROWS, PITCH), dtype=torch.int16, device="cuda")
    for k in (0, 1):
        rc, ref = oracle.pipeline(host[k], 1, int(St.Clahe))
        assert rc == 0
        out16.zero_()
        c.dev_autoscale_band_u16(band[k].data_ptr(), ROWS, COLS, PITCH, St.Clahe, Bd.U16, out16.data_ptr(), PITCH)
        _same(out16[:, :COLS], ref, f"CLAHE u16 band {k + 1}")
        del ref
    ref_ratio = oracle.polop(int(Op.LogRatio), host[0], host[1])
    rc, ref = oracle.pipeline(ref_ratio, 1, int(St.Clahe))
    assert rc == 0
    out16.zero_()
    c.dev_polop_autoscale_band(Op.LogRatio, band[0].data_ptr(), band[1].data_ptr(), True, ROWS, COLS, PITCH, St.Clahe, Bd.U16,
                               out16.data_ptr(), PITCH, want_stats=False)
    _same(out16[:, :COLS], ref, "CLAHE u16 of the log-ratio band (pol-op inside the passes)")
    f = []
    for k in (0, 1):
        x = band[k][:, :COLS].to(torch.float32).contiguous()
        x[x < 0] += 65536.0  # the u16 bit pattern was held as int16
        f.append(x)
    ratio = torch.empty((ROWS, COLS), dtype=torch.float32, device="cuda")
    c.dev_polop_f32(Op.LogRatio, f[0].data_ptr(), f[1].data_ptr(), ROWS * COLS, ratio.data_ptr())
    del f
    _same(ratio.view(torch.int32), ref_ratio.view(np.int32), "log-ratio band (f32 bit patterns)")
    out16.zero_()
    c.dev_autoscale_band_f32(ratio.data_ptr(), ROWS, COLS, COLS, St.Clahe, Bd.U16, out16.data_ptr(), PITCH)
    _same(out16[:, :COLS], ref, "CLAHE u16 of the materialised log-ratio raster")


def _run_stripes(band, rgb, n, attrs=None):
    """The headline scene as n row stripes: one context + host thread per rank on this GPU, joined by the in-process communicator,
    every rank through the ONE-CALL stripe entry point (sarpro_hip_stripe_run_u16) on views of the resident rasters.  Returns the
    ranks' kernel names and speculation reports."""
    import threading
    r0s, nrs = S.host_stripe_plan(ROWS, n)
    group = S.LocalGroup(n)
    ctxs = [S.Context(0, timing=True) for _ in range(n)]
    for k, c in enumerate(ctxs):
        c.comm_init_local(group, k)
        for name, v in (attrs or {}).items():
            c.set_attr(name, v)
    names, reps, errs = [None] * n, [None] * n, []
    torch.cuda.synchronize()

    def work(k):
        try:
            r0, nr = int(r0s[k]), int(nrs[k])
            ctxs[k].stripe_run_u16(band[0].data_ptr() + r0 * PITCH * 2, band[1].data_ptr() + r0 * PITCH * 2, ROWS, COLS, r0, nr, PITCH, St.Clahe,
                                   Mode.Default, rgb.data_ptr() + r0 * PITCH * 3, PITCH)
            names[k] = [x for x, _ in ctxs[k].last_kernel_times()]
            reps[k] = ctxs[k].spec_report()
        except Exception as e:
            errs.append((k, repr(e)))
    ths = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    alive = [t.is_alive() for t in ths]
    assert not errs and not any(alive), (errs, alive)
    for c in ctxs:
        c.close()
    group.close()
    return list(zip(r0s, nrs)), names, reps


@pytest.mark.parametrize("ranks,force", [(8, None), (8, "mispredict"), (8, "mispredict2"), (4, None), (2, None)])
def test_config3_400mp_scene_as_row_stripes_equals_oracle_every_pixel(scene, headline_ref, ranks, force):
    """configs[3] at its stated size: ONE 400 MP scene as 8 (4, 2) row stripes of 2500 (5000, 10000) rows, each rank's single-call
    chain with its integer all-reduces (in-process communicator: the ranks are contexts of this GPU), the fused CLAHE -> RGB route on
    every rank -- item geometry, border strips and chunk boundaries of full-height stripes, which the 403-row rasters of
    test_gpu_multirank_local.py never meet -- and, forced once, the refuted route.  The assembled raster == the oracle's, every pixel."""
    c, band, host = scene
    ref, _, _ = headline_ref
    rgb = torch.zeros((ROWS, PITCH * 3), dtype=torch.uint8, device="cuda")
    splits, names, reps = _run_stripes(band, rgb, ranks, {"SPEC_FORCE": force} if force else None)
    assert [int(nr) for _, nr in splits] == [ROWS // ranks] * ranks
    for nm in names:
        assert "clahe_rgb_fused" in nm and "allreduce_sample_hist" in nm and "allreduce_spec_counts" in nm, nm
    key = [(r["spec_ok"], r["verdict"], r["floor_pred"], tuple(r["n_lt"]), r["target"], r["retried"]) for r in reps]
    assert all(k == key[0] for k in key), key  # every rank proved, predicted and decided the same
    # one level off: the second fused pass stands on every rank; two off: refuted twice, the exact kernels
    assert key[0][0] == 1 and key[0][1] == (1 if force == "mispredict2" else 0) and key[0][5] == (1 if force else 0), key[0]
    _same(rgb.view(ROWS, PITCH, 3)[:, :COLS], ref, f"{ranks} row stripes{' (forced refutation)' if force else ''}")


def test_config4_batch_of_64_scenes_to_1024_padded_synrgb_equals_oracle():
    """configs[4] at its stated size: 64 scenes dealt to four workers (device 0 listed four times: one context + host thread each, as
    eight GPUs would be listed once each), every scene -> per-band autoscale -> Lanczos3 to 1024 on the long side -> pad to 1024^2 ->
    synRGB (api/mod.rs:474-536 over save.rs:317-367), BatchReport{processed: 64}; every scene's raster == the oracle's flow."""
    n = 64
    shapes = [(3000 + 37 * (i % 7), 2600 + 53 * (i % 5)) if i % 2 else (2500 + 41 * (i % 6), 3100 + 29 * (i % 4)) for i in range(n)]
    scenes = []
    with S.Context(0) as c:
        q = synth.q_tables()
        for i, (r, cc) in enumerate(shapes):
            pitch = (cc + 63) // 64 * 64
            bands = []
            for k in (0, 1):
                t = torch.empty((r, pitch), dtype=torch.int16, device="cuda")
                c.dev_synth_scene_u16(synth.SEED_SCENE_A + 100 + i, k, q, r, cc, 0, r, t.data_ptr(), pitch)
                torch.cuda.synchronize()
                bands.append(np.ascontiguousarray(t[:, :cc].cpu().numpy().view(np.uint16)))
            scenes.append(tuple(bands))
    outs, rep, st, rc = S.batch_dualpol_synrgb_resized([0, 0, 0, 0], scenes, St.Clahe, 1024, True)
    assert rc == 0 and rep.processed == n and rep.errors == 0 and rep.skipped == 0 and not any(st)
    for i, ((b1, b2), got) in enumerate(zip(scenes, outs)):
        u8 = [oracle.resize_image_data_with_meta(oracle.pipeline(x.astype(np.float32), 0, int(St.Clahe))[1], 1024, True)[0] for x in (b1, b2)]
        want = oracle.synrgb(0, int(St.Clahe), u8[0], u8[1])
        assert got.shape == (1024, 1024, 3) and np.array_equal(got, want), (i, shapes[i])

"""GPU tests of the device-pointer entry points: rasters resident in HBM (torch only allocates),
odd pitches (scalar kernels), the synthetic generator, the row-stripe protocol, and full-size
(400 MP) parity by decomposition."""
import ctypes

import numpy as np
import pytest

import emul
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SyntheticRgbMode as Mode, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev_u16(a: np.ndarray, pitch=None):
    rows, cols = a.shape
    pitch = pitch or cols
    t = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
    t[:, :cols] = torch.from_numpy(a.view(np.int16)).cuda()
    return t


def test_synth_generator_device_equals_numpy(ctx):
    q = synth.q_tables()
    rows, cols, row0, nloc = 300, 420, 37, 200
    for band in (0, 1):
        t = torch.zeros((nloc, 448), dtype=torch.int16, device="cuda")
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, band, q, rows, cols, row0, nloc, t.data_ptr(), 448)
        got = t[:, :cols].cpu().numpy().view(np.uint16)
        assert np.array_equal(got, synth.scene_u16(rows, cols, band, row0=row0, rows_local=nloc))
    # the scenes bench.py cycles over (other class maps, block sizes, sigma sets, no wedge, no bright targets, a missing band)
    for name, off, flags, qkw, _ in synth.BENCH_SCENES:
        qs = synth.q_tables(**qkw)
        for band in (0, 1):
            t = torch.zeros((nloc, 448), dtype=torch.int16, device="cuda")
            ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + off, band, qs, rows, cols, row0, nloc, t.data_ptr(), 448, flags)
            got = t[:, :cols].cpu().numpy().view(np.uint16)
            assert np.array_equal(got, synth.scene_u16(rows, cols, band, seed=synth.SEED_SCENE_A + off, q=qs, row0=row0, rows_local=nloc, flags=flags)), name


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Standard])
@pytest.mark.parametrize("bit_depth", list(Bd))
def test_odd_pitch_takes_scalar_kernels(ctx, strategy, bit_depth):
    rows, cols = 130, 333                       # pitch 333: not a multiple of 8 -> VEC=1 kernels
    dn = synth.scene_u16(rows, cols, 0)
    d_in = dev_u16(dn)
    out = torch.zeros((rows, cols), dtype=torch.uint8 if bit_depth == Bd.U8 else torch.int16, device="cuda")
    ctx.dev_autoscale_band_u16(d_in.data_ptr(), rows, cols, cols, strategy, bit_depth, out.data_ptr(), cols)
    got = out.cpu().numpy()
    got = got if bit_depth == Bd.U8 else got.view(np.uint16)
    rc, ref = oracle.pipeline(dn.astype(np.float32), int(bit_depth), int(strategy))
    assert rc == 0 and np.array_equal(got, ref)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Tamed, St.Adaptive])
def test_dualpol_dev_odd_and_padded_pitches(ctx, strategy):
    rows, cols = 150, 273
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, rrgb, r1, r2 = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    for pitch in (cols, 320):
        d = [dev_u16(x, pitch) for x in b]
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        u = [torch.zeros((rows, pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
        ctx.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default,
                                   rgb.data_ptr(), pitch, u[0].data_ptr(), u[1].data_ptr(), pitch)
        got = rgb.cpu().numpy().reshape(rows, pitch, 3)[:, :cols]
        assert np.array_equal(u[0].cpu().numpy()[:, :cols], r1) and np.array_equal(u[1].cpu().numpy()[:, :cols], r2)
        assert np.array_equal(got, rrgb)


def test_kernel_times_are_recorded():
    with S.Context(0, timing=True) as c:
        dn = synth.scene_u16(256, 256, 0)
        c.process_scalar_data_pipeline(dn, Bd.U8, St.Clahe)
        names = [n for n, ms in c.last_kernel_times() if ms >= 0.0]
        assert "dn_hist_u16" in names and "clahe_apply_u8_spec" in names
        # events on one kernel only (what bench.py's timed region uses), then on all of them again
        c.time_only("clahe_apply_u8_spec")
        want, _ = c.process_scalar_data_pipeline(dn, Bd.U8, St.Clahe)
        names = [n for n, _ in c.last_kernel_times() if not n.startswith("host:")]
        assert names == ["clahe_apply_u8_spec"]
        c.time_only(None)
        got, _ = c.process_scalar_data_pipeline(dn, Bd.U8, St.Clahe)
        assert "dn_hist_u16" in [n for n, _ in c.last_kernel_times()] and np.array_equal(got, want)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust])
def test_stream_ordered_dev_calls_match_synchronous_ones(strategy):
    """SARPRO_HIP_CTX_ASYNC_DEV: three different scenes enqueued back to back without a host round trip give the
    rasters of the synchronous calls; the event pairs of all three are read afterwards."""
    rows, cols, pitch = 520, 712, 768
    scenes = []
    for k in range(3):
        b = [torch.zeros((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for i in range(2):
            b[i][:, :cols] = torch.from_numpy(synth.scene_u16(rows, cols, i, seed=synth.SEED_SCENE_A + k).view(np.int16)).cuda()
        scenes.append(b)
    want = []
    with S.Context(0) as c:
        for b in scenes:
            rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
            c.dev_dualpol_synrgb_u16(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch)
            want.append(rgb.cpu().numpy())
    with S.Context(0, timing=True, async_dev=True) as c:
        outs = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in scenes]
        for b, rgb in zip(scenes, outs):
            assert c.dev_dualpol_synrgb_u16(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch,
                                            want_stats=False) is None
        c.synchronize()
        names = [n for n, _ in c.last_kernel_times()]
        assert names.count("dn_hist_u16") == 3
        for rgb, w in zip(outs, want):
            assert np.array_equal(rgb.cpu().numpy(), w)
        # with statistics requested the call is synchronous again
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        st = c.dev_dualpol_synrgb_u16(scenes[0][0].data_ptr(), scenes[0][1].data_ptr(), rows, cols, pitch, strategy, Mode.Default,
                                      rgb.data_ptr(), pitch, want_stats=True)
        assert st is not None and np.array_equal(rgb.cpu().numpy(), want[0])


# ---------------------------------------------------------------------------- row stripes
_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return _hip


def reduce_device_buffers(bufs):
    """all-reduce(sum) stand-in for ranks that live in one process: D2H, add, H2D."""
    bufs = [(p, n) for p, n in bufs]
    n = bufs[0][1]
    if n == 0:
        return
    acc = np.zeros(n, np.uint64)
    for p, _ in bufs:
        tmp = np.empty(n, np.uint64)
        assert hip().hipMemcpy(tmp.ctypes.data, p, n * 8, 2) == 0
        acc += tmp
    for p, _ in bufs:
        assert hip().hipMemcpy(p, acc.ctypes.data, n * 8, 1) == 0


def run_striped(b, rows, cols, strategy, splits):
    """Process one scene as len(splits) row stripes (one context each, same GPU)."""
    ctxs = [S.Context(0) for _ in splits]
    d, stripes, rgbs = [], [], []
    pitch = (cols + 63) // 64 * 64
    for c, (r0, nr) in zip(ctxs, splits):
        dd = [dev_u16(x[r0:r0 + nr], pitch) if nr else torch.zeros((1, pitch), dtype=torch.int16, device="cuda") for x in b]
        d.append(dd)
        stripes.append(c.stripe_begin_u16(dd[0].data_ptr(), dd[1].data_ptr(), rows, cols, r0, nr, pitch, strategy, Mode.Default))
        rgbs.append(torch.zeros((max(nr, 1), pitch * 3), dtype=torch.uint8, device="cuda"))
    reduce_device_buffers([s.phase1() for s in stripes])
    reduce_device_buffers([s.phase2() for s in stripes])
    reduce_device_buffers([s.phase3() for s in stripes])
    for s, t in zip(stripes, rgbs):
        s.phase4(t.data_ptr(), pitch)
    out = [t.cpu().numpy().reshape(-1, pitch, 3)[:nr, :cols] for t, (_, nr) in zip(rgbs, splits)]
    for s in stripes:
        s.end()
    for c in ctxs:
        c.close()
    return np.concatenate(out, axis=0)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Tamed, St.Robust, St.Adaptive])
def test_row_stripes_are_bit_identical_to_one_piece(ctx, strategy):
    rows, cols = 403, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    assert rc == 0
    for n in (2, 3, 8):
        r0, nr = S.host_stripe_plan(rows, n)
        got = run_striped(b, rows, cols, strategy, list(zip(r0, nr)))
        assert np.array_equal(got, rrgb), (strategy, n)
    # stripes that are not aligned to anything, one of them empty
    got = run_striped(b, rows, cols, strategy, [(0, 5), (5, 0), (5, 301), (306, 97)])
    assert np.array_equal(got, rrgb)


# ---------------------------------------------------------------------------- full size
def clahe_bin_of_every_dn(low_clip: float, high_clip: float) -> np.ndarray:
    """CLAHE bin of each integer DN, restated from the reference in plain Python (CPython's math.log10 is glibc's log10,
    SURVEY 8c): pipeline.rs:18-23 (dB of the sample), autoscale.rs:583-591 (clip to the window, normalise by
    max(high - low, 1)), autoscale.rs:261-264 (clamp to [0, 1], round-half-away(v * 255), clamp to the bins).  DN = 0 is an
    invalid sample (dB = -100 < -50): it never reaches a bin; entry 0 is left 0."""
    import math
    rng = max(high_clip - low_clip, 1.0)
    out = np.zeros(65536, np.uint8)
    for dn in range(1, 65536):
        db = 10.0 * math.log10(max(float(dn), 1e-10))
        v = (min(max(db, low_clip), high_clip) - low_clip) / rng
        v = min(max(v, 0.0), 1.0) * 255.0
        f = math.floor(v)
        b = int(f) + (1 if v - f >= 0.5 else 0)  # f64::round: half away from zero
        out[dn] = min(max(b, 0), 255)
    return out


def test_full_size_400mp_parity_by_decomposition(ctx):
    """20000 x 20000 dual-pol CLAHE + synRGB.  The oracle cannot run this in seconds, so every stage
    is re-derived independently at full size: histograms with torch.bincount, CDFs / tables with the
    host half (itself oracle-checked on CPU), the f64 blend in numpy on 200k sampled pixels, the
    composition with torch gathers over ALL pixels."""
    rows = cols = 20000
    pitch = 20032
    q = synth.q_tables()
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in (0, 1):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, band[k].data_ptr(), pitch)
    rgb = torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    u8 = [torch.empty((rows, pitch), dtype=torch.uint8, device="cuda") for _ in range(2)]
    stats = ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default,
                                       rgb.data_ptr(), pitch, u8[0].data_ptr(), u8[1].data_ptr(), pitch)
    rng = np.random.default_rng(0)
    rs = rng.integers(0, rows, 200000)
    cs = rng.integers(0, cols, 200000)
    # include the extrapolation band (first half tile), tile seams and the last rows/cols
    rs[:2000] = rng.integers(0, 1250, 2000); cs[2000:4000] = rng.integers(0, 1250, 2000)
    rs[4000:6000] = 2500 * rng.integers(1, 8, 2000) + rng.integers(-2, 2, 2000)
    cs[6000:8000] = 2500 * rng.integers(1, 8, 2000) + rng.integers(-2, 2, 2000)
    rs[8000:9000] = rows - 1 - rng.integers(0, 3, 1000); cs[9000:10000] = cols - 1 - rng.integers(0, 3, 1000)
    final_hists = []
    th, tw = 2500, 2500
    for k in (0, 1):
        dn = band[k][:, :cols]
        dn_i = (dn.to(torch.int32) & 0xFFFF)
        hist = torch.bincount(dn_i.flatten(), minlength=65536).cpu().numpy().astype(np.uint64)
        st = S.host_stats_from_dn_hist(hist)
        S.host_window(st, St.Clahe)
        for name in ("valid_count", "min_db", "max_db", "median_db", "p01", "p99", "low_clip", "high_clip"):
            assert getattr(st, name) == getattr(stats[k], name), name
        binlut = clahe_bin_of_every_dn(st.low_clip, st.high_clip)  # numpy / math restatement: nothing of the product on this side
        assert np.array_equal(binlut, S.host_clahe_bin_lut_u16(st))  # (and the product's host half agrees with it)
        bl = torch.from_numpy(binlut.astype(np.int64)).cuda()
        tile_h = np.zeros((64, 256), np.uint64)
        for ty in range(8):
            for tx in range(8):
                blk = dn_i[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
                bins = bl[blk.long()]
                tile_h[ty * 8 + tx] = torch.bincount(bins[blk > 0].flatten(), minlength=256).cpu().numpy()
        cdfs = S.host_clahe_cdfs(tile_h, rows, cols)
        # levels of the sampled pixels through the reference formula (autoscale.rs:307-330, 602)
        d = dn_i[torch.from_numpy(rs).cuda(), torch.from_numpy(cs).cuda()].cpu().numpy().astype(np.uint16)
        rf = rs / float(th) - 0.5; cf = cs / float(tw) - 0.5
        ty = np.maximum(np.floor(rf), 0).astype(np.int64); tx = np.maximum(np.floor(cf), 0).astype(np.int64)
        dy = rf - ty; dx = cf - tx
        ty0, ty1, tx0, tx1 = np.clip(ty, 0, 7), np.clip(ty + 1, 0, 7), np.clip(tx, 0, 7), np.clip(tx + 1, 0, 7)
        bn = binlut[d]
        top = cdfs[ty0 * 8 + tx0, bn] * (1.0 - dx) + cdfs[ty0 * 8 + tx1, bn] * dx
        bot = cdfs[ty1 * 8 + tx0, bn] * (1.0 - dx) + cdfs[ty1 * 8 + tx1, bn] * dx
        lv = np.where(d > 0, (np.clip(top * (1.0 - dy) + bot * dy, 0.0, 1.0) * 255.0).astype(np.uint16), 0)
        got_u8 = u8[k][:, :cols]
        lh = torch.bincount(got_u8.flatten().long(), minlength=256).cpu().numpy().astype(np.uint64)
        assert int(lh.sum()) == rows * cols
        nz = np.nonzero(lh)[0]
        # the u8 raster is rescale(levels); with min = 0 and max = 255 present the rescale is the identity
        assert nz[0] == 0 and nz[-1] == 255
        got_s = got_u8[torch.from_numpy(rs).cuda(), torch.from_numpy(cs).cuda()].cpu().numpy()
        assert np.array_equal(got_s, lv.astype(np.uint8)), f"band {k}: {(got_s != lv).sum()} sampled px differ"
        final_hists.append(lh)
    lut_r, lut_g, lut_b, fl = S.host_synrgb_luts(St.Clahe, final_hists[0] + final_hists[1], rows * cols)
    R = torch.from_numpy(lut_r).cuda(); G = torch.from_numpy(lut_g).cuda(); B = torch.from_numpy(lut_b.reshape(-1)).cuda()
    img = rgb.view(rows, pitch, 3)[:, :cols]
    for r0 in range(0, rows, 2500):  # all pixels, in slabs to bound temporaries
        a = u8[0][r0:r0 + 2500, :cols].long(); b = u8[1][r0:r0 + 2500, :cols].long()
        water = (a <= fl) & (b <= fl)
        exp = torch.stack([R[a], G[b], B[a * 256 + b]], dim=-1)
        exp[water] = 0
        assert torch.equal(img[r0:r0 + 2500], exp)


@pytest.mark.parametrize("shape", [(20000, 20000), (3001, 2777)])
def test_speculative_apply_equals_exact_blend_full_raster(shape, monkeypatch):
    """Every pixel of the scene: the speculative f32 blend with its exact fallback (product path) against the same
    chain with every pixel through the reference's f64 blend (SARPRO_HIP_NO_SPEC=1), at BASELINE's full size."""
    rows, cols = shape
    pitch = (cols + 63) // 64 * 64
    q = synth.q_tables()
    ctx = S.Context(0, timing=True)
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for b in range(2):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + 77, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    rgb = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for which in (0, 1):
        if which:
            monkeypatch.setenv("SARPRO_HIP_NO_SPEC", "1")
        ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default,
                                   rgb[which].data_ptr(), pitch)
        names = [n for n, _ in ctx.last_kernel_times()]
        # default: the fused CLAHE -> RGB pass at 400 MP (the apply + compose route at sizes below its threshold); exact-only: one f64 apply pass
        if which:
            assert "clahe_apply_u16" in names and "clahe_rgb_fused" not in names
        else:
            assert ("clahe_rgb_fused" in names) != ("clahe_apply_u8_spec" in names)
    ctx.close()
    a, b = (t.view(rows, pitch, 3)[:, :cols] for t in rgb)
    assert int((a != b).sum().item()) == 0
    assert int(a.max().item()) > 0


def test_library_rccl_communicator_single_rank():
    """The library-owned RCCL communicator (dlopen'ed librccl, ncclAllReduce(sum, u64) on the context's
    stream) driving the stripe protocol; one rank is all a 1-GPU box offers, the N-rank arithmetic is
    covered by test_row_stripes_are_bit_identical_to_one_piece and the gloo test."""
    rows, cols = 300, 392
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(St.Clahe))
    with S.Context(0) as c:
        c.comm_init(1, 0, S.comm_unique_id())
        pitch = 448
        d = [dev_u16(x, pitch) for x in b]
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        s = c.stripe_begin_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, 0, rows, pitch, St.Clahe, Mode.Default)
        c.comm_allreduce_sum_u64(*s.phase1())
        c.comm_allreduce_sum_u64(*s.phase2())
        c.comm_allreduce_sum_u64(*s.phase3())
        s.phase4(rgb.data_ptr(), pitch)
        s.end()
        assert np.array_equal(rgb.cpu().numpy().reshape(rows, pitch, 3)[:, :cols], rrgb)


@pytest.mark.parametrize("strategy", [St.Standard, St.Robust, St.Adaptive, St.Equalized, St.Tamed, St.Default])
@pytest.mark.parametrize("shape", [(200, 320), (150, 273)])
def test_fused_lut_compose_pass_equals_unfused(ctx, strategy, shape, monkeypatch):
    """Percentile strategies, dual-pol, no per-band outputs: one fused DN,DN -> RGB kernel (7 B/px).
    It must equal the oracle and the unfused path (SARPRO_HIP_NO_FUSED=1), incl. ragged row tails."""
    rows, cols = shape
    pitch = (cols + 63) // 64 * 64
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    d = [dev_u16(x, pitch) for x in b]
    # three routes to the same raster: device-resident chain (default), host-orchestrated fused pass, unfused passes
    for no_chain, no_fused in (("0", "0"), ("1", "0"), ("1", "1")):
        monkeypatch.setenv("SARPRO_HIP_NO_CHAIN", no_chain)
        monkeypatch.setenv("SARPRO_HIP_NO_FUSED", no_fused)
        if no_fused == "0":
            monkeypatch.delenv("SARPRO_HIP_NO_FUSED")
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        with S.Context(0, timing=True) as c:
            st = c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch)
            names = [n for n, _ in c.last_kernel_times()]
        assert ("lut_compose_u16" in names) == (no_fused == "0")
        assert ("chain_stats" in names) == (no_chain == "0")
        assert np.array_equal(rgb.cpu().numpy().reshape(rows, pitch, 3)[:, :cols], rrgb), (strategy, no_chain, no_fused)
        for k in (0, 1):  # statistics and the chosen window are the reference's on every route
            so = oracle.pipeline(b[k].astype(np.float32), 0, int(strategy), want_stats=True)[2] if strategy != St.Tamed else None
            if so is not None:
                for f in ("valid_count", "min_db", "max_db", "median_db", "p01", "p25", "p75", "p99", "low_clip", "high_clip", "gamma"):
                    assert getattr(st[k], f) == getattr(so, f), (strategy, no_chain, f)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Standard, St.Tamed])
def test_stripe_run_single_call_over_library_communicator(strategy):
    """sarpro_hip_stripe_run_u16: the device chains with their all-reduces enqueued on the stream (ncclAllReduce on
    the context's stream between kernels).  One rank here.  An odd pitch is staged through library-owned aligned rasters (round 5:
    the route -- and with it the sequence of collectives -- must not depend on a rank's layout), so both pitches take the chain."""
    rows, cols = 300, 392
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    with S.Context(0, timing=True) as c:
        c.comm_init(1, 0, S.comm_unique_id())
        for pitch in (448, 393):
            d = [dev_u16(x, pitch) for x in b]
            rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
            st = c.stripe_run_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, 0, rows, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch)
            names = [n for n, _ in c.last_kernel_times()]
            assert "allreduce_dn_hist" in names and "chain_stats" in names, (pitch, names)  # the device chain, whatever the caller's pitch
            assert np.array_equal(rgb.cpu().numpy().reshape(rows, pitch, 3)[:, :cols], rrgb), (strategy, pitch)
            assert st[0].valid_count == int((b[0] > 0).sum())


def test_stripe_run_empty_stripe_still_joins_the_reductions():
    rows, cols = 256, 448
    with S.Context(0) as c:
        c.comm_init(1, 0, S.comm_unique_id())
        rgb = torch.zeros((1, cols * 3), dtype=torch.uint8, device="cuda")
        for strategy in (St.Clahe, St.Robust):
            st = c.stripe_run_u16(0, 0, rows, cols, rows, 0, cols, strategy, Mode.Default, rgb.data_ptr(), cols)
            assert st[0].valid_count == 0

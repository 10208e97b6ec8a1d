"""N > 1 ranks through the SINGLE-CALL stripe entry points (sarpro_hip_stripe_run_u16 / _f32 / _polop) on one GPU: one context and
one host thread per rank, joined by the library's in-process communicator (sarpro_hip_comm_init_local) -- the code path that
`bench.py --gpus N --mode stripe` and a multi-GPU caller take (device chain with its all-reduces between the kernels), which RCCL
cannot serve on a one-GPU box (it refuses two ranks on one device).  Every rank count gives the oracle's one-piece raster, bit
for bit (all reductions are integer sums)."""
import threading

import numpy as np
import pytest

import f32data
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, SyntheticRgbMode as Mode, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def to_dev(x, pitch, dtype):
    t = torch.zeros((max(x.shape[0], 1), pitch), dtype=dtype, device="cuda")
    if x.shape[0]:
        src = x.view(np.int16) if x.dtype == np.uint16 else x
        t[: x.shape[0], : x.shape[1]] = torch.from_numpy(np.ascontiguousarray(src)).cuda()
    return t


def run_ranks(splits, body, attrs=None):
    """One thread + context per (row0, rows_local) stripe; body(ctx, rank, row0, nr) -> anything.  Returns (results, kernel names per rank)."""
    n = len(splits)
    group = S.LocalGroup(n)
    ctxs = [S.Context(0, timing=True) for _ in range(n)]
    for k, c in enumerate(ctxs):
        c.comm_init_local(group, k)
        for name, v in (attrs or {}).items():
            c.set_attr(name, v)
    out, names, errs = [None] * n, [None] * n, []
    torch.cuda.synchronize()

    def work(k):
        try:
            out[k] = body(ctxs[k], k, *splits[k])
            names[k] = [x for x, _ in ctxs[k].last_kernel_times()]
        except Exception as e:  # (a failing rank leaves its peers in a barrier: surface it and let the join time out)
            errs.append((k, repr(e)))
    ths = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    alive = [t.is_alive() for t in ths]
    assert not errs and not any(alive), (errs, alive)
    for c in ctxs:
        c.close()
    group.close()
    return out, names


SPLITS = {2: None, 3: None, 8: None, "ragged+empty": [(0, 5), (5, 0), (5, 301), (306, 97)]}


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Tamed])
@pytest.mark.parametrize("ranks", [2, 3, 8, "ragged+empty"])
def test_stripe_run_u16_with_n_ranks_in_one_process(strategy, ranks):
    rows, cols, pitch = 403, 520, 576
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    assert rc == 0
    splits = SPLITS[ranks] or list(zip(*S.host_stripe_plan(rows, ranks)))
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.int16) for x in b] for r0, nr in splits]
    rgb = [torch.zeros((max(nr, 1), pitch * 3), dtype=torch.uint8, device="cuda") for _, nr in splits]

    def body(c, k, r0, nr):
        return c.stripe_run_u16(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, strategy, Mode.Default, rgb[k].data_ptr(), pitch)
    stats, names = run_ranks(splits, body)
    got = np.concatenate([t.cpu().numpy().reshape(-1, pitch, 3)[:nr, :cols] for t, (_, nr) in zip(rgb, splits)], axis=0)
    assert np.array_equal(got, ref), (strategy, ranks)
    assert all("allreduce_dn_hist" in nm for nm in names)  # the device chain with the all-reduces between its kernels, on every rank
    assert all(st[0].valid_count == int((b[0] > 0).sum()) for st in stats)  # the scene's statistics, on every rank


@pytest.mark.parametrize("bd", list(Bd))
@pytest.mark.parametrize("strategy", [St.Clahe, St.Standard, St.Adaptive])
def test_stripe_run_f32_and_polop_with_n_ranks_in_one_process(strategy, bd):
    rows, cols, pitch = 300, 421, 448
    x = f32data.resampled_scene(rows, cols)
    rc, ref = oracle.pipeline(x, int(bd), int(strategy))
    assert rc == 0
    odt = torch.uint8 if bd == Bd.U8 else torch.int16
    for n in (2, 3):
        splits = list(zip(*S.host_stripe_plan(rows, n)))
        d = [to_dev(x[r0:r0 + nr], pitch, torch.float32) for r0, nr in splits]
        o = [torch.zeros((max(nr, 1), pitch), dtype=odt, device="cuda") for _, nr in splits]
        run_ranks(splits, lambda c, k, r0, nr: c.stripe_run_f32(d[k].data_ptr(), rows, cols, r0, nr, pitch, strategy, bd, o[k].data_ptr(), pitch))
        got = np.concatenate([t.cpu().numpy()[:nr, :cols] for t, (_, nr) in zip(o, splits)], axis=0)
        got = got.view(np.uint16) if bd == Bd.U16 else got
        assert np.array_equal(got, ref), (strategy, bd, n)
    # the log-ratio of two u16 DN bands, computed inside the passes (BASELINE config 3(ii) as stripes)
    a, b = synth.scene_u16(rows, cols, 0), synth.scene_u16(rows, cols, 1)
    rc, ref = oracle.pipeline(oracle.polop(int(Op.LogRatio), a.astype(np.float32), b.astype(np.float32)), int(bd), int(strategy))
    assert rc == 0
    splits = list(zip(*S.host_stripe_plan(rows, 4)))
    da = [to_dev(a[r0:r0 + nr], pitch, torch.int16) for r0, nr in splits]
    db = [to_dev(b[r0:r0 + nr], pitch, torch.int16) for r0, nr in splits]
    o = [torch.zeros((max(nr, 1), pitch), dtype=odt, device="cuda") for _, nr in splits]
    run_ranks(splits, lambda c, k, r0, nr: c.stripe_run_polop(Op.LogRatio, da[k].data_ptr(), db[k].data_ptr(), True, rows, cols, r0, nr, pitch, strategy, bd,
                                                           o[k].data_ptr(), pitch))
    got = np.concatenate([t.cpu().numpy()[:nr, :cols] for t, (_, nr) in zip(o, splits)], axis=0)
    got = got.view(np.uint16) if bd == Bd.U16 else got
    assert np.array_equal(got, ref), (strategy, bd, "polop")


@pytest.mark.parametrize("force", [None, "mispredict", "mispredict,noretry", "mispredict2", "nospec", "few_values", "few_values+lowmin", "few_values+lowmin,noretry"])
@pytest.mark.parametrize("ranks", [2, 8, "ragged+empty"])
def test_row_stripes_take_the_fused_clahe_rgb_route(ranks, force):
    """A striped CLAHE scene above the speculative route's size threshold (lowered to zero here) runs the fused CLAHE -> RGB pass on
    every rank: the sampled level histogram, the valid-sample counts and the pass's verification counts are all-reduced, so all ranks
    prove the identity of the u8 rescale, predict the floor and take the verdict TOGETHER; refuted (forced here) or unproven, the
    gated exact kernels run with their level histogram reduced as well.  Whatever happens: the oracle's one-piece raster."""
    rows, cols, pitch = 403, 520, 576
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    few = force is not None and force.startswith("few_values")
    if few:  # band 2 without level 0: its lowest level is predicted, the count of bytes below it joins the all-reduce of the verification counts
        b[1] = np.random.default_rng(5).choice(np.array([40, 130, 260, 500, 700, 1000], np.uint16), size=(rows, cols))
        force = force.split("+", 1)[1] if "+" in force else None
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(St.Clahe))
    assert rc == 0
    splits = SPLITS[ranks] or list(zip(*S.host_stripe_plan(rows, ranks)))
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.int16) for x in b] for r0, nr in splits]
    rgb = [torch.zeros((max(nr, 1), pitch * 3), dtype=torch.uint8, device="cuda") for _, nr in splits]
    attrs = {"SAMPLED_HIST_MIN_PX": 0, "SAMPLE_STRIDE": 5}
    if force:
        attrs["SPEC_FORCE"] = force
    reports = [None] * len(splits)

    def body(c, k, r0, nr):
        st = c.stripe_run_u16(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, St.Clahe, Mode.Default, rgb[k].data_ptr(), pitch)
        reports[k] = c.spec_report()
        return st
    _, names = run_ranks(splits, body, attrs)
    got = np.concatenate([t.cpu().numpy().reshape(-1, pitch, 3)[:nr, :cols] for t, (_, nr) in zip(rgb, splits)], axis=0)
    assert np.array_equal(got, ref), (ranks, force, reports[0])
    for (r0, nr), nm in zip(splits, names):
        assert "allreduce_sample_hist" in nm and "allreduce_spec_counts" in nm and "allreduce_level_hist" in nm, nm
        assert ("clahe_rgb_fused" in nm) and "clahe_apply_u8_spec" not in nm, nm  # (the gated recount is timed as spec_fallback_apply)
    # every rank holds the same state: same proof, same prediction, same (summed) counts, same verdict
    key = [(r["spec_ok"], r["verdict"], r["floor_pred"], tuple(r["n_lt"]), r["target"], r["n_below_min"], tuple(r["min_pred"]), r["retried"], r["floor_first"]) for r in reports]
    assert all(k == key[0] for k in key), key
    if few:
        if force == "lowmin,noretry":   # the prediction + 1 is undercut: refuted, the exact kernels (round 5's behaviour)
            assert key[0][0] == 2 and key[0][6][1] > 0 and key[0][1] == 1 and key[0][5] > 0 and key[0][7] == 0, key[0]
        elif force == "lowmin":         # round 6: the ranks' summed presence counts give the true lowest level, every rank runs the second pass on it
            assert key[0][1] == 0 and key[0][7] == 1 and key[0][5] == 0, key[0]
        else:
            assert key[0][0] == 2 and key[0][6][0] == 0 and key[0][6][1] > 0 and key[0][5] == 0, key[0]
    if force == "nospec":
        assert key[0][0] == 0 and key[0][1] == 1
    elif force == "mispredict":  # one off: the ranks' summed counts point back, every rank runs the second pass, the second verdict accepts
        assert key[0][1] == 0 and key[0][7] == 1 and key[0][2] == key[0][8] - 1, key[0]
        assert all("allreduce_spec_counts_retry" in nm and "clahe_rgb_fused_retry" in nm for nm in names)
    elif force in ("mispredict,noretry", "mispredict2"):
        assert key[0][1] == 1 and key[0][7] == (1 if force == "mispredict2" else 0), key[0]
    if key[0][0] and force is None and key[0][1] == 0:
        u = [oracle.pipeline(x.astype(np.float32), 0, int(St.Clahe))[1] for x in b]
        lv = np.concatenate([u[0].ravel(), u[1].ravel()])
        f = key[0][2]
        assert key[0][3][0] == int((lv < f).sum()) and (f == 37 or key[0][3][1] == int((lv <= f).sum()))


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust])
def test_ranks_with_different_raster_layouts_take_the_same_route(strategy):
    """The route (and with it the sequence of collectives) follows from the layout of the rasters; a rank whose stripe is not in the
    aligned form is staged through library-owned rasters, so that ranks with different layouts still join the same collectives
    (round 4 advice: a rank that alone failed the fused pass's alignment test left its peers waiting in an all-reduce)."""
    rows, cols = 403, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    assert rc == 0
    splits = list(zip(*S.host_stripe_plan(rows, 3)))
    layouts = [(576, 576, 0), (523, 521, 1), (528, 520, 0)]  # (in_pitch, rgb_pitch_px, element offset of the first sample): aligned, odd + offset, rgb pitch % 16 != 0
    d, rgb = [], []
    for (r0, nr), (ip, rp, off) in zip(splits, layouts):
        bands = []
        for x in b:
            t = torch.zeros((max(nr, 1) * ip + 8,), dtype=torch.int16, device="cuda")
            t[off: off + nr * ip].view(nr, ip)[:, :cols] = torch.from_numpy(x[r0:r0 + nr].view(np.int16)).cuda()
            bands.append(t)
        d.append(bands)
        rgb.append(torch.zeros((max(nr, 1) * rp * 3 + 16,), dtype=torch.uint8, device="cuda"))

    def body(c, k, r0, nr):
        ip, rp, off = layouts[k]
        return c.stripe_run_u16(d[k][0].data_ptr() + 2 * off, d[k][1].data_ptr() + 2 * off, rows, cols, r0, nr, ip, strategy, Mode.Default,
                                rgb[k].data_ptr() + off, rp)
    _, names = run_ranks(splits, body, {"SAMPLED_HIST_MIN_PX": 0, "SAMPLE_STRIDE": 5})
    got = np.concatenate([t[layouts[k][2]: layouts[k][2] + nr * layouts[k][1] * 3].cpu().numpy().reshape(nr, layouts[k][1], 3)[:, :cols]
                          for k, (t, (_, nr)) in enumerate(zip(rgb, splits))], axis=0)
    assert np.array_equal(got, ref), strategy
    routes = [tuple(n for n in nm if n.startswith("allreduce_")) for nm in names]
    assert all(r == routes[0] for r in routes) and routes[0], routes  # the same collectives, in the same order, on every rank
    if strategy == St.Clahe:
        assert all("clahe_rgb_fused" in nm for nm in names), names


def test_a_failing_rank_releases_its_peers():
    """A rank that fails outside a collective (here: a null band) aborts the in-process group: its peers' all-reduces return an error
    instead of waiting for ever (sarpro_hip_local_group has no timeout; RCCL behaves like any NCCL program and is not covered)."""
    rows, cols, pitch = 403, 520, 576
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    splits = list(zip(*S.host_stripe_plan(rows, 3)))
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.int16) for x in b] for r0, nr in splits]
    rgb = [torch.zeros((max(nr, 1), pitch * 3), dtype=torch.uint8, device="cuda") for _, nr in splits]
    group = S.LocalGroup(3)
    ctxs = [S.Context(0) for _ in range(3)]
    for k, c in enumerate(ctxs):
        c.comm_init_local(group, k)
    res = [None] * 3
    torch.cuda.synchronize()

    def work(k):
        r0, nr = splits[k]
        try:
            ctxs[k].stripe_run_u16(0 if k == 1 else d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, St.Clahe, Mode.Default, rgb[k].data_ptr(), pitch)
            res[k] = "ok"
        except S.SarproHipError as e:
            res[k] = str(e)
    ths = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ths), "a peer of the failed rank is still waiting"
    assert "null raster" in res[1] and all(r != "ok" and "aborted" in r for r in (res[0], res[2])), res
    for c in ctxs:
        c.close()
    group.close()


def test_recorded_all_reduces_replay_the_rank_alone():
    """COMM_RECORD / COMM_REPLAY (the measurement aid behind bench.py's `stripe_rank_model`): a rank of a real 3-rank run keeps the
    results of its all-reduces; the same context then runs the same stripe ALONE, its all-reduces answered from the recorded sums --
    the same chain, the same route, the same stripe of the oracle's raster."""
    rows, cols, pitch = 403, 520, 576
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(St.Clahe))
    assert rc == 0
    splits = list(zip(*S.host_stripe_plan(rows, 3)))
    d = [[to_dev(x[r0:r0 + nr], pitch, torch.int16) for x in b] for r0, nr in splits]
    rgb = [torch.zeros((max(n_, 1), pitch * 3), dtype=torch.uint8, device="cuda") for _, n_ in splits]
    group = S.LocalGroup(3)
    ctxs = [S.Context(0, timing=True) for _ in range(3)]
    for k, c in enumerate(ctxs):
        c.comm_init_local(group, k)
        c.set_attr("SAMPLED_HIST_MIN_PX", 0)
        c.set_attr("SAMPLE_STRIDE", 5)
    ctxs[1].set_attr("COMM_RECORD", 1)
    torch.cuda.synchronize()

    def work(k):
        r0, nr = splits[k]
        ctxs[k].stripe_run_u16(d[k][0].data_ptr(), d[k][1].data_ptr(), rows, cols, r0, nr, pitch, St.Clahe, Mode.Default, rgb[k].data_ptr(), pitch)
    ths = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths)
    r0, nr = splits[1]
    want = ref[r0:r0 + nr]
    assert np.array_equal(rgb[1].cpu().numpy().reshape(-1, pitch, 3)[:nr, :cols], want)
    names3 = [n for n, _ in ctxs[1].last_kernel_times()]
    rep3 = ctxs[1].spec_report()
    c = ctxs[1]
    c.reset_attr("COMM_RECORD")
    for x in ctxs:
        x.comm_destroy()
    c.set_attr("COMM_REPLAY", 1)
    rgb[1].zero_()
    c.stripe_run_u16(d[1][0].data_ptr(), d[1][1].data_ptr(), rows, cols, r0, nr, pitch, St.Clahe, Mode.Default, rgb[1].data_ptr(), pitch)
    assert np.array_equal(rgb[1].cpu().numpy().reshape(-1, pitch, 3)[:nr, :cols], want)
    assert [n for n, _ in c.last_kernel_times()] == names3 and "clahe_rgb_fused" in names3
    rep1 = c.spec_report()
    assert (rep1["spec_ok"], rep1["verdict"], rep1["floor_pred"], rep1["n_lt"]) == (rep3["spec_ok"], rep3["verdict"], rep3["floor_pred"], rep3["n_lt"])
    for x in ctxs:
        x.close()
    group.close()

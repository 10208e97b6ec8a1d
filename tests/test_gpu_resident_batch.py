"""GPU parity of the resident batch (sarpro_hip_batch_dualpol_synrgb_u16_dev, csrc/pipeline.cpp): K dual-pol scenes already in HBM
through ONE context over internal lanes -- the batch loop of api/mod.rs:484-533 for resident rasters, each scene the product of
save.rs:317-367 at native resolution.  Scene i + 1's histogram chain is enqueued beside scene i's fused CLAHE -> RGB pass; every
scene takes the route it would take alone, and every raster must be the oracle's, bit for bit, whatever the lanes overlap.
"""
import numpy as np
import pytest
import torch

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode
from sarpro_amd import synth

pytestmark = pytest.mark.gpu


def make_scenes(rows, cols, pitch, defs):
    """-> (host bands per scene, device tensors per scene) for synth.BENCH_SCENES-style definitions."""
    host, dev = [], []
    for name, off, flags, qkw, _ in defs:
        q = synth.q_tables(**qkw) if qkw else None
        hb = [synth.scene_u16(rows, cols, b, seed=synth.SEED_SCENE_A + off, q=q, flags=flags) for b in range(2)]
        db = []
        for b in range(2):
            t = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
            t[:, :cols] = torch.from_numpy(hb[b].view(np.int16)).cuda()
            db.append(t)
        host.append(hb)
        dev.append(db)
    return host, dev


def oracle_rgb(hb, strategy):
    rc, rgb, _, _ = oracle.dualpol_synrgb(hb[0].astype(np.float32), hb[1].astype(np.float32), int(strategy))
    assert rc == 0
    return rgb


def rgb_of(t, rows, cols, pitch):
    return t.view(rows, pitch, 3)[:, :cols].cpu().numpy()


@pytest.mark.parametrize("lanes", [1, 2, 3])
@pytest.mark.parametrize("order", [0, 1, 3])
def test_resident_batch_matches_oracle_scene_by_scene(lanes, order, monkeypatch):
    rows, cols = 1000, 1300
    pitch = (cols + 63) // 64 * 64
    monkeypatch.setenv("SARPRO_HIP_SAMPLED_HIST_MIN_PX", "0")
    monkeypatch.setenv("SARPRO_HIP_PIPE_ORDER", str(order))
    defs = synth.BENCH_SCENES
    host, dev = make_scenes(rows, cols, pitch, defs)
    refs = [oracle_rgb(h, St.Clahe) for h in host]
    outs = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in defs]
    with S.Context(0, timing=True) as c:
        batch = [(d[0].data_ptr(), d[1].data_ptr(), o.data_ptr()) for d, o in zip(dev, outs)]
        # twice over the scenes: every lane sees several scenes, the second round on dirty workspaces
        rep, st, routes = c.dev_batch_dualpol_synrgb_u16(batch + batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=lanes)
        names = [n for n, _ in c.last_kernel_times()]
    assert rep == {"processed": 2 * len(defs), "skipped": 0, "errors": 0, "rc": 0} and not any(st)
    assert names.count("clahe_rgb_fused") == 2 * len(defs)  # the lanes' kernels are reported through the parent
    for i, d in enumerate(defs):
        assert np.array_equal(rgb_of(outs[i], rows, cols, pitch), refs[i]), (d[0], routes[i])
    assert routes[: len(defs)] == routes[len(defs):]
    assert "accepted" in routes and "n/a" not in routes
    # a band without a valid sample is level 0 everywhere: scale_u16_to_u8 is the identity (autoscale.rs:356,466-468), nothing to prove
    assert routes[[d[0] for d in defs].index("I-no-VH")] == "accepted", routes


@pytest.mark.parametrize("force,route", [("mispredict", "retried"), ("mispredict,noretry", "refuted"), ("mispredict2", "refuted")])
def test_resident_batch_forced_refutation_in_the_middle(force, route, monkeypatch):
    """Refuted scenes (SPEC_FORCE = mispredict shifts every predicted floor by one: a prediction that was right is refuted, one that was
    off by one the other way becomes right) run a second fused pass -- or, noretry / two levels off, their exact kernels -- beside the
    other lane's chains."""
    rows, cols = 700, 1100
    pitch = (cols + 63) // 64 * 64
    monkeypatch.setenv("SARPRO_HIP_SAMPLED_HIST_MIN_PX", "0")
    monkeypatch.setenv("SARPRO_HIP_SPEC_FORCE", force)
    monkeypatch.setenv("SARPRO_HIP_SAMPLE_STRIDE", "5")
    defs = synth.BENCH_SCENES[:4]
    host, dev = make_scenes(rows, cols, pitch, defs)
    outs = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in defs]
    with S.Context(0) as c:
        batch = [(d[0].data_ptr(), d[1].data_ptr(), o.data_ptr()) for d, o in zip(dev, outs)]
        rep, st, routes = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=2)
    assert rep["processed"] == len(defs) and route in routes and "n/a" not in routes, routes
    for i in range(len(defs)):
        assert np.array_equal(rgb_of(outs[i], rows, cols, pitch), oracle_rgb(host[i], St.Clahe))


@pytest.mark.parametrize("strategy", [St.Robust, St.Tamed, St.Standard])
def test_resident_batch_other_strategies(strategy):
    rows, cols = 600, 900
    pitch = (cols + 63) // 64 * 64
    defs = synth.BENCH_SCENES[:3]
    host, dev = make_scenes(rows, cols, pitch, defs)
    outs = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in defs]
    with S.Context(0) as c:
        batch = [(d[0].data_ptr(), d[1].data_ptr(), o.data_ptr()) for d, o in zip(dev, outs)]
        rep, st, routes = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, strategy, Mode.Default, pitch, lanes=2)
    assert rep["processed"] == len(defs) and routes == ["n/a"] * len(defs)
    for i in range(len(defs)):
        assert np.array_equal(rgb_of(outs[i], rows, cols, pitch), oracle_rgb(host[i], strategy))


def test_resident_batch_errors_are_counted_not_fatal():
    """BatchReport semantics of api/mod.rs:453-458, 518-526: a bad scene is counted; continue_on_error decides about the rest."""
    rows, cols = 520, 640
    pitch = 640
    defs = synth.BENCH_SCENES[:3]
    host, dev = make_scenes(rows, cols, pitch, defs)
    outs = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in defs]
    with S.Context(0) as c:
        batch = [(d[0].data_ptr(), d[1].data_ptr(), o.data_ptr()) for d, o in zip(dev, outs)]
        batch[1] = (0, batch[1][1], batch[1][2])  # a null band
        rep, st, _ = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=2, check=False)
        assert rep["processed"] == 2 and rep["errors"] == 1 and rep["skipped"] == 0 and rep["rc"] == 0 and st[1] != 0 and st[0] == st[2] == 0
        for i in (0, 2):
            assert np.array_equal(rgb_of(outs[i], rows, cols, pitch), oracle_rgb(host[i], St.Clahe))
        rep, st, _ = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=2, continue_on_error=False, check=False)
        assert rep["processed"] == 1 and rep["errors"] == 1 and rep["skipped"] == 1 and rep["rc"] != 0
        # an empty batch is a no-op
        rep, st, _ = c.dev_batch_dualpol_synrgb_u16([], rows, cols, pitch, St.Clahe, Mode.Default, pitch)
        assert rep == {"processed": 0, "skipped": 0, "errors": 0, "rc": 0}


def test_resident_batch_36mp_four_scenes_every_pixel(monkeypatch):
    """The size from which the product takes the speculative fused route by itself (>= 32 MP): four different scenes at 6000 x 6000
    through three lanes, every pixel against the oracle; then the same batch with every predicted floor shifted by one
    (SPEC_FORCE = mispredict): refuted scenes run their exact kernels beside the other lanes' chains, same rasters."""
    rows = cols = 6000
    pitch = (cols + 63) // 64 * 64
    defs = [synth.BENCH_SCENES[i] for i in (0, 3, 4, 8)]  # A, D-no-wedge, E-wide-windows (the pass's WIDE form), I-no-VH
    q0 = synth.q_tables()
    dev, refs = [], []
    with S.Context(0) as c:
        for name, off, flags, qkw, _ in defs:
            q = synth.q_tables(**qkw) if qkw else q0
            d = [torch.zeros((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
            for b in range(2):
                c.dev_synth_scene_u16(synth.SEED_SCENE_A + off, b, q, rows, cols, 0, rows, d[b].data_ptr(), pitch, flags)
            torch.cuda.synchronize()
            hb = [t[:, :cols].cpu().numpy().view(np.uint16) for t in d]
            dev.append(d)
            refs.append(oracle_rgb(hb, St.Clahe))
        outs = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in defs]
        batch = [(d[0].data_ptr(), d[1].data_ptr(), o.data_ptr()) for d, o in zip(dev, outs)]
        rep, st, routes = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=3)
        assert rep["processed"] == 4 and "n/a" not in routes and "unproven" not in routes, routes
        for i, d in enumerate(defs):
            assert np.array_equal(rgb_of(outs[i], rows, cols, pitch), refs[i]), (d[0], routes[i])
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for force, route in (("mispredict", "retried"), ("mispredict2", "refuted")):
            monkeypatch.setenv("SARPRO_HIP_SPEC_FORCE", force)
            rep, st, forced = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=3)
            assert rep["processed"] == 4 and route in forced, forced
            for i, d in enumerate(defs):
                assert np.array_equal(rgb_of(outs[i], rows, cols, pitch), refs[i]), (d[0], forced[i])
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()

"""The conflict-free exact CLAHE kernel with u16 levels out (kernels.hip 4a: sixteen copies of every bin's CDF pairs, persistent
1024-thread workgroups over items of up to 1024 rows) against the oracle and against the kernel it replaces on this route
(SARPRO_HIP_NO_U16_CF=1: kernel 4), on shapes with ragged items, items shorter than a workgroup's 16 waves, and windows of every
width; also with short items (SARPRO_HIP_U16_ITEM_ROWS), which exercise the per-item table staging hundreds of times per launch."""
import numpy as np
import pytest
import torch

import oracle
import sarpro_amd as S
from sarpro_amd import synth

pytestmark = pytest.mark.gpu
St, Bd = S.AutoscaleStrategy, S.BitDepth


def _scene(c, seed, rows, cols, flags=0, sigma=None):
    pitch = (cols + 63) // 64 * 64
    q = synth.q_tables(sigma=sigma) if sigma else synth.q_tables()
    band = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
    c.dev_synth_scene_u16(seed, 0, q, rows, cols, 0, rows, band.data_ptr(), pitch, flags)
    return band, pitch


def _run(c, band, rows, cols, pitch):
    out = torch.full((rows, pitch), -1, dtype=torch.int16, device="cuda")
    c.dev_autoscale_band_u16(band.data_ptr(), rows, cols, pitch, St.Clahe, Bd.U16, out.data_ptr(), pitch)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("shape", [(64, 512), (100, 777), (1203, 1501), (2600, 3000), (3333, 2111), (24, 4099)])
def test_cf_kernel_equals_oracle_and_kernel_4(shape):
    rows, cols = shape
    with S.Context(0) as c:
        band, pitch = _scene(c, synth.SEED_SCENE_A + rows, rows, cols)
        got = _run(c, band, rows, cols, pitch)
        c.set_attr("NO_U16_CF", 1)
        old = _run(c, band, rows, cols, pitch)
        c.reset_attr("NO_U16_CF")
        assert torch.equal(got[:, :cols], old[:, :cols])
        assert bool((got[:, cols:] == -1).all())  # nothing written beyond the scene's columns
        host = band[:, :cols].cpu().numpy().view(np.uint16).astype(np.float32)
        rc, ref = oracle.pipeline(host, int(Bd.U16), int(St.Clahe))
        assert rc == 0
        assert np.array_equal(got[:, :cols].cpu().numpy().view(np.uint16), ref)


@pytest.mark.parametrize("item_rows", [16, 48, 200])
def test_cf_kernel_short_items(item_rows, monkeypatch):
    rows, cols = 2500, 2300
    monkeypatch.setenv("SARPRO_HIP_U16_ITEM_ROWS", str(item_rows))  # read when the context is created; the plan is built per context
    with S.Context(0) as c:
        assert c.get_attr("U16_ITEM_ROWS") == item_rows
        band, pitch = _scene(c, 4242 + item_rows, rows, cols)
        got = _run(c, band, rows, cols, pitch)
        c.set_attr("NO_U16_CF", 1)
        old = _run(c, band, rows, cols, pitch)
        assert torch.equal(got[:, :cols], old[:, :cols])


def test_cf_kernel_wide_window_takes_the_global_table():
    """A window of more than 32768 DNs does not fit the kernel's LDS byte table: the bins are gathered from the global table."""
    rows, cols = 1100, 1300
    with S.Context(0) as c:
        pitch = (cols + 63) // 64 * 64
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        dn = torch.randint(1, 65536, (rows, pitch), generator=g, device="cuda", dtype=torch.int32)
        dn[::7, ::5] = 0
        band = (dn - 65536 * (dn >= 32768).to(torch.int32)).to(torch.int16)
        got = _run(c, band, rows, cols, pitch)
        c.set_attr("NO_U16_CF", 1)
        old = _run(c, band, rows, cols, pitch)
        assert torch.equal(got[:, :cols], old[:, :cols])
        host = band[:, :cols].cpu().numpy().view(np.uint16).astype(np.float32)
        rc, ref = oracle.pipeline(host, int(Bd.U16), int(St.Clahe))
        assert rc == 0 and np.array_equal(got[:, :cols].cpu().numpy().view(np.uint16), ref)


def test_cf_kernel_band_by_band():
    rows, cols = 2400, 2100
    with S.Context(0) as c:
        pitch = (cols + 63) // 64 * 64
        q = synth.q_tables()
        bands = [torch.zeros((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for b in range(2):
            c.dev_synth_scene_u16(777, b, q, rows, cols, 0, rows, bands[b].data_ptr(), pitch)
        outs = [_run(c, bands[b], rows, cols, pitch) for b in range(2)]
        for b in range(2):
            host = bands[b][:, :cols].cpu().numpy().view(np.uint16).astype(np.float32)
            rc, ref = oracle.pipeline(host, int(Bd.U16), int(St.Clahe))
            assert rc == 0 and np.array_equal(outs[b][:, :cols].cpu().numpy().view(np.uint16), ref)


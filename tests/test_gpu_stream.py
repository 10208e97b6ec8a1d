"""Streaming ingest / egress (sarpro_hip_dualpol_synrgb_stream_u16): row-chunk reader -> pinned ring -> side-stream
H2D with the DN histogram running behind each chunk -> device chain -> RGB back through a row sink.  Checked against
the oracle at sizes and chunkings that exercise the ring (more chunks than slots, ragged last chunk) and through the
strip-TIFF shims end to end."""
import numpy as np
import pytest

import sarpro_amd as S
from sarpro_amd import _lib, synth
from sarpro_amd.types import AutoscaleStrategy as St, SyntheticRgbMode as Mode

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with S.Context(0, timing=True) as c:
        yield c


def _scene(rows, cols):
    return [synth.scene_u16(rows, cols, k) for k in (0, 1)]


@pytest.mark.parametrize("strategy", [St.Clahe, St.Robust, St.Tamed, St.Standard])
@pytest.mark.parametrize("shape,chunk", [((300, 392), 37), ((513, 640), 64), ((129, 1000), 0), ((64, 200), 1000)])
def test_stream_matches_oracle(ctx, strategy, shape, chunk):
    rows, cols = shape
    b = _scene(rows, cols)
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    assert rc == 0
    got = np.zeros((rows, cols, 3), np.uint8)
    calls = {"read": 0, "sink": 0}

    def reader(band, row0, nrows, dst):
        calls["read"] += 1
        dst[:] = b[band][row0:row0 + nrows]

    def sink(row0, nrows, rgb):
        calls["sink"] += 1
        got[row0:row0 + nrows] = rgb

    st = ctx.dualpol_synrgb_stream(reader, rows, cols, strategy, Mode.Default, sink, chunk_rows=chunk, want_stats=True)
    assert np.array_equal(got, rrgb), (strategy, shape, chunk)
    assert st[0].valid_count == int((b[0] > 0).sum())
    if chunk and chunk < rows:
        assert calls["read"] == 2 * -(-rows // chunk)
    names = [n for n, _ in ctx.last_kernel_times()]
    assert "host:reader" in names and "host:sink" in names


def test_reader_and_sink_errors_abort_with_err_io(ctx):
    rows, cols = 128, 256
    b = _scene(rows, cols)

    def bad_reader(band, row0, nrows, dst):
        return 7 if row0 >= 64 else 0

    with pytest.raises(S.SarproHipError) as ei:
        ctx.dualpol_synrgb_stream(bad_reader, rows, cols, St.Robust, Mode.Default, lambda *a: 0, chunk_rows=32)
    assert ei.value.code == _lib.ERR_IO and "7" in str(ei.value)

    def reader(band, row0, nrows, dst):
        dst[:] = b[band][row0:row0 + nrows]

    with pytest.raises(S.SarproHipError) as ei:
        ctx.dualpol_synrgb_stream(reader, rows, cols, St.Robust, Mode.Default, lambda *a: 3, chunk_rows=32)
    assert ei.value.code == _lib.ERR_IO
    # the context is still usable afterwards
    got = np.zeros((rows, cols, 3), np.uint8)
    ctx.dualpol_synrgb_stream(reader, rows, cols, St.Robust, Mode.Default, lambda r0, n, rgb: got.__setitem__(slice(r0, r0 + n), rgb))
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(St.Robust))
    assert np.array_equal(got, rrgb)


@pytest.mark.parametrize("strategy", [St.Clahe, St.Default])
def test_tiff_files_in_tiff_file_out(ctx, tmp_path, strategy):
    """Two GRD-like band files -> RGB TIFF with the C callbacks of tiff_io.cpp on both ends (no Python in the loop)."""
    rows, cols = 700, 520
    b = _scene(rows, cols)
    paths = [str(tmp_path / f"band{k}.tif") for k in (0, 1)]
    for p, a in zip(paths, b):
        w = S.TiffWriter(p, cols, rows, 1, 16)
        w.write_rows(0, a)
        w.finish()
    ra, rb = S.TiffReader(paths[0]), S.TiffReader(paths[1])
    out = str(tmp_path / "rgb.tif")
    w = S.TiffWriter(out, cols, rows, 3, 8, geotransform=[10.0, 2.0, 0.0, 90.0, 0.0, -2.0])
    pair = S.TiffPair(ra, rb)
    ctx.dualpol_synrgb_stream(pair.reader(), rows, cols, strategy, Mode.Default, w.sink(), chunk_rows=96)
    w.finish()
    r = S.TiffReader(out)
    assert (r.info.width, r.info.height, r.info.samples_per_pixel, r.info.bits_per_sample) == (cols, rows, 3, 8)
    got = np.stack([r.read_rows(0, rows, sample=s) for s in range(3)], axis=-1).astype(np.uint8)
    rc, rrgb, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
    assert rc == 0 and np.array_equal(got, rrgb)


@pytest.mark.parametrize("strategy", [St.Robust, St.Clahe])
@pytest.mark.parametrize("target,pad", [(128, True), (96, False), (None, True)])
def test_resized_product_from_a_row_reader_equals_the_host_pointer_entry(ctx, strategy, target, pad):
    """sarpro_hip_dualpol_synrgb_resized_stream_u16 == sarpro_hip_dualpol_synrgb_resized_u16 (itself checked against
    the oracle in test_gpu_resize.py): only the way the bands reach the device differs."""
    rows, cols = 420, 610
    b = _scene(rows, cols)
    want, m = ctx.dualpol_synrgb_resized(b[0], b[1], strategy, target, pad)

    def reader(band, row0, nrows, dst):
        dst[:] = b[band][row0:row0 + nrows]

    got, meta = ctx.dualpol_synrgb_resized_stream(reader, rows, cols, strategy, target, pad)
    assert np.array_equal(got, want)
    assert (meta["final_cols"], meta["final_rows"], meta["pad_left"], meta["pad_top"]) == (m.final_cols, m.final_rows, m.pad_left, m.pad_top)


def test_batch_of_scenes_streamed_from_tiff_files(tmp_path):
    """Config 5's shape: a batch whose scenes are opened from files by the worker that picks them up (C reader
    callbacks, no arrays in host memory) equals the in-memory batch; a missing band file is a per-scene error."""
    shapes = [(300, 420), (257, 333), (410, 512)]
    scenes_mem, scenes_file, keep = [], [], []
    for i, (rows, cols) in enumerate(shapes):
        b = [synth.scene_u16(rows, cols, k, seed=synth.SEED_SCENE_A + i) for k in (0, 1)]
        scenes_mem.append((b[0], b[1]))
        readers = []
        for k in (0, 1):
            p = str(tmp_path / f"s{i}_b{k}.tif")
            w = S.TiffWriter(p, cols, rows, 1, 16)
            w.write_rows(0, b[k])
            w.finish()
            readers.append(S.TiffReader(p))
        pair = S.TiffPair(*readers)
        keep.append(pair)
        scenes_file.append((pair.reader(), rows, cols))
    want, rep_m, st_m, rc_m = S.batch_dualpol_synrgb_resized([0], scenes_mem, St.Robust, 128, True)
    got, rep_f, st_f, rc_f = S.batch_dualpol_synrgb_resized([0], scenes_file, St.Robust, 128, True)
    assert rc_m == 0 and rc_f == 0 and rep_f.processed == 3 and rep_f.errors == 0
    for a, b in zip(want, got):
        assert np.array_equal(a, b)

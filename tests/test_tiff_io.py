"""Host ingest / egress shims: the uncompressed strip TIFF / BigTIFF reader and sink of
csrc/tiff_io.cpp and the geotransform bookkeeping of save.rs:71-81.  No GPU."""
import struct

import numpy as np
import pytest

import sarpro_amd as S
from sarpro_amd import _lib


def _hand_tiff(path, arr, big_endian=False, bigtiff=False, rows_per_strip=1, planar=False, compression=1, tiled=False):
    """A TIFF written by hand (not by the library under test): arr is (rows, cols) or (rows, cols, samples) u8/u16."""
    e = ">" if big_endian else "<"
    a = arr if arr.ndim == 3 else arr[:, :, None]
    rows, cols, spp = a.shape
    bits = a.dtype.itemsize * 8
    data = a.astype(a.dtype.newbyteorder(e))
    strips = []
    planes = [data[:, :, s] for s in range(spp)] if planar else [data]
    for p in planes:
        for r in range(0, rows, rows_per_strip):
            strips.append(np.ascontiguousarray(p[r:r + rows_per_strip]).tobytes())
    hdr = 16 if bigtiff else 8
    offs, pos = [], hdr
    for s in strips:
        offs.append(pos)
        pos += len(s)
    pos += pos & 1
    ifd_off = pos
    T = {"H": 3, "I": 4, "Q": 16}
    ents = [(256, "I", [cols]), (257, "I", [rows]), (258, "H", [bits] * spp), (259, "H", [compression]),
            (262, "H", [2 if spp >= 3 else 1]), (273, "Q" if bigtiff else "I", offs), (277, "H", [spp]),
            (278, "I", [rows_per_strip]), (279, "Q" if bigtiff else "I", [len(s) for s in strips]),
            (284, "H", [2 if planar else 1]), (339, "H", [1] * spp)]
    if tiled:
        ents.append((322, "H", [16]))
    ents.sort()
    esz, cap = (20, 8) if bigtiff else (12, 4)
    ifd_len = (8 if bigtiff else 2) + len(ents) * esz + (8 if bigtiff else 4)
    extra_off, extra, body = ifd_off + ifd_len, b"", b""
    for tag, code, vals in ents:
        raw = struct.pack(e + code * len(vals), *vals)
        if len(raw) <= cap:
            val = raw.ljust(cap, b"\0")
        else:
            val = struct.pack(e + ("Q" if bigtiff else "I"), extra_off + len(extra))
            extra += raw + (b"\0" if len(raw) & 1 else b"")
        body += struct.pack(e + "HH" + ("Q" if bigtiff else "I"), tag, T[code], len(vals)) + val
    with open(path, "wb") as f:
        f.write((b"MM" if big_endian else b"II") + (struct.pack(e + "HHHQ", 43, 8, 0, ifd_off) if bigtiff else struct.pack(e + "HI", 42, ifd_off)))
        for s in strips:
            f.write(s)
        f.write(b"\0" * (ifd_off - f.tell()))
        f.write(struct.pack(e + ("Q" if bigtiff else "H"), len(ents)) + body + struct.pack(e + ("Q" if bigtiff else "I"), 0) + extra)


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("bigtiff", [False, True])
@pytest.mark.parametrize("rows_per_strip", [1, 7, 1000])
def test_reads_hand_written_grd_like_tiffs(tmp_path, big_endian, bigtiff, rows_per_strip):
    rng = np.random.default_rng(3)
    a = rng.integers(0, 65536, size=(53, 97), dtype=np.uint16)  # S1 GRD measurement rasters: u16, one strip per row
    p = str(tmp_path / "a.tif")
    _hand_tiff(p, a, big_endian, bigtiff, rows_per_strip)
    r = S.TiffReader(p)
    assert (r.info.width, r.info.height, r.info.bits_per_sample, r.info.samples_per_pixel) == (97, 53, 16, 1)
    assert bool(r.info.big_endian) == big_endian and bool(r.info.bigtiff) == bigtiff
    assert np.array_equal(r.read_rows(0, 53), a)
    assert np.array_equal(r.read_rows(20, 11), a[20:31])
    r.close()


@pytest.mark.parametrize("planar", [False, True])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_reads_multi_sample_chunky_and_planar(tmp_path, planar, dtype):
    rng = np.random.default_rng(4)
    a = rng.integers(0, np.iinfo(dtype).max + 1, size=(31, 40, 3)).astype(dtype)
    p = str(tmp_path / "m.tif")
    _hand_tiff(p, a, rows_per_strip=5, planar=planar)
    r = S.TiffReader(p)
    for s in range(3):
        assert np.array_equal(r.read_rows(3, 20, sample=s), a[3:23, :, s].astype(np.uint16))


def test_unsupported_layouts_fail_loudly(tmp_path):
    a = np.zeros((8, 8), np.uint16)
    for kw in (dict(compression=5), dict(tiled=True)):
        p = str(tmp_path / "x.tif")
        _hand_tiff(p, a, **kw)
        with pytest.raises(S.SarproHipError) as ei:
            S.TiffReader(p)
        assert ei.value.code == _lib.ERR_IO and "not supported" in str(ei.value)
    with pytest.raises(S.SarproHipError):
        S.TiffReader(str(tmp_path / "missing.tif"))
    (tmp_path / "junk.tif").write_bytes(b"not a tiff at all")
    with pytest.raises(S.SarproHipError):
        S.TiffReader(str(tmp_path / "junk.tif"))


@pytest.mark.parametrize("samples,bits", [(1, 8), (1, 16), (3, 8), (2, 16)])
def test_writer_round_trip_in_chunks(tmp_path, samples, bits):
    rng = np.random.default_rng(5)
    dt = np.uint8 if bits == 8 else np.uint16
    a = rng.integers(0, np.iinfo(dt).max + 1, size=(77, 130, samples)).astype(dt)
    p = str(tmp_path / "w.tif")
    w = S.TiffWriter(p, 130, 77, samples, bits)
    for r0 in (40, 0, 60):                       # out of order: rows land at their place
        r1 = {40: 60, 0: 40, 60: 77}[r0]
        w.write_rows(r0, a[r0:r1])
    w.finish()
    r = S.TiffReader(p)
    assert (r.info.width, r.info.height, r.info.samples_per_pixel, r.info.bits_per_sample) == (130, 77, samples, bits)
    for s in range(samples):
        assert np.array_equal(r.read_rows(0, 77, sample=s), a[:, :, s].astype(np.uint16))
    # the file also parses as a plain TIFF 6.0 directory (independent check of the header / IFD)
    raw = open(p, "rb").read()
    assert raw[:4] == b"II*\0"
    ifd = struct.unpack("<I", raw[4:8])[0]
    n = struct.unpack("<H", raw[ifd:ifd + 2])[0]
    tags = [struct.unpack("<H", raw[ifd + 2 + 12 * i: ifd + 4 + 12 * i])[0] for i in range(n)]
    assert tags == sorted(tags) and {256, 257, 258, 259, 262, 273, 277, 278, 279}.issubset(tags)


def test_geotiff_tags_round_trip_and_key_directory_is_carried_over(tmp_path):
    a = np.zeros((10, 12), np.uint16)
    gt = [399960.0, 10.0, 0.0, 4800000.0, 0.0, -10.0]
    p1, p2 = str(tmp_path / "g1.tif"), str(tmp_path / "g2.tif")
    w = S.TiffWriter(p1, 12, 10, 1, 16, geotransform=gt)
    w.write_rows(0, a)
    w.finish()
    r = S.TiffReader(p1)
    assert r.info.has_geo & 3 == 3
    assert list(r.info.pixel_scale) == [10.0, 10.0, 0.0] and list(r.info.tiepoint) == [0, 0, 0, 399960.0, 4800000.0, 0]
    w2 = S.TiffWriter(p2, 12, 10, 1, 16, geotransform=gt, geo_keys_from=r)  # no key directory in the source: nothing to copy
    w2.write_rows(0, a)
    w2.finish()
    assert S.TiffReader(p2).info.has_geo & 4 == 0


def test_update_geotransform_follows_save_rs():
    # save.rs:71-81: pixel size scaled by cols / final_cols (final = after padding), origin moved by the padding
    gt = [100.0, 10.0, 0.0, 5000.0, 0.0, -10.0]
    meta = dict(final_cols=250, final_rows=250, scale_x=0.5, scale_y=0.5, pad_left=0, pad_top=50)
    got = S.host_update_geotransform(gt, 500, 300, meta)
    px = 10.0 * (500 / 250)
    py = -10.0 * (300 / 250)
    assert got == [100.0 - 0 * px, px, 0.0, 5000.0 - 50 * py, 0.0, py]
    # scale 0 (no resize requested): pixel size untouched, padding still shifts the origin
    meta0 = dict(final_cols=500, final_rows=500, scale_x=0.0, scale_y=0.0, pad_left=0, pad_top=100)
    assert S.host_update_geotransform(gt, 500, 300, meta0) == [100.0, 10.0, 0.0, 5000.0 + 1000.0, 0.0, -10.0]

"""ctypes binding of the CPU oracle (oracle/sarpro_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Builds oracle/libsarpro_oracle.so on demand with gcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "libsarpro_oracle.so")

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED_SHAPE = 0, -1, -3


class Stats(C.Structure):
    _fields_ = [("valid_count", C.c_uint64)] + [
        (n, C.c_double)
        for n in ("min_db", "max_db", "mean_db", "std_db", "median_db", "p01", "p02", "p05", "p10",
                  "p25", "p75", "p90", "p95", "p98", "p99", "low_clip", "high_clip", "gamma",
                  "skew_factor", "tail_heaviness")
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "sarpro_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libsarpro_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.sarpro_oracle_version.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_lib_mt = None


def lib_mt():
    """Row-parallel variant of the headline path (oracle/sarpro_oracle_mt.c; OpenMP).  Not the reference's behaviour:
    bench.py reports it next to the single-thread figure as 'what the host's cores would give'."""
    global _lib_mt
    if _lib_mt is None:
        so = os.path.join(ORACLE_DIR, "libsarpro_oracle_mt.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("sarpro_oracle_mt.c", "sarpro_oracle.c")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libsarpro_oracle_mt.so"], stdout=subprocess.DEVNULL)
        _lib_mt = C.CDLL(so)
    return _lib_mt


def dualpol_clahe_synrgb_mt(b1: np.ndarray, b2: np.ndarray):
    b1 = np.ascontiguousarray(b1, np.float32)
    b2 = np.ascontiguousarray(b2, np.float32)
    rows, cols = b1.shape
    rgb = np.empty((rows, cols, 3), np.uint8)
    rc = lib_mt().sarpro_oracle_mt_dualpol_clahe_synrgb_f32(_p(b1), _p(b2), C.c_size_t(rows), C.c_size_t(cols), _p(rgb))
    return rc, rgb


def db_mask(x: np.ndarray):
    x = np.ascontiguousarray(x, np.float32)
    db = np.empty(x.shape, np.float64)
    mask = np.empty(x.shape, np.uint8)
    rc = lib().sarpro_oracle_db_mask_f32(_p(x), C.c_size_t(x.size), _p(db), _p(mask))
    assert rc == OK
    return db, mask


def stats(db: np.ndarray, mask: np.ndarray) -> Stats:
    s = Stats()
    rc = lib().sarpro_oracle_stats(_p(np.ascontiguousarray(db, np.float64)),
                                   _p(np.ascontiguousarray(mask, np.uint8)), C.c_size_t(db.size), C.byref(s))
    assert rc == OK
    return s


def pipeline(x: np.ndarray, bit_depth: int, strategy: int, want_stats: bool = False):
    """process_scalar_data_pipeline: returns (rc, raster[, stats]); raster is u8 or u16."""
    x = np.ascontiguousarray(x, np.float32)
    rows, cols = x.shape
    out8 = np.empty((rows, cols), np.uint8)
    out16 = np.empty((rows, cols), np.uint16)
    s = Stats()
    rc = lib().sarpro_oracle_pipeline_f32(_p(x), C.c_size_t(rows), C.c_size_t(cols), bit_depth, strategy,
                                          _p(out8), _p(out16), None, None, C.byref(s))
    out = out8 if bit_depth == 0 else out16
    return (rc, out, s) if want_stats else (rc, out)


def tamed_synrgb_u8(x: np.ndarray, is_copol: bool) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    db, mask = db_mask(x)
    out = np.empty(x.shape, np.uint8)
    rc = lib().sarpro_oracle_tamed_synrgb_u8(_p(db), _p(mask), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]),
                                             int(is_copol), _p(out))
    assert rc == OK
    return out


def polop(op: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty(a.shape, np.float32)
    rc = lib().sarpro_oracle_polop_f32(op, _p(a), _p(b), C.c_size_t(a.size), _p(out))
    assert rc == OK
    return out


def synrgb(mode: int, strategy: int, b1: np.ndarray, b2: np.ndarray) -> np.ndarray:
    b1 = np.ascontiguousarray(b1, np.uint8)
    b2 = np.ascontiguousarray(b2, np.uint8)
    rgb = np.empty(b1.shape + (3,), np.uint8)
    rc = lib().sarpro_oracle_synrgb(mode, strategy, _p(b1), _p(b2), C.c_size_t(b1.size), _p(rgb))
    assert rc == OK
    return rgb


def synrgb_luts(suppressed: bool, b1=None, b2=None):
    """(lut_r, lut_g, lut_b[256,256], floor) of the default or suppressed composition."""
    if b1 is None:
        b1 = np.zeros(1, np.uint8)
        b2 = np.zeros(1, np.uint8)
    b1 = np.ascontiguousarray(b1, np.uint8)
    b2 = np.ascontiguousarray(b2, np.uint8)
    rgb = np.empty(b1.size * 3, np.uint8)
    luts = np.empty(66048, np.uint8)
    fl = C.c_int(-1)
    if suppressed:
        rc = lib().sarpro_oracle_synrgb_suppressed(_p(b1), _p(b2), C.c_size_t(b1.size), _p(rgb), _p(luts), C.byref(fl))
    else:
        rc = lib().sarpro_oracle_synrgb_default(_p(b1), _p(b2), C.c_size_t(b1.size), _p(rgb), _p(luts))
    assert rc == OK
    return luts[:256].copy(), luts[256:512].copy(), luts[512:].reshape(256, 256).copy(), fl.value


def scale_u16_to_u8(v: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(v, np.uint16)
    out = np.empty(v.shape, np.uint8)
    rc = lib().sarpro_oracle_scale_u16_to_u8(_p(v), C.c_size_t(v.size), _p(out))
    assert rc == OK
    return out


def clahe(norm: np.ndarray, mask: np.ndarray, want_cdfs: bool = False):
    norm = np.ascontiguousarray(norm, np.float64)
    mask = np.ascontiguousarray(mask, np.uint8)
    rows, cols = norm.shape
    out = np.empty((rows, cols), np.float64)
    cdfs = np.empty((64, 256), np.float64)
    rc = lib().sarpro_oracle_clahe(_p(norm), _p(mask), C.c_size_t(rows), C.c_size_t(cols), C.c_size_t(8),
                                   C.c_size_t(8), C.c_double(2.0), C.c_size_t(256), _p(out), _p(cdfs))
    return (rc, out, cdfs) if want_cdfs else (rc, out)


def clahe_tile_cdf(hist: np.ndarray, tile_rows: int, tile_cols: int) -> np.ndarray:
    h = np.ascontiguousarray(hist, np.uint32).copy()
    cdf = np.empty(256, np.float64)
    lib().sarpro_oracle_clahe_tile_cdf(_p(h), C.c_size_t(256), C.c_size_t(tile_rows), C.c_size_t(tile_cols),
                                       C.c_double(2.0), _p(cdf))
    return cdf


def clahe_shape_ok(rows: int, cols: int) -> bool:
    return bool(lib().sarpro_oracle_clahe_shape_ok(C.c_size_t(rows), C.c_size_t(cols), C.c_size_t(8), C.c_size_t(8)))


def dualpol_synrgb(b1: np.ndarray, b2: np.ndarray, strategy: int, mode: int = 0):
    """save.rs:317-367 at native resolution: returns (rc, rgb[rows,cols,3], u8_band1, u8_band2)."""
    b1 = np.ascontiguousarray(b1, np.float32)
    b2 = np.ascontiguousarray(b2, np.float32)
    rows, cols = b1.shape
    rgb = np.empty((rows, cols, 3), np.uint8)
    u1 = np.empty((rows, cols), np.uint8)
    u2 = np.empty((rows, cols), np.uint8)
    rc = lib().sarpro_oracle_dualpol_synrgb_f32(_p(b1), _p(b2), C.c_size_t(rows), C.c_size_t(cols), strategy, mode,
                                                _p(rgb), _p(u1), _p(u2))
    return rc, rgb, u1, u2


def pad_to_square(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    rows, cols = a.shape
    m = max(rows, cols)
    out = np.empty((m, m), a.dtype)
    lib().sarpro_oracle_pad_to_square(_p(a), C.c_size_t(cols), C.c_size_t(rows), C.c_size_t(a.itemsize), _p(out))
    return out


def resize_dims(cols: int, rows: int, target: int):
    nc, nr = C.c_size_t(), C.c_size_t()
    lib().sarpro_oracle_resize_dims(C.c_size_t(cols), C.c_size_t(rows), C.c_size_t(target), C.byref(nc), C.byref(nr))
    return nc.value, nr.value


def resize_image_data_with_meta(a: np.ndarray, target_size, pad: bool):
    """resize.rs:91-236 -> (raster, meta dict)"""
    a = np.ascontiguousarray(a)
    rows, cols = a.shape
    nc, nr = (cols, rows)
    if target_size and max(cols, rows) != target_size:
        nc, nr = resize_dims(cols, rows, target_size)
    fc, fr = (max(nc, nr),) * 2 if pad else (nc, nr)
    out = np.empty((fr, fc), a.dtype)
    meta = np.zeros(6, np.float64)
    rc = lib().sarpro_oracle_resize_image_data_with_meta(_p(a), C.c_size_t(cols), C.c_size_t(rows), C.c_size_t(target_size or 0),
                                                         C.c_size_t(a.itemsize), int(pad), _p(out), _p(meta))
    assert rc == OK, rc
    assert (int(meta[0]), int(meta[1])) == (fc, fr)
    return out, dict(final_cols=int(meta[0]), final_rows=int(meta[1]), scale_x=meta[2], scale_y=meta[3],
                     pad_left=int(meta[4]), pad_top=int(meta[5]))

"""Host-side mirror of the reference's raster-core interface, over the C ABI.

Function names, argument meaning and error behaviour follow the Rust functions they
replace (file:line in the reference tree):

    process_scalar_data_pipeline               src/core/processing/pipeline.rs:42
    process_scalar_data_inplace                pipeline.rs:8
    autoscale_db_image_tamed_synrgb_u8         autoscale.rs:710
    sum_arrays / difference_arrays / ratio_arrays /
    normalized_diff_arrays / log_ratio_arrays  ops.rs:4-44
    create_synthetic_rgb_by_mode_and_strategy  synthetic_rgb.rs:182
    save_multiband JPEG branch (native res)    save.rs:317-367  -> dualpol_synrgb

numpy arrays in, numpy arrays out (host entry points).  `Context.dev_*` methods take device
pointers (e.g. torch.Tensor.data_ptr()) for rasters already resident in HBM.
All raster work runs in libsarpro_hip.so on the GPU; nothing here computes pixels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import F32Partial, Stats, lib
from .types import AutoscaleStrategy, BitDepth, PolarizationOperation, SyntheticRgbMode


class SarproHipError(RuntimeError):
    """Maps a non-zero C-ABI status (reference: Error::Processing / Error::External, src/error.rs:39-46)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"sarpro_hip status {code}: {msg}")
        self.code = code


def _vp(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


class Context:
    """One sarpro_hip_ctx: a device, a stream, a grow-only workspace.  One per host thread."""

    _live = None  # weakref.WeakSet of the open contexts (the test suite mirrors route-switch changes onto them: tests/conftest.py)

    def __init__(self, device: int = 0, timing: bool = False, async_dev: bool = False):
        h = C.c_void_p()
        # async_dev: SARPRO_HIP_CTX_ASYNC_DEV -- dev_dualpol_synrgb_u16 returns once enqueued; call synchronize()
        rc = lib.sarpro_hip_ctx_create(device, (1 if timing else 0) | (2 if async_dev else 0), C.byref(h))
        if rc != _lib.OK:
            raise SarproHipError(rc, (lib.sarpro_hip_last_error(None) or b"").decode())
        self._h = h
        self.device = device
        if Context._live is None:
            import weakref
            Context._live = weakref.WeakSet()
        Context._live.add(self)

    def close(self):
        if getattr(self, "_h", None):
            lib.sarpro_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc: int):
        if rc != _lib.OK:
            raise SarproHipError(rc, (lib.sarpro_hip_last_error(self._h) or b"").decode())

    @property
    def stream(self) -> int:
        return lib.sarpro_hip_ctx_stream(self._h) or 0

    def synchronize(self):
        self._chk(lib.sarpro_hip_ctx_synchronize(self._h))

    def last_kernel_times(self):
        names = (C.c_char_p * 1024)()
        ms = (C.c_float * 1024)()
        n = lib.sarpro_hip_last_kernel_times(self._h, names, ms, 1024)
        return [(names[i].decode(), float(ms[i])) for i in range(max(n, 0))]

    def spec_report(self) -> dict:
        """Diagnostics of the last speculative CLAHE chain (sarpro_hip_ctx_spec_report); synchronises the stream."""
        from ._lib import SpecReport
        r = SpecReport()
        self._chk(lib.sarpro_hip_ctx_spec_report(self._h, C.byref(r)))
        return {"spec_ok": int(r.spec_ok), "verdict": int(r.verdict), "floor_pred": int(r.floor_pred), "n_lt": [int(x) for x in r.n_lt],
                "target": int(r.target), "est_lt": [float(x) for x in r.est_lt], "sample_valid": [int(x) for x in r.sample_valid], "pool_overflow": int(r.pool_overflow),
                "n_below_min": int(r.n_below_min), "min_pred": [int(x) for x in r.min_pred], "retried": int(r.retried), "floor_first": int(r.floor_first),
                "outcome": ("unproven" if not r.spec_ok else "pool_overflow" if r.pool_overflow else "refuted" if r.verdict else "retried" if r.retried else "accepted")}

    def chain_report(self) -> dict:
        """What stood between the CLAHE levels and the composition in the last u16 chain (sarpro_hip_ctx_chain_report): the u8 rescale
        tables, the synRGB floor + cushion, the exact kernels' level histogram; synchronises the stream."""
        import numpy as np
        from ._lib import ChainReport
        r = ChainReport()
        self._chk(lib.sarpro_hip_ctx_chain_report(self._h, C.byref(r)))
        return {"floor_with_cushion": int(r.floor_with_cushion), "identity": [int(x) for x in r.identity],
                "rescale": np.ctypeslib.as_array(r.rescale).reshape(2, 256).copy(), "level_hist": np.ctypeslib.as_array(r.level_hist).reshape(2, 256).copy()}

    # ------------------------------------------------------------------ context attributes (route switches)
    def set_attr(self, name: str, value=1):
        """sarpro_hip_ctx_set_attr: `name` as in DESIGN.md's list of cross-check switches ("NO_SPEC", "SAMPLE_STRIDE", ...; the
        SARPRO_HIP_ prefix is optional).  value None resets the attribute to "unset"; the two word-valued attributes also take their
        words (SPEC_FORCE: "mispredict", "nospec", "lowmin", "noretry", "mispredict2", comma-joined; F32_ZONES: "tiny")."""
        if value is None:
            return self.reset_attr(name)
        if isinstance(value, str):
            words = {"tiny": 2} if name.upper().endswith("F32_ZONES") else {"mispredict": 1, "nospec": 2, "lowmin": 4, "noretry": 8, "mispredict2": 16}
            try:
                value = sum(words[w] for w in value.split(",")) if all(w in words for w in value.split(",")) else int(value)
            except ValueError:
                value = 1  # any other word switches the attribute on, as a bare environment variable does at creation
        self._chk(lib.sarpro_hip_ctx_set_attr(self._h, name.encode(), int(value)))

    def reset_attr(self, name: str):
        self._chk(lib.sarpro_hip_ctx_reset_attr(self._h, name.encode()))

    def get_attr(self, name: str):
        """-> the attribute's value, or None when it is unset"""
        v, st = C.c_int64(0), C.c_int(0)
        self._chk(lib.sarpro_hip_ctx_get_attr(self._h, name.encode(), C.byref(v), C.byref(st)))
        return int(v.value) if st.value else None

    @staticmethod
    def attr_names():
        out, i = [], 0
        while True:
            n = lib.sarpro_hip_attr_name(i)
            if not n:
                return out
            out.append(n.decode())
            i += 1

    def time_only(self, kernel_name=None):
        """Bracket only this kernel with events (None: every kernel); see sarpro_hip_ctx_time_only."""
        self._chk(lib.sarpro_hip_ctx_time_only(self._h, kernel_name.encode() if kernel_name else None))

    # ------------------------------------------------------------------ pipeline.rs:42
    def process_scalar_data_pipeline(self, processed: np.ndarray, bit_depth: BitDepth,
                                     strategy: AutoscaleStrategy, want_stats: bool = False):
        """Returns (scaled_u8, scaled_u16) like the reference's last two tuple members:
        U8 -> (u8[rows,cols], None); U16 -> (empty u8, u16[rows,cols]).  The dB buffer and
        mask the reference also returns are not materialised (see process_scalar_data_inplace)."""
        if processed.ndim != 2:
            raise ValueError("processed must be 2-D (rows, cols)")
        rows, cols = processed.shape
        st = Stats()
        out8 = np.empty((rows, cols), np.uint8) if bit_depth == BitDepth.U8 else None
        out16 = np.empty((rows, cols), np.uint16) if bit_depth == BitDepth.U16 else None
        if processed.dtype == np.uint16:
            x = np.ascontiguousarray(processed)
            fn = lib.sarpro_hip_autoscale_band_u16
        else:
            x = np.ascontiguousarray(processed, np.float32)
            fn = lib.sarpro_hip_autoscale_band_f32
        # stats_out = NULL lets the f32 flavour skip the dB moments and the eleven-percentile histogram (zone route)
        self._chk(fn(self._h, _vp(x), rows, cols, int(strategy), int(bit_depth), _vp(out8), _vp(out16), C.byref(st) if want_stats else None))
        res = (out8, None) if bit_depth == BitDepth.U8 else (np.empty(0, np.uint8), out16)
        return res + (st,) if want_stats else res

    def polop_autoscale_band(self, op, a: np.ndarray, b: np.ndarray, bit_depth: BitDepth, strategy: AutoscaleStrategy,
                             want_stats: bool = False):
        """ops.rs:4-44 followed by process_scalar_data_pipeline on the result (io/sentinel1.rs:1501-1578), fused: the f32 pol-op
        raster is never materialised.  a, b: both f32 or both u16 DN.  Returns like process_scalar_data_pipeline."""
        if a.shape != b.shape or a.ndim != 2:
            raise SarproHipError(_lib.ERR_SHAPE_MISMATCH, "band shapes differ")
        rows, cols = a.shape
        st = Stats()
        out8 = np.empty((rows, cols), np.uint8) if bit_depth == BitDepth.U8 else None
        out16 = np.empty((rows, cols), np.uint16) if bit_depth == BitDepth.U16 else None
        if a.dtype == np.uint16 and b.dtype == np.uint16:
            fn, dt = lib.sarpro_hip_polop_autoscale_band_u16, np.uint16
        else:
            fn, dt = lib.sarpro_hip_polop_autoscale_band_f32, np.float32
        x, y = np.ascontiguousarray(a, dt), np.ascontiguousarray(b, dt)
        self._chk(fn(self._h, int(op), _vp(x), _vp(y), rows, cols, int(strategy), int(bit_depth), _vp(out8), _vp(out16),
                     C.byref(st) if want_stats else None))
        res = (out8, None) if bit_depth == BitDepth.U8 else (np.empty(0, np.uint8), out16)
        return res + (st,) if want_stats else res

    def selftest_polop_division(self) -> int:
        """Pairs of u16 values (of all 2^32) for which the u16 pol-op kernels' division differs from the IEEE division: must be 0."""
        n = C.c_uint64()
        self._chk(lib.sarpro_hip_selftest_polop_division(self._h, C.byref(n)))
        return n.value

    def dev_polop_autoscale_band(self, op, d_a: int, d_b: int, u16_in: bool, rows: int, cols: int, in_pitch: int, strategy, bit_depth,
                                 d_out: int, out_pitch: int, want_stats: bool = True) -> Stats | None:
        st = Stats() if want_stats else None
        fn = lib.sarpro_hip_polop_autoscale_band_u16_dev if u16_in else lib.sarpro_hip_polop_autoscale_band_f32_dev
        self._chk(fn(self._h, int(op), _vp(d_a), _vp(d_b), rows, cols, in_pitch, int(strategy), int(bit_depth), _vp(d_out), out_pitch,
                     C.byref(st) if want_stats else None))
        return st

    # ------------------------------------------------------------------ pipeline.rs:8
    def process_scalar_data_inplace(self, processed: np.ndarray):
        x = np.ascontiguousarray(processed, np.float32)
        rows, cols = x.shape
        db = np.empty((rows, cols), np.float64)
        mask = np.empty((rows, cols), np.uint8)
        self._chk(lib.sarpro_hip_db_mask_f32(self._h, _vp(x), rows, cols, _vp(db), _vp(mask)))
        return db, mask.astype(bool)

    # ------------------------------------------------------------------ autoscale.rs:710
    def autoscale_db_image_tamed_synrgb_u8(self, band: np.ndarray, is_copol: bool) -> np.ndarray:
        """Takes the band (u16 or f32), not the dB buffer: dB is recomputed on the device."""
        rows, cols = band.shape
        out = np.empty((rows, cols), np.uint8)
        if band.dtype == np.uint16:
            self._chk(lib.sarpro_hip_tamed_synrgb_u8_u16(self._h, _vp(np.ascontiguousarray(band)), rows, cols, int(is_copol), _vp(out)))
        else:
            x = np.ascontiguousarray(band, np.float32)
            self._chk(lib.sarpro_hip_tamed_synrgb_u8_f32(self._h, _vp(x), rows, cols, int(is_copol), _vp(out)))
        return out

    # ------------------------------------------------------------------ ops.rs
    def _polop(self, op: PolarizationOperation, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        if a.shape != b.shape:
            raise SarproHipError(_lib.ERR_SHAPE_MISMATCH, "operand shapes differ")
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.empty(a.shape, np.float32)
        self._chk(lib.sarpro_hip_polop_f32(self._h, int(op), _vp(a), _vp(b), a.size, _vp(out)))
        return out

    def sum_arrays(self, a, b):
        return self._polop(PolarizationOperation.Sum, a, b)

    def difference_arrays(self, a, b):
        return self._polop(PolarizationOperation.Diff, a, b)

    def ratio_arrays(self, a, b):
        return self._polop(PolarizationOperation.Ratio, a, b)

    def normalized_diff_arrays(self, a, b):
        return self._polop(PolarizationOperation.NDiff, a, b)

    def log_ratio_arrays(self, a, b):
        return self._polop(PolarizationOperation.LogRatio, a, b)

    # ------------------------------------------------------------------ synthetic_rgb.rs:182
    def create_synthetic_rgb_by_mode_and_strategy(self, mode: SyntheticRgbMode, strategy: AutoscaleStrategy,
                                                  band1_data: np.ndarray, band2_data: np.ndarray) -> np.ndarray:
        if band1_data.size != band2_data.size:
            raise SarproHipError(_lib.ERR_SHAPE_MISMATCH, "band lengths differ")  # debug_assert_eq!, synthetic_rgb.rs:11,89
        b1 = np.ascontiguousarray(band1_data, np.uint8)
        b2 = np.ascontiguousarray(band2_data, np.uint8)
        rgb = np.empty(b1.shape + (3,), np.uint8)
        self._chk(lib.sarpro_hip_synrgb_u8(self._h, int(mode), int(strategy), _vp(b1), _vp(b2), b1.size, _vp(rgb)))
        return rgb

    # ------------------------------------------------------------------ save.rs:317-367
    def dualpol_synrgb(self, band1: np.ndarray, band2: np.ndarray, strategy: AutoscaleStrategy,
                       mode: SyntheticRgbMode = SyntheticRgbMode.Default, want_u8: bool = False,
                       want_stats: bool = False):
        if band1.shape != band2.shape or band1.ndim != 2:
            raise SarproHipError(_lib.ERR_SHAPE_MISMATCH, "band shapes differ")
        rows, cols = band1.shape
        rgb = np.empty((rows, cols, 3), np.uint8)
        u1 = np.empty((rows, cols), np.uint8) if want_u8 else None
        u2 = np.empty((rows, cols), np.uint8) if want_u8 else None
        st = (Stats * 2)()
        if band1.dtype == np.uint16 and band2.dtype == np.uint16:
            fn, dt = lib.sarpro_hip_dualpol_synrgb_u16, np.uint16
        else:
            fn, dt = lib.sarpro_hip_dualpol_synrgb_f32, np.float32
        b1 = np.ascontiguousarray(band1, dt)
        b2 = np.ascontiguousarray(band2, dt)
        self._chk(fn(self._h, _vp(b1), _vp(b2), rows, cols, int(strategy), int(mode), _vp(rgb), _vp(u1), _vp(u2), st if want_stats else None))
        out = (rgb,)
        if want_u8:
            out += (u1, u2)
        if want_stats:
            out += ([st[0], st[1]],)
        return out if len(out) > 1 else rgb

    # ------------------------------------------------------------------ resize.rs:91 / save.rs:317-367
    def resize_image_data_with_meta(self, data: np.ndarray, target_size: int | None, pad: bool):
        """Returns (final raster, ResizeMeta): Lanczos3 to `target_size` on the long side, optional square pad."""
        from ._lib import ResizeMeta
        if data.dtype not in (np.uint8, np.uint16) or data.ndim != 2:
            raise ValueError("data must be a 2-D uint8 or uint16 raster")
        rows, cols = data.shape
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        out = np.empty((fr, fc), data.dtype)
        m = ResizeMeta()
        x = np.ascontiguousarray(data)
        self._chk(lib.sarpro_hip_resize_image_data(self._h, _vp(x), cols, rows, target_size or 0,
                                                   0 if data.dtype == np.uint8 else 1, int(pad), _vp(out), C.byref(m)))
        return out, m

    def dualpol_synrgb_resized(self, band1: np.ndarray, band2: np.ndarray, strategy: AutoscaleStrategy,
                               target_size: int | None, pad: bool, mode: SyntheticRgbMode = SyntheticRgbMode.Default):
        """save.rs:317-367 with its resize / pad steps: u16 bands in, (final_rows, final_cols, 3) RGB out."""
        from ._lib import ResizeMeta
        rows, cols = band1.shape
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        rgb = np.empty((fr, fc, 3), np.uint8)
        m = ResizeMeta()
        b1 = np.ascontiguousarray(band1, np.uint16)
        b2 = np.ascontiguousarray(band2, np.uint16)
        self._chk(lib.sarpro_hip_dualpol_synrgb_resized_u16(self._h, _vp(b1), _vp(b2), rows, cols, int(strategy), int(mode),
                                                            target_size or 0, int(pad), _vp(rgb), C.byref(m)))
        return rgb, m

    def dev_dualpol_synrgb_resized(self, d_b1: int, d_b2: int, rows: int, cols: int, in_pitch: int, strategy: AutoscaleStrategy,
                                   target_size: int | None, pad: bool, d_rgb: int, mode: SyntheticRgbMode = SyntheticRgbMode.Default):
        """dualpol_synrgb_resized with the bands and the compact final_rows x final_cols x 3 RGB raster in device memory."""
        from ._lib import ResizeMeta
        m = ResizeMeta()
        self._chk(lib.sarpro_hip_dualpol_synrgb_resized_u16_dev(self._h, _vp(d_b1), _vp(d_b2), rows, cols, in_pitch, int(strategy), int(mode),
                                                                target_size or 0, int(pad), _vp(d_rgb), C.byref(m)))
        return m

    def dualpol_synrgb_resized_f32(self, band1: np.ndarray, band2: np.ndarray, strategy: AutoscaleStrategy, target_size: int | None, pad: bool,
                                   mode: SyntheticRgbMode = SyntheticRgbMode.Default, plain_pipeline: bool = False):
        """f32 bands (resampled on read, the reference's default flow) -> per-band u8 -> resize -> pad -> synRGB.
        plain_pipeline: api/mod.rs:404-437 (no Tamed re-autoscale); default: save.rs:317-367."""
        from ._lib import ResizeMeta
        rows, cols = band1.shape
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        rgb = np.empty((fr, fc, 3), np.uint8)
        m = ResizeMeta()
        b1 = np.ascontiguousarray(band1, np.float32)
        b2 = np.ascontiguousarray(band2, np.float32)
        self._chk(lib.sarpro_hip_dualpol_synrgb_resized_f32(self._h, _vp(b1), _vp(b2), rows, cols, int(strategy), int(mode), 1 if plain_pipeline else 0,
                                                            target_size or 0, int(pad), _vp(rgb), C.byref(m)))
        return rgb, m

    def dev_dualpol_synrgb_f32(self, d_b1: int, d_b2: int, rows: int, cols: int, in_pitch: int, strategy: AutoscaleStrategy, mode: SyntheticRgbMode,
                               d_rgb: int, rgb_pitch_px: int, d_u8_1: int = 0, d_u8_2: int = 0, u8_pitch: int = 0, plain_pipeline: bool = False,
                               want_stats: bool = False):
        st = (Stats * 2)()
        self._chk(lib.sarpro_hip_dualpol_synrgb_f32_dev(self._h, _vp(d_b1), _vp(d_b2), rows, cols, in_pitch, int(strategy), int(mode),
                                                        1 if plain_pipeline else 0, _vp(d_rgb), rgb_pitch_px, _vp(d_u8_1) if d_u8_1 else None,
                                                        _vp(d_u8_2) if d_u8_2 else None, u8_pitch, st if want_stats else None))
        return [st[0], st[1]] if want_stats else None

    def dev_dualpol_synrgb_resized_f32(self, d_b1: int, d_b2: int, rows: int, cols: int, in_pitch: int, strategy: AutoscaleStrategy,
                                       target_size: int | None, pad: bool, d_rgb: int, mode: SyntheticRgbMode = SyntheticRgbMode.Default,
                                       plain_pipeline: bool = False):
        from ._lib import ResizeMeta
        m = ResizeMeta()
        self._chk(lib.sarpro_hip_dualpol_synrgb_resized_f32_dev(self._h, _vp(d_b1), _vp(d_b2), rows, cols, in_pitch, int(strategy), int(mode),
                                                                1 if plain_pipeline else 0, target_size or 0, int(pad), _vp(d_rgb), C.byref(m)))
        return m

    def save_processed_image_raster(self, processed: np.ndarray, bit_depth: BitDepth, strategy: AutoscaleStrategy,
                                    target_size: int | None, pad: bool):
        """save_processed_image (save.rs:23-170) up to the raster its writer receives -> (raster, ResizeMeta)."""
        from ._lib import ResizeMeta
        rows, cols = processed.shape
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        out = np.empty((fr, fc), np.uint8 if bit_depth == BitDepth.U8 else np.uint16)
        m = ResizeMeta()
        if processed.dtype == np.uint16:
            x, fn = np.ascontiguousarray(processed), lib.sarpro_hip_process_band_resized_u16
        else:
            x, fn = np.ascontiguousarray(processed, np.float32), lib.sarpro_hip_process_band_resized_f32
        self._chk(fn(self._h, _vp(x), rows, cols, int(strategy), int(bit_depth), target_size or 0, int(pad), _vp(out), C.byref(m)))
        return out, m

    # ------------------------------------------------------------------ device-pointer variants
    def dev_autoscale_band_u16(self, d_in: int, rows: int, cols: int, in_pitch: int, strategy, bit_depth,
                               d_out: int, out_pitch: int) -> Stats:
        st = Stats()
        self._chk(lib.sarpro_hip_autoscale_band_u16_dev(self._h, _vp(d_in), rows, cols, in_pitch, int(strategy),
                                                        int(bit_depth), _vp(d_out), out_pitch, C.byref(st)))
        return st

    def dev_autoscale_band_f32(self, d_in: int, rows: int, cols: int, in_pitch: int, strategy, bit_depth,
                               d_out: int, out_pitch: int, want_stats: bool = True) -> Stats | None:
        """want_stats=False passes stats_out = NULL: the pre-pass then skips the per-sample dB moments."""
        st = Stats() if want_stats else None
        self._chk(lib.sarpro_hip_autoscale_band_f32_dev(self._h, _vp(d_in), rows, cols, in_pitch, int(strategy),
                                                        int(bit_depth), _vp(d_out), out_pitch, C.byref(st) if want_stats else None))
        return st

    def dev_dualpol_synrgb_u16(self, d_b1: int, d_b2: int, rows: int, cols: int, in_pitch: int, strategy, mode,
                               d_rgb: int, rgb_pitch_px: int, d_u8_1: int | None = None, d_u8_2: int | None = None,
                               u8_pitch: int = 0, want_stats: bool = True):
        """want_stats=False passes stats_out = NULL: on a Context(async_dev=True) the call then returns once the
        device chain is enqueued (synchronize() before reading the rasters)."""
        st = (Stats * 2)() if want_stats else None
        self._chk(lib.sarpro_hip_dualpol_synrgb_u16_dev(self._h, _vp(d_b1), _vp(d_b2), rows, cols, in_pitch,
                                                        int(strategy), int(mode), _vp(d_rgb), rgb_pitch_px,
                                                        _vp(d_u8_1), _vp(d_u8_2), u8_pitch, st))
        return [st[0], st[1]] if want_stats else None

    ROUTES = {-1: "n/a", 0: "accepted", 1: "refuted", 2: "unproven", 3: "pool_overflow", 4: "retried"}

    def dev_batch_dualpol_synrgb_u16(self, scenes, rows: int, cols: int, in_pitch: int, strategy, mode, rgb_pitch_px: int,
                                     lanes: int = 0, continue_on_error: bool = True, check: bool = True):
        """sarpro_hip_batch_dualpol_synrgb_u16_dev: `scenes` = [(d_band1, d_band2, d_rgb), ...] device pointers of one shape; the
        batch loop of api/mod.rs:484-533 for resident scenes, pipelined over `lanes` internal lanes.  Returns (report dict,
        per-scene statuses, per-scene routes); raises on a failed batch unless check=False."""
        from ._lib import BatchReport, ResidentScene
        arr = (ResidentScene * max(len(scenes), 1))()
        for i, (b1, b2, rgb) in enumerate(scenes):
            arr[i].d_band1, arr[i].d_band2, arr[i].d_rgb = b1, b2, rgb
        rep = BatchReport()
        rc = lib.sarpro_hip_batch_dualpol_synrgb_u16_dev(self._h, arr, len(scenes), rows, cols, in_pitch, int(strategy), int(mode), rgb_pitch_px,
                                                         int(lanes), 1 if continue_on_error else 0, C.byref(rep))
        if check:
            self._chk(rc)
        n = len(scenes)
        return ({"processed": int(rep.processed), "skipped": int(rep.skipped), "errors": int(rep.errors), "rc": int(rc)},
                [int(arr[i].status) for i in range(n)], [self.ROUTES.get(int(arr[i].route), "n/a") for i in range(n)])

    def dev_batch_dualpol_synrgb_resized_f32(self, scenes, rows: int, cols: int, in_pitch: int, strategy, target_size, pad: bool,
                                             mode=SyntheticRgbMode.Default, plain_pipeline: bool = False, lanes: int = 0, continue_on_error: bool = True,
                                             check: bool = True):
        """sarpro_hip_batch_dualpol_synrgb_resized_f32_dev: `scenes` = [(d_band1, d_band2, d_rgb), ...] device pointers of one shape (f32 bands,
        compact RGB out); the reference's default flow over the scenes of a directory, on `lanes` internal lanes with a host thread each.
        Returns (report dict, per-scene statuses)."""
        from ._lib import BatchReport, ResidentSceneF32
        arr = (ResidentSceneF32 * max(len(scenes), 1))()
        for i, (b1, b2, rgb) in enumerate(scenes):
            arr[i].d_band1, arr[i].d_band2, arr[i].d_rgb = b1, b2, rgb
        rep = BatchReport()
        rc = lib.sarpro_hip_batch_dualpol_synrgb_resized_f32_dev(self._h, arr, len(scenes), rows, cols, in_pitch, int(strategy), int(mode),
                                                                 1 if plain_pipeline else 0, target_size or 0, int(pad), int(lanes), 1 if continue_on_error else 0, C.byref(rep))
        if check:
            self._chk(rc)
        return ({"processed": int(rep.processed), "skipped": int(rep.skipped), "errors": int(rep.errors), "rc": int(rc)}, [int(arr[i].status) for i in range(len(scenes))])

    def dev_polop_f32(self, op, d_a: int, d_b: int, n: int, d_out: int):
        self._chk(lib.sarpro_hip_polop_f32_dev(self._h, int(op), _vp(d_a), _vp(d_b), n, _vp(d_out)))

    def dev_synrgb_u8(self, mode, strategy, d_b1: int, d_b2: int, n: int, d_rgb: int):
        self._chk(lib.sarpro_hip_synrgb_u8_dev(self._h, int(mode), int(strategy), _vp(d_b1), _vp(d_b2), n, _vp(d_rgb)))

    def dev_synth_scene_u16(self, seed: int, band: int, q_tables: np.ndarray, rows_total: int, cols: int,
                            row0: int, rows_local: int, d_out: int, pitch: int, flags: int = 0):
        """flags: synth.NO_WEDGE | synth.NO_BRIGHT | synth.class_map(m) | synth.blocks(n) (sarpro_hip_synth_scene_u16_dev_ex)"""
        q = np.ascontiguousarray(q_tables, np.uint16)
        assert q.shape == (2, 4, 65536)
        self._chk(lib.sarpro_hip_synth_scene_u16_dev_ex(self._h, seed, band, _vp(q), rows_total, cols, row0, rows_local,
                                                        _vp(d_out), pitch, flags))

    # ------------------------------------------------------------------ row-stripe protocol
    def stripe_begin_u16(self, d_b1: int, d_b2: int, rows_total: int, cols: int, row0: int, rows_local: int,
                         in_pitch: int, strategy, mode) -> "Stripe":
        h = C.c_void_p()
        self._chk(lib.sarpro_hip_stripe_begin_u16(self._h, _vp(d_b1), _vp(d_b2), rows_total, cols, row0, rows_local,
                                                  in_pitch, int(strategy), int(mode), C.byref(h)))
        return Stripe(self, h)

    def stripe_run_u16(self, d_b1: int, d_b2: int, rows_total: int, cols: int, row0: int, rows_local: int, in_pitch: int,
                       strategy, mode, d_rgb: int, rgb_pitch_px: int):
        """One row stripe, reductions over the library's RCCL communicator, everything on the stream."""
        st = (Stats * 2)()
        self._chk(lib.sarpro_hip_stripe_run_u16(self._h, _vp(d_b1), _vp(d_b2), rows_total, cols, row0, rows_local, in_pitch,
                                                int(strategy), int(mode), _vp(d_rgb), rgb_pitch_px, st))
        return [st[0], st[1]]

    def stripe_run_resized_u16(self, d_b1: int, d_b2: int, rows_total: int, cols: int, row0: int, rows_local: int, in_pitch: int,
                               strategy, mode, target_size: int, pad: bool, d_rgb_slice: int):
        """One row stripe of a scene -> this rank's rows of the resized, padded RGB product (sarpro_hip_stripe_run_resized_u16).
        Returns (out_row0, out_rows, ResizeMeta); d_rgb_slice receives out_rows x final_cols x 3 bytes (host_stripe_resized_rows sizes it)."""
        from ._lib import ResizeMeta
        m = ResizeMeta()
        r0, nr = C.c_size_t(), C.c_size_t()
        self._chk(lib.sarpro_hip_stripe_run_resized_u16(self._h, _vp(d_b1), _vp(d_b2), rows_total, cols, row0, rows_local, in_pitch, int(strategy),
                                                        int(mode), int(target_size or 0), int(bool(pad)), _vp(d_rgb_slice), C.byref(r0), C.byref(nr), C.byref(m)))
        return int(r0.value), int(nr.value), m

    def stripe_run_resized_f32(self, d_b1: int, d_b2: int, rows_total: int, cols: int, row0: int, rows_local: int, in_pitch: int,
                               strategy, mode, target_size: int, pad: bool, d_rgb_slice: int, plain_pipeline: bool = True):
        """stripe_run_resized_u16 for f32 bands (sarpro_hip_stripe_run_resized_f32); Tamed needs plain_pipeline."""
        from ._lib import ResizeMeta
        m = ResizeMeta()
        r0, nr = C.c_size_t(), C.c_size_t()
        self._chk(lib.sarpro_hip_stripe_run_resized_f32(self._h, _vp(d_b1), _vp(d_b2), rows_total, cols, row0, rows_local, in_pitch, int(strategy),
                                                        int(mode), 1 if plain_pipeline else 0, int(target_size or 0), int(bool(pad)), _vp(d_rgb_slice),
                                                        C.byref(r0), C.byref(nr), C.byref(m)))
        return int(r0.value), int(nr.value), m

    def stripe_begin_f32(self, d_in: int, rows_total: int, cols: int, row0: int, rows_local: int, in_pitch: int, strategy, bit_depth,
                         d_out: int, out_pitch: int) -> "StripeF32":
        h = C.c_void_p()
        self._chk(lib.sarpro_hip_stripe_begin_f32(self._h, _vp(d_in), rows_total, cols, row0, rows_local, in_pitch, int(strategy),
                                                  int(bit_depth), _vp(d_out), out_pitch, C.byref(h)))
        return StripeF32(self, h)

    def stripe_begin_polop(self, op, d_a: int, d_b: int, u16_in: bool, rows_total: int, cols: int, row0: int, rows_local: int,
                           in_pitch: int, strategy, bit_depth, d_out: int, out_pitch: int) -> "StripeF32":
        h = C.c_void_p()
        self._chk(lib.sarpro_hip_stripe_begin_polop(self._h, int(op), _vp(d_a), _vp(d_b), int(bool(u16_in)), rows_total, cols, row0,
                                                    rows_local, in_pitch, int(strategy), int(bit_depth), _vp(d_out), out_pitch, C.byref(h)))
        return StripeF32(self, h)

    def stripe_run_f32(self, d_in: int, rows_total: int, cols: int, row0: int, rows_local: int, in_pitch: int, strategy, bit_depth,
                       d_out: int, out_pitch: int) -> Stats:
        """One row stripe of an f32 band, reductions over the library's RCCL communicator."""
        st = Stats()
        self._chk(lib.sarpro_hip_stripe_run_f32(self._h, _vp(d_in), rows_total, cols, row0, rows_local, in_pitch, int(strategy),
                                                int(bit_depth), _vp(d_out), out_pitch, C.byref(st)))
        return st

    def stripe_run_polop(self, op, d_a: int, d_b: int, u16_in: bool, rows_total: int, cols: int, row0: int, rows_local: int,
                         in_pitch: int, strategy, bit_depth, d_out: int, out_pitch: int) -> Stats:
        st = Stats()
        self._chk(lib.sarpro_hip_stripe_run_polop(self._h, int(op), _vp(d_a), _vp(d_b), int(bool(u16_in)), rows_total, cols, row0,
                                                  rows_local, in_pitch, int(strategy), int(bit_depth), _vp(d_out), out_pitch, C.byref(st)))
        return st

    @staticmethod
    def _as_reader(r, cols):
        """(fn, user, keepalive) of a row reader given as a Python callable or as a (fn_ptr, user_ptr) pair"""
        if isinstance(r, tuple):
            return r[0], r[1], None

        def cb(_user, band, row0, nrows, dst, pitch):
            try:
                buf = (C.c_uint16 * (nrows * pitch)).from_address(dst)
                view = np.frombuffer(buf, np.uint16).reshape(nrows, pitch)[:, :cols]
                return int(r(band, row0, nrows, view) or 0)
            except Exception:  # never unwind through the C frames
                return -1
        f = _lib.ROW_READER(cb)
        return C.cast(f, C.c_void_p), None, f

    def dualpol_synrgb_stream(self, reader, rows: int, cols: int, strategy, mode, sink, chunk_rows: int = 0,
                              want_stats: bool = False):
        """Streaming ingest / egress (sarpro_hip_dualpol_synrgb_stream_u16).

        `reader(band, row0, nrows, dst)` fills `dst`, a (nrows, cols) uint16 view of the library's pinned ring, and
        `sink(row0, nrows, rgb)` consumes an (nrows, cols, 3) uint8 view; both may return a non-zero int to abort.
        Alternatively pass the C callbacks themselves: `reader=(fn_ptr, user_ptr)` / `sink=(fn_ptr, user_ptr)`
        (e.g. TiffPair.reader(), TiffWriter.sink())."""
        def as_sink(k):
            if isinstance(k, tuple):
                return k[0], k[1], None
            def cb(_user, row0, nrows, src, pitch_bytes):
                try:
                    buf = (C.c_uint8 * (nrows * pitch_bytes)).from_address(src)
                    view = np.frombuffer(buf, np.uint8).reshape(nrows, pitch_bytes)[:, :cols * 3].reshape(nrows, cols, 3)
                    return int(k(row0, nrows, view) or 0)
                except Exception:
                    return -1
            f = _lib.ROW_SINK(cb)
            return C.cast(f, C.c_void_p), None, f
        rf, ru, _keep_r = self._as_reader(reader, cols)
        sf, su, _keep_s = as_sink(sink)
        st = (Stats * 2)() if want_stats else None
        self._chk(lib.sarpro_hip_dualpol_synrgb_stream_u16(self._h, rf, ru, rows, cols, int(strategy), int(mode), chunk_rows,
                                                           sf, su, st))
        return [st[0], st[1]] if want_stats else None

    def dualpol_synrgb_resized_stream(self, reader, rows: int, cols: int, strategy, target_size: int | None, pad: bool,
                                      mode=SyntheticRgbMode.Default):
        """sarpro_hip_dualpol_synrgb_resized_stream_u16: the resized / padded synRGB product from a row reader."""
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        rgb = np.empty((fr, fc, 3), np.uint8)
        meta = _lib.ResizeMeta()
        rf, ru, _keep = self._as_reader(reader, cols)
        self._chk(lib.sarpro_hip_dualpol_synrgb_resized_stream_u16(self._h, rf, ru, rows, cols, int(strategy), int(mode),
                                                                   target_size or 0, 1 if pad else 0, _vp(rgb), C.byref(meta)))
        return rgb, dict(final_cols=meta.final_cols, final_rows=meta.final_rows, scale_x=meta.scale_x, scale_y=meta.scale_y,
                         pad_left=meta.pad_left, pad_top=meta.pad_top)

    def comm_init(self, nranks: int, rank: int, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(lib.sarpro_hip_comm_init(self._h, nranks, rank, buf))

    def comm_init_local(self, group: "LocalGroup", rank: int):
        """Join the in-process communicator `group` as `rank` (sarpro_hip_comm_init_local): one context and one host thread per rank."""
        self._chk(lib.sarpro_hip_comm_init_local(self._h, group._h, rank))
        self._group = group  # (keeps the group alive as long as the context)

    def comm_destroy(self):
        """Leave the communicator (sarpro_hip_comm_destroy); an in-process group itself belongs to its creator."""
        lib.sarpro_hip_comm_destroy(self._h)
        self._group = None

    def comm_allreduce_sum_u64(self, d_buf: int, count: int):
        self._chk(lib.sarpro_hip_comm_allreduce_sum_u64(self._h, _vp(d_buf), count))


class LocalGroup:
    """sarpro_hip_local_group: the communicator of `nranks` contexts of this process (one thread each); no RCCL, no rendezvous."""

    def __init__(self, nranks: int):
        h = C.c_void_p()
        rc = lib.sarpro_hip_local_group_create(nranks, C.byref(h))
        if rc != _lib.OK:
            raise SarproHipError(rc, "sarpro_hip_local_group_create failed")
        self._h, self.nranks = h, nranks

    def close(self):
        if getattr(self, "_h", None):
            lib.sarpro_hip_local_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stripe:
    """sarpro_hip_stripe: phases of one row stripe; the caller all-reduces between phases."""

    def __init__(self, ctx: Context, h):
        self.ctx, self._h = ctx, h

    def _phase(self, fn):
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx._chk(fn(self._h, C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def phase1(self):
        return self._phase(lib.sarpro_hip_stripe_phase1)

    def phase2(self):
        return self._phase(lib.sarpro_hip_stripe_phase2)

    def phase3(self):
        return self._phase(lib.sarpro_hip_stripe_phase3)

    def phase4(self, d_rgb: int, rgb_pitch_px: int):
        st = (Stats * 2)()
        self.ctx._chk(lib.sarpro_hip_stripe_phase4(self._h, _vp(d_rgb), rgb_pitch_px, st))
        return [st[0], st[1]]

    def end(self):
        if self._h:
            lib.sarpro_hip_stripe_end(self._h)
            self._h = None


class StripeF32:
    """sarpro_hip_stripe_f32: phases of one row stripe of the f32 flavour; the caller merges between phases."""

    def __init__(self, ctx: Context, h):
        self.ctx, self._h = ctx, h

    def phase1(self) -> F32Partial:
        p = F32Partial()
        self.ctx._chk(lib.sarpro_hip_stripe_f32_phase1(self._h, C.byref(p)))
        return p

    def phase2(self, merged: F32Partial):
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx._chk(lib.sarpro_hip_stripe_f32_phase2(self._h, C.byref(merged), C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def _phase(self, fn):
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx._chk(fn(self._h, C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def phase3(self):
        return self._phase(lib.sarpro_hip_stripe_f32_phase3)

    def phase4(self):
        return self._phase(lib.sarpro_hip_stripe_f32_phase4)

    def phase5(self) -> Stats:
        st = Stats()
        self.ctx._chk(lib.sarpro_hip_stripe_f32_phase5(self._h, C.byref(st)))
        return st

    def end(self):
        if self._h:
            lib.sarpro_hip_stripe_f32_end(self._h)
            self._h = None


def host_f32_merge_partials(parts) -> F32Partial:
    """Merge of the stripes' partials in rank order (count, moments: sum; min / max)."""
    arr = (F32Partial * len(parts))(*parts)
    out = F32Partial()
    rc = lib.sarpro_hip_host_f32_merge_partials(arr, len(parts), C.byref(out))
    if rc:
        raise SarproHipError(rc, "host_f32_merge_partials")
    return out


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    rc = lib.sarpro_hip_comm_unique_id(buf)
    if rc != _lib.OK:
        raise SarproHipError(rc, "ncclGetUniqueId failed")
    return bytes(buf)


# ---------------------------------------------------------------------- host half (no GPU)
def host_stats_from_dn_hist(dn_hist: np.ndarray) -> Stats:
    h = np.ascontiguousarray(dn_hist, np.uint64)
    assert h.size == 65536
    st = Stats()
    rc = lib.sarpro_hip_host_stats_from_dn_hist(_vp(h), C.byref(st))
    if rc:
        raise SarproHipError(rc, "host_stats_from_dn_hist")
    return st


def host_window(st: Stats, strategy, tamed_synrgb: int = 0) -> Stats:
    rc = lib.sarpro_hip_host_window(C.byref(st), int(strategy), tamed_synrgb)
    if rc:
        raise SarproHipError(rc, "host_window")
    return st


def host_level_lut_u16(st: Stats, bit_depth, tamed_synrgb: int = 0) -> np.ndarray:
    lut = np.empty(65536, np.uint16)
    rc = lib.sarpro_hip_host_level_lut_u16(C.byref(st), int(bit_depth), tamed_synrgb, _vp(lut))
    if rc:
        raise SarproHipError(rc, "host_level_lut_u16")
    return lut


def host_clahe_bin_lut_u16(st: Stats) -> np.ndarray:
    lut = np.empty(65536, np.uint8)
    rc = lib.sarpro_hip_host_clahe_bin_lut_u16(C.byref(st), _vp(lut))
    if rc:
        raise SarproHipError(rc, "host_clahe_bin_lut_u16")
    return lut


def host_clahe_cdfs(tile_hists: np.ndarray, rows: int, cols: int) -> np.ndarray:
    th = np.ascontiguousarray(tile_hists, np.uint64)
    assert th.size == 64 * 256
    cdfs = np.empty((64, 256), np.float64)
    rc = lib.sarpro_hip_host_clahe_cdfs(_vp(th), rows, cols, _vp(cdfs))
    if rc:
        raise SarproHipError(rc, "host_clahe_cdfs")
    return cdfs


def host_u8_rescale_lut(min_level: int, max_level: int) -> np.ndarray:
    lut = np.empty(256, np.uint8)
    rc = lib.sarpro_hip_host_u8_rescale_lut(min_level, max_level, _vp(lut))
    if rc:
        raise SarproHipError(rc, "host_u8_rescale_lut")
    return lut


def host_synrgb_luts(strategy, combined_hist: np.ndarray | None = None, n_per_band: int = 0):
    luts = np.empty(66048, np.uint8)
    fl = C.c_int(-1)
    h = None if combined_hist is None else np.ascontiguousarray(combined_hist, np.uint64)
    rc = lib.sarpro_hip_host_synrgb_luts(int(strategy), _vp(h), n_per_band, _vp(luts), C.byref(fl))
    if rc:
        raise SarproHipError(rc, "host_synrgb_luts")
    return luts[:256].copy(), luts[256:512].copy(), luts[512:].reshape(256, 256).copy(), fl.value


def host_clahe_saturated_levels(rows: int, cols: int):
    """(col_class[cols], row_bits[rows]) of sarpro_hip_host_clahe_saturated_levels; the level of a saturated pixel (r, c) is
    255 if (row_bits[r] >> col_class[c]) & 1 else 254."""
    cc, rb = np.zeros(cols, np.uint8), np.zeros(rows, np.uint8)
    rc = lib.sarpro_hip_host_clahe_saturated_levels(rows, cols, _vp(cc), _vp(rb))
    if rc != _lib.OK:
        raise SarproHipError(rc, "no saturation table for this shape")
    return cc, rb


def host_clahe_shape_ok(rows: int, cols: int) -> bool:
    return bool(lib.sarpro_hip_host_clahe_shape_ok(rows, cols))


def host_stripe_plan(rows: int, nranks: int):
    r0 = np.empty(nranks, np.uint64)
    nr = np.empty(nranks, np.uint64)
    rc = lib.sarpro_hip_host_stripe_plan(rows, nranks, _vp(r0), _vp(nr))
    if rc:
        raise SarproHipError(rc, "host_stripe_plan")
    return [int(x) for x in r0], [int(x) for x in nr]


def host_stripe_resized_rows(rows_total: int, cols: int, row0: int, rows_local: int, target_size: int, pad: bool):
    """Which rows of the resized, padded product the holder of input rows [row0, row0 + rows_local) produces:
    (out_row0, out_rows, final_cols, final_rows) (sarpro_hip_stripe_resized_rows)."""
    v = [C.c_size_t() for _ in range(4)]
    rc = lib.sarpro_hip_stripe_resized_rows(rows_total, cols, row0, rows_local, int(target_size or 0), int(bool(pad)), *[C.byref(x) for x in v])
    if rc:
        raise SarproHipError(rc, "stripe_resized_rows")
    return tuple(int(x.value) for x in v)


def host_f32_valid_threshold() -> float:
    return float(lib.sarpro_hip_host_f32_valid_threshold())


def host_f32_bin4096_thresholds(min_db: float, max_db: float) -> np.ndarray:
    thr = np.empty(4096, np.float32)
    rc = lib.sarpro_hip_host_f32_bin4096_thresholds(min_db, max_db, _vp(thr))
    if rc:
        raise SarproHipError(rc, "host_f32_bin4096_thresholds")
    return thr


def host_f32_level_thresholds(st: Stats, bit_depth) -> np.ndarray:
    thr = np.empty(256 if int(bit_depth) == 0 else 65536, np.float32)
    rc = lib.sarpro_hip_host_f32_level_thresholds(C.byref(st), int(bit_depth), _vp(thr))
    if rc:
        raise SarproHipError(rc, "host_f32_level_thresholds")
    return thr


def host_f32_clahe_bin_thresholds(st: Stats) -> np.ndarray:
    thr = np.empty(256, np.float32)
    rc = lib.sarpro_hip_host_f32_clahe_bin_thresholds(C.byref(st), _vp(thr))
    if rc:
        raise SarproHipError(rc, "host_f32_clahe_bin_thresholds")
    return thr


def host_stats_from_bins4096(count: int, min_db: float, max_db: float, mean_db: float, std_db: float,
                             hist4096: np.ndarray) -> Stats:
    h = np.ascontiguousarray(hist4096, np.uint64)
    assert h.size == 4096
    st = Stats()
    rc = lib.sarpro_hip_host_stats_from_bins4096(count, min_db, max_db, mean_db, std_db, _vp(h), C.byref(st))
    if rc:
        raise SarproHipError(rc, "host_stats_from_bins4096")
    return st


def resize_output_dims(cols: int, rows: int, target_size: int | None, pad: bool):
    fc, fr = C.c_size_t(), C.c_size_t()
    rc = lib.sarpro_hip_resize_output_dims(cols, rows, target_size or 0, int(pad), C.byref(fc), C.byref(fr))
    if rc:
        raise SarproHipError(rc, "resize_output_dims")
    return fc.value, fr.value


def batch_dualpol_synrgb_resized(devices, scenes, strategy, target_size, pad, mode=SyntheticRgbMode.Default,
                                 continue_on_error: bool = True, workers_per_device: int = 0):
    """process_directory_to_path semantics (api/mod.rs:474-536).
    scenes: list of (band1_u16, band2_u16) arrays, or of (reader, rows, cols) with reader a (fn_ptr, user_ptr) pair such as
    TiffPair.reader() -- the scene is then streamed from its files by the worker that picks it up.
    Returns (list of RGB arrays or None, BatchReport, statuses, rc)."""
    from ._lib import BatchReport, BatchScene
    n = len(scenes)
    arr = (BatchScene * max(n, 1))()
    keep, outs, stats = [], [], (C.c_int * max(n, 1))()
    for i, sc in enumerate(scenes):
        if len(sc) == 3:  # (reader, rows, cols)
            (rf, ru), rows, cols = sc
            fc, fr = resize_output_dims(cols, rows, target_size, pad)
            rgb = np.empty((fr, fc, 3), np.uint8)
            keep.append(sc)
            outs.append(rgb)
            arr[i] = BatchScene(None, None, rows, cols, rgb.ctypes.data,
                                C.cast(C.byref(stats, i * C.sizeof(C.c_int)), C.POINTER(C.c_int)), rf, ru)
            continue
        b1, b2 = sc
        b1 = np.ascontiguousarray(b1, np.uint16)
        b2 = np.ascontiguousarray(b2, np.uint16)
        rows, cols = b1.shape
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        rgb = np.empty((fr, fc, 3), np.uint8)
        keep.append((b1, b2))
        outs.append(rgb)
        arr[i] = BatchScene(b1.ctypes.data, b2.ctypes.data if b2.shape == b1.shape else None, rows, cols, rgb.ctypes.data,
                            C.cast(C.byref(stats, i * C.sizeof(C.c_int)), C.POINTER(C.c_int)), None, None)
    dev = (C.c_int * len(devices))(*devices)
    rep = BatchReport()
    rc = lib.sarpro_hip_batch_dualpol_synrgb_resized_u16(dev, len(devices), workers_per_device, arr, n, int(strategy), int(mode), target_size or 0,
                                                         int(pad), int(continue_on_error), C.byref(rep))
    st = [stats[i] for i in range(n)]
    return [o if s == 0 else None for o, s in zip(outs, st)], rep, st, rc


def batch_dualpol_synrgb_resized_f32(devices, scenes, strategy, target_size, pad, mode=SyntheticRgbMode.Default, plain_pipeline: bool = False,
                                     continue_on_error: bool = True, workers_per_device: int = 0):
    """The batch driver for scenes with f32 bands: scenes = list of (band1_f32, band2_f32).  Returns (RGB arrays or None, BatchReport, statuses, rc)."""
    from ._lib import BatchReport, BatchSceneF32
    n = len(scenes)
    arr = (BatchSceneF32 * max(n, 1))()
    keep, outs, stats = [], [], (C.c_int * max(n, 1))()
    for i, (b1, b2) in enumerate(scenes):
        b1 = np.ascontiguousarray(b1, np.float32)
        b2 = np.ascontiguousarray(b2, np.float32)
        rows, cols = b1.shape
        fc, fr = resize_output_dims(cols, rows, target_size, pad)
        rgb = np.empty((fr, fc, 3), np.uint8)
        keep.append((b1, b2))
        outs.append(rgb)
        arr[i] = BatchSceneF32(b1.ctypes.data, b2.ctypes.data if b2.shape == b1.shape else None, rows, cols, rgb.ctypes.data,
                               C.cast(C.byref(stats, i * C.sizeof(C.c_int)), C.POINTER(C.c_int)))
    dev = (C.c_int * len(devices))(*devices)
    rep = BatchReport()
    rc = lib.sarpro_hip_batch_dualpol_synrgb_resized_f32(dev, len(devices), workers_per_device, arr, n, int(strategy), int(mode), 1 if plain_pipeline else 0,
                                                         target_size or 0, int(pad), int(continue_on_error), C.byref(rep))
    st = [stats[i] for i in range(n)]
    return [o if s == 0 else None for o, s in zip(outs, st)], rep, st, rc


# ---- uncompressed strip TIFF / BigTIFF shims (no GPU) ----
def _tiff_chk(rc):
    if rc != _lib.OK:
        raise SarproHipError(rc, (lib.sarpro_hip_tiff_last_error() or b"").decode())


class TiffReader:
    """sarpro_hip_tiff_open / read_rows_u16 / close."""

    def __init__(self, path: str):
        h = C.c_void_p()
        self.info = _lib.TiffInfo()
        _tiff_chk(lib.sarpro_hip_tiff_open(path.encode(), C.byref(h), C.byref(self.info)))
        self._h = h

    def read_rows(self, row0: int, nrows: int, sample: int = 0) -> np.ndarray:
        out = np.empty((nrows, int(self.info.width)), np.uint16)
        _tiff_chk(lib.sarpro_hip_tiff_read_rows_u16(self._h, sample, row0, nrows, _vp(out), out.shape[1]))
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib.sarpro_hip_tiff_close(self._h)
            self._h = None

    __del__ = close


class TiffPair:
    """Two single-band files as the (fn, user) row reader of Context.dualpol_synrgb_stream."""

    def __init__(self, a: TiffReader, b: TiffReader):
        self._arr = (C.c_void_p * 2)(a._h, b._h)
        self._keep = (a, b)

    def reader(self):
        return C.cast(lib.sarpro_hip_tiff_pair_reader, C.c_void_p), C.cast(self._arr, C.c_void_p)


class TiffWriter:
    """sarpro_hip_tiff_create / write_rows / finish."""

    def __init__(self, path: str, width: int, height: int, samples: int, bits: int, geotransform=None,
                 geo_keys_from: TiffReader | None = None):
        h = C.c_void_p()
        gt = (C.c_double * 6)(*geotransform) if geotransform is not None else None
        _tiff_chk(lib.sarpro_hip_tiff_create(path.encode(), width, height, samples, bits, gt,
                                             geo_keys_from._h if geo_keys_from else None, C.byref(h)))
        self._h = h

    def write_rows(self, row0: int, data: np.ndarray):
        data = np.ascontiguousarray(data)
        nrows = data.shape[0]
        _tiff_chk(lib.sarpro_hip_tiff_write_rows(self._h, row0, nrows, _vp(data), data.nbytes // max(nrows, 1)))

    def sink(self):
        return C.cast(lib.sarpro_hip_tiff_row_sink, C.c_void_p), self._h

    def finish(self):
        h, self._h = self._h, None
        if h:
            _tiff_chk(lib.sarpro_hip_tiff_finish(h))


def host_update_geotransform(gt, cols: int, rows: int, meta: dict):
    """save.rs:71-81"""
    g = (C.c_double * 6)(*gt)
    m = _lib.ResizeMeta(meta["final_cols"], meta["final_rows"], meta["scale_x"], meta["scale_y"], meta["pad_left"], meta["pad_top"])
    lib.sarpro_hip_host_update_geotransform(g, cols, rows, C.byref(m))
    return list(g)

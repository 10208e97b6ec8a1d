"""Loads libsarpro_hip.so (the C-ABI boundary) and declares its prototypes.

There is no Python or CPU fallback: if the shared library has not been built the import
fails, and every raster entry point needs a HIP device (ctx_create fails without one).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SARPRO_HIP_LIB: another build of the same library (A/B timing of kernel variants inside one process tree)
LIB_PATH = os.environ.get("SARPRO_HIP_LIB") or os.path.join(_HERE, "libsarpro_hip.so")

OK = 0
ERR_INVALID_ARG, ERR_SHAPE_MISMATCH, ERR_UNSUPPORTED_SHAPE = -1, -2, -3
ERR_HIP, ERR_RCCL, ERR_OOM, ERR_NO_DEVICE, ERR_IO = -4, -5, -6, -7, -8


class Stats(C.Structure):
    _fields_ = [("valid_count", C.c_uint64)] + [
        (n, C.c_double)
        for n in ("min_db", "max_db", "mean_db", "std_db", "median_db", "p01", "p02", "p05", "p10",
                  "p25", "p75", "p90", "p95", "p98", "p99", "low_clip", "high_clip", "gamma",
                  "skew_factor", "tail_heaviness")
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class F32Partial(C.Structure):
    """sarpro_hip_f32_partial: count / dB moments / min / max of the valid samples of one stripe."""
    _fields_ = [("count", C.c_uint64), ("sum_db", C.c_double), ("sumsq_db", C.c_double), ("min_v", C.c_float), ("max_v", C.c_float)]


class ResizeMeta(C.Structure):
    _fields_ = [("final_cols", C.c_size_t), ("final_rows", C.c_size_t), ("scale_x", C.c_double), ("scale_y", C.c_double),
                ("pad_left", C.c_size_t), ("pad_top", C.c_size_t)]


class TiffInfo(C.Structure):
    _fields_ = [("width", C.c_uint64), ("height", C.c_uint64), ("rows_per_strip", C.c_uint64)] + [
        (n, C.c_uint32) for n in ("bits_per_sample", "samples_per_pixel", "sample_format", "planar", "compression",
                                  "big_endian", "bigtiff", "tiled", "has_geo", "tiepoint_count")
    ] + [("pixel_scale", C.c_double * 3), ("tiepoint", C.c_double * 6)]


ROW_READER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t)
ROW_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t)


class BatchScene(C.Structure):
    _fields_ = [("band1", C.c_void_p), ("band2", C.c_void_p), ("rows", C.c_size_t), ("cols", C.c_size_t),
                ("rgb_out", C.c_void_p), ("status_out", C.POINTER(C.c_int)), ("reader", C.c_void_p), ("reader_user", C.c_void_p)]


class BatchSceneF32(C.Structure):
    _fields_ = [("band1", C.c_void_p), ("band2", C.c_void_p), ("rows", C.c_size_t), ("cols", C.c_size_t),
                ("rgb_out", C.c_void_p), ("status_out", C.POINTER(C.c_int))]


class ResidentSceneF32(C.Structure):
    """sarpro_hip_resident_scene_f32: device pointers in, status out (sarpro_hip_batch_dualpol_synrgb_resized_f32_dev)."""
    _fields_ = [("d_band1", C.c_void_p), ("d_band2", C.c_void_p), ("d_rgb", C.c_void_p), ("status", C.c_int)]


class ResidentScene(C.Structure):
    """sarpro_hip_resident_scene: device pointers in, status / route out (sarpro_hip_batch_dualpol_synrgb_u16_dev)."""
    _fields_ = [("d_band1", C.c_void_p), ("d_band2", C.c_void_p), ("d_rgb", C.c_void_p), ("status", C.c_int), ("route", C.c_int)]


class BatchReport(C.Structure):  # api/mod.rs:453-458
    _fields_ = [("processed", C.c_size_t), ("skipped", C.c_size_t), ("errors", C.c_size_t)]


# every symbol include/sarpro_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "sarpro_hip_ctx_create", "sarpro_hip_ctx_destroy", "sarpro_hip_last_error", "sarpro_hip_version",
    "sarpro_hip_ctx_stream", "sarpro_hip_ctx_synchronize",
    "sarpro_hip_autoscale_band_f32", "sarpro_hip_autoscale_band_u16", "sarpro_hip_db_mask_f32",
    "sarpro_hip_tamed_synrgb_u8_f32", "sarpro_hip_tamed_synrgb_u8_u16", "sarpro_hip_polop_f32",
    "sarpro_hip_synrgb_u8", "sarpro_hip_dualpol_synrgb_u16", "sarpro_hip_dualpol_synrgb_f32",
    "sarpro_hip_autoscale_band_u16_dev", "sarpro_hip_autoscale_band_f32_dev",
    "sarpro_hip_polop_autoscale_band_f32", "sarpro_hip_polop_autoscale_band_u16", "sarpro_hip_polop_autoscale_band_f32_dev", "sarpro_hip_polop_autoscale_band_u16_dev",
    "sarpro_hip_dualpol_synrgb_u16_dev", "sarpro_hip_polop_f32_dev", "sarpro_hip_synrgb_u8_dev",
    "sarpro_hip_last_kernel_times",
    "sarpro_hip_ctx_time_only", "sarpro_hip_ctx_spec_report", "sarpro_hip_ctx_chain_report",
    "sarpro_hip_stripe_begin_u16", "sarpro_hip_stripe_phase1", "sarpro_hip_stripe_phase2",
    "sarpro_hip_stripe_phase3", "sarpro_hip_stripe_phase4", "sarpro_hip_stripe_end", "sarpro_hip_stripe_run_u16", "sarpro_hip_stripe_resized_rows", "sarpro_hip_stripe_run_resized_u16", "sarpro_hip_stripe_run_resized_f32",
    "sarpro_hip_dualpol_synrgb_stream_u16", "sarpro_hip_dualpol_synrgb_resized_stream_u16", "sarpro_hip_tiff_open", "sarpro_hip_tiff_read_rows_u16", "sarpro_hip_tiff_close",
    "sarpro_hip_tiff_pair_reader", "sarpro_hip_tiff_create", "sarpro_hip_tiff_write_rows", "sarpro_hip_tiff_row_sink",
    "sarpro_hip_tiff_finish", "sarpro_hip_tiff_last_error", "sarpro_hip_host_update_geotransform",
    "sarpro_hip_selftest_polop_division",
    "sarpro_hip_stripe_begin_f32", "sarpro_hip_stripe_begin_polop", "sarpro_hip_stripe_f32_phase1", "sarpro_hip_stripe_f32_phase2",
    "sarpro_hip_stripe_f32_phase3", "sarpro_hip_stripe_f32_phase4", "sarpro_hip_stripe_f32_phase5", "sarpro_hip_stripe_f32_end",
    "sarpro_hip_stripe_run_f32", "sarpro_hip_stripe_run_polop", "sarpro_hip_host_f32_merge_partials",
    "sarpro_hip_comm_unique_id", "sarpro_hip_comm_init", "sarpro_hip_comm_allreduce_sum_u64",
    "sarpro_hip_comm_destroy",
    "sarpro_hip_host_stats_from_dn_hist", "sarpro_hip_host_window", "sarpro_hip_host_level_lut_u16",
    "sarpro_hip_host_clahe_bin_lut_u16", "sarpro_hip_host_clahe_cdfs", "sarpro_hip_host_u8_rescale_lut",
    "sarpro_hip_host_synrgb_luts", "sarpro_hip_host_clahe_shape_ok", "sarpro_hip_host_parse_cpulist", "sarpro_hip_host_stripe_plan",
    "sarpro_hip_host_stats_from_bins4096", "sarpro_hip_host_f32_valid_threshold", "sarpro_hip_host_f32_bin4096_thresholds",
    "sarpro_hip_host_f32_level_thresholds", "sarpro_hip_host_f32_clahe_bin_thresholds",
    "sarpro_hip_resize_output_dims", "sarpro_hip_resize_image_data", "sarpro_hip_resize_image_data_dev",
    "sarpro_hip_dualpol_synrgb_resized_u16", "sarpro_hip_dualpol_synrgb_resized_u16_dev", "sarpro_hip_dualpol_synrgb_f32_dev",
    "sarpro_hip_dualpol_synrgb_resized_f32", "sarpro_hip_dualpol_synrgb_resized_f32_dev", "sarpro_hip_batch_dualpol_synrgb_resized_f32", "sarpro_hip_process_band_resized_u16", "sarpro_hip_process_band_resized_f32",
    "sarpro_hip_batch_dualpol_synrgb_resized_u16", "sarpro_hip_batch_dualpol_synrgb_u16_dev", "sarpro_hip_batch_dualpol_synrgb_resized_f32_dev",
    "sarpro_hip_synth_scene_u16_dev", "sarpro_hip_synth_scene_u16_dev_ex",
    "sarpro_hip_local_group_create", "sarpro_hip_local_group_destroy", "sarpro_hip_comm_init_local", "sarpro_hip_host_clahe_saturated_levels", "sarpro_hip_ctx_set_attr", "sarpro_hip_ctx_reset_attr", "sarpro_hip_ctx_get_attr", "sarpro_hip_attr_name",
]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C sarpro_amd/csrc). sarpro_amd has no fallback implementation.")



def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so / libhsa-runtime64.so.  Two HIP runtimes in
    one process cannot both own the GPU ("No HIP GPUs are available" from whichever initialises
    second), so when torch is installed we load ITS runtime first: libsarpro_hip.so's NEEDED
    libamdhip64.so.7 then binds to the copy torch will use too.  Standalone users (the Rust / C
    callers of the C ABI) get the system ROCm runtime.  SARPRO_HIP_RUNTIME=system skips this."""
    if os.environ.get("SARPRO_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


_share_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)

_vp, _sz, _i, _u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
_S = C.POINTER(Stats)


def _proto(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)


_proto("sarpro_hip_ctx_create", _i, _i, C.c_uint, C.POINTER(_vp))
_proto("sarpro_hip_ctx_destroy", None, _vp)
_proto("sarpro_hip_last_error", C.c_char_p, _vp)
_proto("sarpro_hip_version", C.c_char_p)
_proto("sarpro_hip_ctx_stream", _vp, _vp)
_proto("sarpro_hip_ctx_synchronize", _i, _vp)
_proto("sarpro_hip_autoscale_band_f32", _i, _vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _S)
_proto("sarpro_hip_autoscale_band_u16", _i, _vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _S)
_proto("sarpro_hip_db_mask_f32", _i, _vp, _vp, _sz, _sz, _vp, _vp)
_proto("sarpro_hip_tamed_synrgb_u8_f32", _i, _vp, _vp, _sz, _sz, _i, _vp)
_proto("sarpro_hip_tamed_synrgb_u8_u16", _i, _vp, _vp, _sz, _sz, _i, _vp)
_proto("sarpro_hip_polop_f32", _i, _vp, _i, _vp, _vp, _sz, _vp)
_proto("sarpro_hip_synrgb_u8", _i, _vp, _i, _i, _vp, _vp, _sz, _vp)
_proto("sarpro_hip_dualpol_synrgb_u16", _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _vp, _S)
_proto("sarpro_hip_dualpol_synrgb_f32", _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _vp, _S)
_proto("sarpro_hip_autoscale_band_u16_dev", _i, _vp, _vp, _sz, _sz, _sz, _i, _i, _vp, _sz, _S)
_proto("sarpro_hip_autoscale_band_f32_dev", _i, _vp, _vp, _sz, _sz, _sz, _i, _i, _vp, _sz, _S)
_proto("sarpro_hip_dualpol_synrgb_u16_dev", _i, _vp, _vp, _vp, _sz, _sz, _sz, _i, _i, _vp, _sz, _vp, _vp, _sz, _S)
for _n in ("sarpro_hip_polop_autoscale_band_f32", "sarpro_hip_polop_autoscale_band_u16"):
    _proto(_n, _i, _vp, _i, _vp, _vp, _sz, _sz, _i, _i, _vp, _vp, _S)
for _n in ("sarpro_hip_polop_autoscale_band_f32_dev", "sarpro_hip_polop_autoscale_band_u16_dev"):
    _proto(_n, _i, _vp, _i, _vp, _vp, _sz, _sz, _sz, _i, _i, _vp, _sz, _S)
_proto("sarpro_hip_polop_f32_dev", _i, _vp, _i, _vp, _vp, _sz, _vp)
_proto("sarpro_hip_synrgb_u8_dev", _i, _vp, _i, _i, _vp, _vp, _sz, _vp)
_proto("sarpro_hip_last_kernel_times", _i, _vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), _i)
_proto("sarpro_hip_ctx_time_only", _i, _vp, C.c_char_p)


class SpecReport(C.Structure):
    _fields_ = [("spec_ok", C.c_uint32), ("verdict", C.c_uint32), ("floor_pred", C.c_int32), ("pool_overflow", C.c_uint32),
                ("n_lt", C.c_uint64 * 2), ("target", C.c_uint64), ("est_lt", C.c_double * 2), ("sample_valid", C.c_uint64 * 2),
                ("n_below_min", C.c_uint64), ("min_pred", C.c_uint32 * 2), ("retried", C.c_uint32), ("floor_first", C.c_int32)]


_proto("sarpro_hip_ctx_spec_report", _i, _vp, C.POINTER(SpecReport))


class ChainReport(C.Structure):
    _fields_ = [("floor_with_cushion", C.c_int32), ("identity", C.c_uint8 * 2), ("reserved", C.c_uint8 * 2),
                ("rescale", C.c_uint8 * 512), ("level_hist", C.c_uint64 * 512)]


_proto("sarpro_hip_ctx_chain_report", _i, _vp, C.POINTER(ChainReport))
_proto("sarpro_hip_ctx_set_attr", _i, _vp, C.c_char_p, C.c_int64)
_proto("sarpro_hip_ctx_reset_attr", _i, _vp, C.c_char_p)
_proto("sarpro_hip_ctx_get_attr", _i, _vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int))
_proto("sarpro_hip_attr_name", C.c_char_p, _i)
_proto("sarpro_hip_stripe_begin_u16", _i, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _i, C.POINTER(_vp))
_proto("sarpro_hip_stripe_phase1", _i, _vp, C.POINTER(_vp), C.POINTER(_sz))
_proto("sarpro_hip_stripe_phase2", _i, _vp, C.POINTER(_vp), C.POINTER(_sz))
_proto("sarpro_hip_stripe_phase3", _i, _vp, C.POINTER(_vp), C.POINTER(_sz))
_proto("sarpro_hip_stripe_phase4", _i, _vp, _vp, _sz, _S)
_proto("sarpro_hip_stripe_end", None, _vp)
_proto("sarpro_hip_selftest_polop_division", _i, _vp, C.POINTER(C.c_uint64))
_proto("sarpro_hip_stripe_begin_f32", _i, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _i, _vp, _sz, C.POINTER(_vp))
_proto("sarpro_hip_stripe_begin_polop", _i, _vp, _i, _vp, _vp, _i, _sz, _sz, _sz, _sz, _sz, _i, _i, _vp, _sz, C.POINTER(_vp))
_proto("sarpro_hip_stripe_f32_phase1", _i, _vp, C.POINTER(F32Partial))
_proto("sarpro_hip_stripe_f32_phase2", _i, _vp, C.POINTER(F32Partial), C.POINTER(_vp), C.POINTER(_sz))
_proto("sarpro_hip_stripe_f32_phase3", _i, _vp, C.POINTER(_vp), C.POINTER(_sz))
_proto("sarpro_hip_stripe_f32_phase4", _i, _vp, C.POINTER(_vp), C.POINTER(_sz))
_proto("sarpro_hip_stripe_f32_phase5", _i, _vp, _S)
_proto("sarpro_hip_stripe_f32_end", None, _vp)
_proto("sarpro_hip_stripe_run_f32", _i, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _i, _vp, _sz, _S)
_proto("sarpro_hip_stripe_run_polop", _i, _vp, _i, _vp, _vp, _i, _sz, _sz, _sz, _sz, _sz, _i, _i, _vp, _sz, _S)
_proto("sarpro_hip_host_f32_merge_partials", _i, _vp, _sz, C.POINTER(F32Partial))
_proto("sarpro_hip_comm_unique_id", _i, _vp)
_proto("sarpro_hip_comm_init", _i, _vp, _i, _i, _vp)
_proto("sarpro_hip_comm_allreduce_sum_u64", _i, _vp, _vp, _sz)
_proto("sarpro_hip_local_group_create", _i, _i, C.POINTER(_vp))
_proto("sarpro_hip_local_group_destroy", None, _vp)
_proto("sarpro_hip_comm_init_local", _i, _vp, _vp, _i)
_proto("sarpro_hip_comm_destroy", None, _vp)
_proto("sarpro_hip_host_stats_from_dn_hist", _i, _vp, _S)
_proto("sarpro_hip_host_window", _i, _S, _i, _i)
_proto("sarpro_hip_host_level_lut_u16", _i, _S, _i, _i, _vp)
_proto("sarpro_hip_host_clahe_bin_lut_u16", _i, _S, _vp)
_proto("sarpro_hip_host_clahe_cdfs", _i, _vp, _sz, _sz, _vp)
_proto("sarpro_hip_host_u8_rescale_lut", _i, C.c_uint, C.c_uint, _vp)
_proto("sarpro_hip_host_synrgb_luts", _i, _i, _vp, _u64, _vp, C.POINTER(_i))
_proto("sarpro_hip_host_clahe_shape_ok", _i, _sz, _sz)
_proto("sarpro_hip_host_clahe_saturated_levels", _i, _sz, _sz, _vp, _vp)
_proto("sarpro_hip_host_stripe_plan", _i, _sz, _i, _vp, _vp)
_proto("sarpro_hip_synth_scene_u16_dev", _i, _vp, _u64, _i, _vp, _sz, _sz, _sz, _sz, _vp, _sz)
_proto("sarpro_hip_synth_scene_u16_dev_ex", _i, _vp, _u64, _i, _vp, _sz, _sz, _sz, _sz, _vp, _sz, C.c_uint32)
_proto("sarpro_hip_host_f32_valid_threshold", C.c_float)
_proto("sarpro_hip_host_f32_bin4096_thresholds", _i, C.c_double, C.c_double, _vp)
_proto("sarpro_hip_host_f32_level_thresholds", _i, _S, _i, _vp)
_proto("sarpro_hip_host_f32_clahe_bin_thresholds", _i, _S, _vp)
_proto("sarpro_hip_host_stats_from_bins4096", _i, _u64, C.c_double, C.c_double, C.c_double, C.c_double, _vp, _S)
_M = C.POINTER(ResizeMeta)
_proto("sarpro_hip_resize_output_dims", _i, _sz, _sz, _sz, _i, C.POINTER(_sz), C.POINTER(_sz))
_proto("sarpro_hip_resize_image_data", _i, _vp, _vp, _sz, _sz, _sz, _i, _i, _vp, _M)
_proto("sarpro_hip_resize_image_data_dev", _i, _vp, _vp, _sz, _sz, _sz, _sz, _i, _i, _vp, _sz, _M)
_proto("sarpro_hip_dualpol_synrgb_resized_u16", _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _sz, _i, _vp, _M)
_proto("sarpro_hip_host_parse_cpulist", _i, C.c_char_p, _vp, _i)
_u = C.c_uint
_proto("sarpro_hip_dualpol_synrgb_f32_dev", _i, _vp, _vp, _vp, _sz, _sz, _sz, _i, _i, _u, _vp, _sz, _vp, _vp, _sz, C.POINTER(Stats))
_proto("sarpro_hip_dualpol_synrgb_resized_f32", _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _u, _sz, _i, _vp, _M)
_proto("sarpro_hip_dualpol_synrgb_resized_f32_dev", _i, _vp, _vp, _vp, _sz, _sz, _sz, _i, _i, _u, _sz, _i, _vp, _M)
_proto("sarpro_hip_batch_dualpol_synrgb_resized_f32", _i, C.POINTER(C.c_int), _i, _i, C.POINTER(BatchSceneF32), _sz, _i, _i, _u, _sz, _i, _i, C.POINTER(BatchReport))
_proto("sarpro_hip_dualpol_synrgb_resized_u16_dev", _i, _vp, _vp, _vp, _sz, _sz, _sz, _i, _i, _sz, _i, _vp, _M)
_proto("sarpro_hip_process_band_resized_u16", _i, _vp, _vp, _sz, _sz, _i, _i, _sz, _i, _vp, _M)
_proto("sarpro_hip_process_band_resized_f32", _i, _vp, _vp, _sz, _sz, _i, _i, _sz, _i, _vp, _M)
_proto("sarpro_hip_batch_dualpol_synrgb_resized_u16", _i, C.POINTER(_i), _i, _i, C.POINTER(BatchScene), _sz, _i, _i, _sz, _i, _i,
       C.POINTER(BatchReport))
_proto("sarpro_hip_batch_dualpol_synrgb_resized_f32_dev", _i, _vp, C.POINTER(ResidentSceneF32), _sz, _sz, _sz, _sz, _i, _i, C.c_uint, _sz, _i, _i, _i, C.POINTER(BatchReport))
_proto("sarpro_hip_batch_dualpol_synrgb_u16_dev", _i, _vp, C.POINTER(ResidentScene), _sz, _sz, _sz, _sz, _i, _i, _sz, _i, _i, C.POINTER(BatchReport))
_proto("sarpro_hip_stripe_run_u16", _i, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _i, _vp, _sz, _S)
_proto("sarpro_hip_stripe_resized_rows", _i, _sz, _sz, _sz, _sz, _sz, _i, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz))
_proto("sarpro_hip_stripe_run_resized_u16", _i, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _i, _sz, _i, _vp, C.POINTER(_sz), C.POINTER(_sz), _M)
_proto("sarpro_hip_stripe_run_resized_f32", _i, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _i, _i, C.c_uint, _sz, _i, _vp, C.POINTER(_sz), C.POINTER(_sz), _M)
_proto("sarpro_hip_dualpol_synrgb_stream_u16", _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _sz, _vp, _vp, _S)
_proto("sarpro_hip_tiff_open", _i, C.c_char_p, C.POINTER(_vp), C.POINTER(TiffInfo))
_proto("sarpro_hip_tiff_read_rows_u16", _i, _vp, _i, _sz, _sz, _vp, _sz)
_proto("sarpro_hip_tiff_close", None, _vp)
_proto("sarpro_hip_tiff_pair_reader", _i, _vp, _i, _sz, _sz, _vp, _sz)
_proto("sarpro_hip_tiff_create", _i, C.c_char_p, _u64, _u64, C.c_uint32, C.c_uint32, _vp, _vp, C.POINTER(_vp))
_proto("sarpro_hip_tiff_write_rows", _i, _vp, _sz, _sz, _vp, _sz)
_proto("sarpro_hip_tiff_row_sink", _i, _vp, _sz, _sz, _vp, _sz)
_proto("sarpro_hip_tiff_finish", _i, _vp)
_proto("sarpro_hip_tiff_last_error", C.c_char_p)
_proto("sarpro_hip_host_update_geotransform", None, _vp, _sz, _sz, _M)
_proto("sarpro_hip_dualpol_synrgb_resized_stream_u16", _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _sz, _i, _vp, _M)

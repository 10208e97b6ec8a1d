//! Raw `extern "C"` declarations of `include/sarpro_hip.h` and safe wrappers that carry the
//! signatures of the functions they replace in sarpro's `src/core/processing`:
//!
//! | sarpro function | wrapper here |
//! |---|---|
//! | `pipeline::process_scalar_data_pipeline` (pipeline.rs:42) | [`RasterCore::process_scalar_data_pipeline`] |
//! | `autoscale::autoscale_db_image_tamed_synrgb_u8` (autoscale.rs:710) | [`RasterCore::autoscale_tamed_synrgb_u8`] |
//! | `ops::{sum,difference,ratio,normalized_diff,log_ratio}_arrays` (ops.rs:4-44) | [`RasterCore::polop`] |
//! | `synthetic_rgb::create_synthetic_rgb_by_mode_and_strategy` (synthetic_rgb.rs:182) | [`RasterCore::create_synthetic_rgb_by_mode_and_strategy`] |
//! | JPEG branch of `save_processed_multiband_image_sequential` (save.rs:317-367) | [`RasterCore::dualpol_synrgb`] |
//!
//! Enum discriminants are the declaration order of sarpro's `src/types.rs`, so
//! `strategy as i32` of the crate's own enums can be passed straight through.
#![allow(non_camel_case_types)]

use ndarray::Array2;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_uint, c_void};

#[repr(C)]
pub struct sarpro_hip_ctx {
    _private: [u8; 0],
}

/// `HistogramStats` (autoscale.rs:7-24) + the chosen window.
#[repr(C)]
#[derive(Debug, Default, Clone, Copy)]
pub struct sarpro_hip_stats {
    pub valid_count: u64,
    pub min_db: f64, pub max_db: f64, pub mean_db: f64, pub std_db: f64, pub median_db: f64,
    pub p01: f64, pub p02: f64, pub p05: f64, pub p10: f64, pub p25: f64,
    pub p75: f64, pub p90: f64, pub p95: f64, pub p98: f64, pub p99: f64,
    pub low_clip: f64, pub high_clip: f64, pub gamma: f64,
    pub skew_factor: f64, pub tail_heaviness: f64,
}

/// Bookkeeping of `resize_image_data_with_meta` (resize.rs:98-108).
#[repr(C)]
#[derive(Debug, Default, Clone, Copy)]
pub struct sarpro_hip_resize_meta {
    pub final_cols: usize, pub final_rows: usize,
    pub scale_x: f64, pub scale_y: f64,
    pub pad_left: usize, pub pad_top: usize,
}

/// One scene of a batch (`process_directory_to_path`, api/mod.rs:474-536).
pub type sarpro_hip_row_reader = Option<unsafe extern "C" fn(user: *mut c_void, band: c_int, row0: usize, nrows: usize, dst: *mut u16, dst_pitch: usize) -> c_int>;
pub type sarpro_hip_row_sink = Option<unsafe extern "C" fn(user: *mut c_void, row0: usize, nrows: usize, src: *const u8, src_pitch_bytes: usize) -> c_int>;

#[repr(C)]
pub struct sarpro_hip_batch_scene {
    pub band1: *const u16, pub band2: *const u16,
    pub rows: usize, pub cols: usize,
    pub rgb_out: *mut u8,
    pub status_out: *mut c_int,
    /// optional: the scene's bands come through this row reader (band1 / band2 are then ignored)
    pub reader: sarpro_hip_row_reader,
    pub reader_user: *mut c_void,
}

/// `BatchReport` (api/mod.rs:453-458).
#[repr(C)]
#[derive(Debug, Default, Clone, Copy)]
pub struct sarpro_hip_batch_report { pub processed: usize, pub skipped: usize, pub errors: usize }

pub const SARPRO_HIP_OK: c_int = 0;
pub const SARPRO_HIP_ERR_UNSUPPORTED_SHAPE: c_int = -3;
pub const SARPRO_HIP_CTX_TIMING: c_uint = 1;
/// stream-ordered `sarpro_hip_dualpol_synrgb_u16_dev` (returns once enqueued when `stats_out` is null)
pub const SARPRO_HIP_CTX_ASYNC_DEV: c_uint = 2;

extern "C" {
    pub fn sarpro_hip_ctx_create(device: c_int, flags: c_uint, ctx_out: *mut *mut sarpro_hip_ctx) -> c_int;
    pub fn sarpro_hip_ctx_destroy(ctx: *mut sarpro_hip_ctx);
    pub fn sarpro_hip_last_error(ctx: *const sarpro_hip_ctx) -> *const c_char;
    pub fn sarpro_hip_autoscale_band_f32(ctx: *mut sarpro_hip_ctx, input: *const f32, rows: usize, cols: usize,
        strategy: c_int, bit_depth: c_int, out_u8: *mut u8, out_u16: *mut u16, stats_out: *mut sarpro_hip_stats) -> c_int;
    pub fn sarpro_hip_autoscale_band_u16(ctx: *mut sarpro_hip_ctx, input: *const u16, rows: usize, cols: usize,
        strategy: c_int, bit_depth: c_int, out_u8: *mut u8, out_u16: *mut u16, stats_out: *mut sarpro_hip_stats) -> c_int;
    pub fn sarpro_hip_db_mask_f32(ctx: *mut sarpro_hip_ctx, input: *const f32, rows: usize, cols: usize,
        db_out: *mut f64, mask_out: *mut u8) -> c_int;
    pub fn sarpro_hip_tamed_synrgb_u8_f32(ctx: *mut sarpro_hip_ctx, input: *const f32, rows: usize, cols: usize,
        is_copol: c_int, out_u8: *mut u8) -> c_int;
    pub fn sarpro_hip_polop_f32(ctx: *mut sarpro_hip_ctx, op: c_int, a: *const f32, b: *const f32, n: usize,
        out: *mut f32) -> c_int;
    pub fn sarpro_hip_synrgb_u8(ctx: *mut sarpro_hip_ctx, mode: c_int, strategy: c_int, band1: *const u8,
        band2: *const u8, n: usize, rgb_out: *mut u8) -> c_int;
    pub fn sarpro_hip_dualpol_synrgb_f32(ctx: *mut sarpro_hip_ctx, band1: *const f32, band2: *const f32, rows: usize,
        cols: usize, strategy: c_int, mode: c_int, rgb_out: *mut u8, u8_band1: *mut u8, u8_band2: *mut u8,
        stats_out: *mut sarpro_hip_stats) -> c_int;
    pub fn sarpro_hip_dualpol_synrgb_u16(ctx: *mut sarpro_hip_ctx, band1: *const u16, band2: *const u16, rows: usize,
        cols: usize, strategy: c_int, mode: c_int, rgb_out: *mut u8, u8_band1: *mut u8, u8_band2: *mut u8,
        stats_out: *mut sarpro_hip_stats) -> c_int;
    pub fn sarpro_hip_resize_output_dims(cols: usize, rows: usize, target_size: usize, pad: c_int,
        final_cols: *mut usize, final_rows: *mut usize) -> c_int;
    pub fn sarpro_hip_resize_image_data(ctx: *mut sarpro_hip_ctx, data: *const c_void, cols: usize, rows: usize,
        target_size: usize, bit_depth: c_int, pad: c_int, out: *mut c_void, meta: *mut sarpro_hip_resize_meta) -> c_int;
    pub fn sarpro_hip_process_band_resized_f32(ctx: *mut sarpro_hip_ctx, input: *const f32, rows: usize, cols: usize,
        strategy: c_int, bit_depth: c_int, target_size: usize, pad: c_int, out: *mut c_void,
        meta: *mut sarpro_hip_resize_meta) -> c_int;
    pub fn sarpro_hip_dualpol_synrgb_resized_u16(ctx: *mut sarpro_hip_ctx, band1: *const u16, band2: *const u16,
        rows: usize, cols: usize, strategy: c_int, mode: c_int, target_size: usize, pad: c_int, rgb_out: *mut u8,
        meta: *mut sarpro_hip_resize_meta) -> c_int;
    pub fn sarpro_hip_batch_dualpol_synrgb_resized_u16(devices: *const c_int, ndevices: c_int,
        scenes: *const sarpro_hip_batch_scene, nscenes: usize, strategy: c_int, mode: c_int, target_size: usize,
        pad: c_int, continue_on_error: c_int, report: *mut sarpro_hip_batch_report) -> c_int;
    // device-pointer, stripe, comm and host-half entry points: see include/sarpro_hip.h
    pub fn sarpro_hip_ctx_stream(ctx: *mut sarpro_hip_ctx) -> *mut c_void;
    pub fn sarpro_hip_ctx_time_only(ctx: *mut sarpro_hip_ctx, kernel_name: *const c_char) -> c_int;
    /// streaming ingest / egress: the decoder's read loop and the encoder's write loop as callbacks
    pub fn sarpro_hip_dualpol_synrgb_stream_u16(ctx: *mut sarpro_hip_ctx, reader: sarpro_hip_row_reader, reader_user: *mut c_void,
        rows: usize, cols: usize, strategy: c_int, mode: c_int, chunk_rows: usize, sink: sarpro_hip_row_sink,
        sink_user: *mut c_void, stats_out: *mut sarpro_hip_stats) -> c_int;
    pub fn sarpro_hip_dualpol_synrgb_resized_stream_u16(ctx: *mut sarpro_hip_ctx, reader: sarpro_hip_row_reader, reader_user: *mut c_void,
        rows: usize, cols: usize, strategy: c_int, mode: c_int, target_size: usize, pad: c_int, rgb_out: *mut u8,
        meta: *mut sarpro_hip_resize_meta) -> c_int;
    pub fn sarpro_hip_tiff_open(path: *const c_char, out: *mut *mut c_void, info_out: *mut c_void) -> c_int;
    pub fn sarpro_hip_tiff_close(t: *mut c_void);
    pub fn sarpro_hip_tiff_pair_reader(user: *mut c_void, band: c_int, row0: usize, nrows: usize, dst: *mut u16, dst_pitch: usize) -> c_int;
    pub fn sarpro_hip_tiff_create(path: *const c_char, width: u64, height: u64, samples: u32, bits: u32, geotransform6: *const f64,
        geo_keys_from: *const c_void, out: *mut *mut c_void) -> c_int;
    pub fn sarpro_hip_tiff_row_sink(user: *mut c_void, row0: usize, nrows: usize, src: *const u8, src_pitch_bytes: usize) -> c_int;
    pub fn sarpro_hip_tiff_finish(w: *mut c_void) -> c_int;
}

/// Error type a sarpro integration maps onto `Error::Processing` (src/error.rs:39-46).
#[derive(Debug)]
pub struct HipError { pub code: i32, pub message: String }
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "sarpro_hip {}: {}", self.code, self.message) }
}
impl std::error::Error for HipError {}

/// One context per host thread (the library is re-entrant; contexts are independent).
pub struct RasterCore { ctx: *mut sarpro_hip_ctx }
unsafe impl Send for RasterCore {}

impl RasterCore {
    pub fn new(device: i32) -> Result<Self, HipError> { Self::with_flags(device, 0) }
    pub fn with_flags(device: i32, flags: c_uint) -> Result<Self, HipError> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sarpro_hip_ctx_create(device, flags, &mut ctx) };
        if rc != SARPRO_HIP_OK { return Err(Self::err(std::ptr::null(), rc)); }
        Ok(Self { ctx })
    }
    fn err(ctx: *const sarpro_hip_ctx, code: c_int) -> HipError {
        let p = unsafe { sarpro_hip_last_error(ctx) };
        let message = if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() };
        HipError { code, message }
    }
    fn chk(&self, rc: c_int) -> Result<(), HipError> { if rc == SARPRO_HIP_OK { Ok(()) } else { Err(Self::err(self.ctx, rc)) } }

    /// pipeline.rs:42 -- returns `(scaled_u8, scaled_u16)`; `bit_depth`/`strategy` are `types.rs` discriminants.
    /// The dB buffer and mask are not materialised (save.rs only uses their dims).
    pub fn process_scalar_data_pipeline(&self, processed: &Array2<f32>, bit_depth: i32, strategy: i32)
        -> Result<(Vec<u8>, Option<Vec<u16>>), HipError> {
        let (rows, cols) = processed.dim();
        let src = processed.as_standard_layout();
        let n = rows * cols;
        if bit_depth == 0 {
            let mut out = vec![0u8; n];
            self.chk(unsafe { sarpro_hip_autoscale_band_f32(self.ctx, src.as_ptr(), rows, cols, strategy, 0,
                out.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) })?;
            Ok((out, None))
        } else {
            let mut out = vec![0u16; n];
            self.chk(unsafe { sarpro_hip_autoscale_band_f32(self.ctx, src.as_ptr(), rows, cols, strategy, 1,
                std::ptr::null_mut(), out.as_mut_ptr(), std::ptr::null_mut()) })?;
            Ok((vec![], Some(out)))
        }
    }

    /// autoscale.rs:710 -- takes the band, not the dB buffer (dB is recomputed on the device).
    pub fn autoscale_tamed_synrgb_u8(&self, band: &Array2<f32>, is_copol: bool) -> Result<Vec<u8>, HipError> {
        let (rows, cols) = band.dim();
        let src = band.as_standard_layout();
        let mut out = vec![0u8; rows * cols];
        self.chk(unsafe { sarpro_hip_tamed_synrgb_u8_f32(self.ctx, src.as_ptr(), rows, cols, is_copol as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }

    /// ops.rs:4-44 -- `op` is the `PolarizationOperation` discriminant.
    pub fn polop(&self, op: i32, a: &Array2<f32>, b: &Array2<f32>) -> Result<Array2<f32>, HipError> {
        assert_eq!(a.dim(), b.dim());
        let (a_, b_) = (a.as_standard_layout(), b.as_standard_layout());
        let mut out = Array2::<f32>::zeros(a.dim());
        self.chk(unsafe { sarpro_hip_polop_f32(self.ctx, op, a_.as_ptr(), b_.as_ptr(), a.len(), out.as_mut_ptr()) })?;
        Ok(out)
    }

    /// synthetic_rgb.rs:182
    pub fn create_synthetic_rgb_by_mode_and_strategy(&self, mode: i32, strategy: i32, band1: &[u8], band2: &[u8])
        -> Result<Vec<u8>, HipError> {
        debug_assert_eq!(band1.len(), band2.len());
        let mut rgb = vec![0u8; band1.len() * 3];
        self.chk(unsafe { sarpro_hip_synrgb_u8(self.ctx, mode, strategy, band1.as_ptr(), band2.as_ptr(), band1.len(), rgb.as_mut_ptr()) })?;
        Ok(rgb)
    }

    /// save.rs:317-367 at native resolution: both pipelines, the Tamed re-autoscale and the composition, fused.
    pub fn dualpol_synrgb(&self, band1: &Array2<f32>, band2: &Array2<f32>, strategy: i32, mode: i32) -> Result<Vec<u8>, HipError> {
        assert_eq!(band1.dim(), band2.dim());
        let (rows, cols) = band1.dim();
        let (b1, b2) = (band1.as_standard_layout(), band2.as_standard_layout());
        let mut rgb = vec![0u8; rows * cols * 3];
        self.chk(unsafe { sarpro_hip_dualpol_synrgb_f32(self.ctx, b1.as_ptr(), b2.as_ptr(), rows, cols, strategy, mode,
            rgb.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut()) })?;
        Ok(rgb)
    }
}

impl RasterCore {
    /// `resize_image_data_with_meta` (resize.rs:91) for a u8 raster: returns (final_cols, final_rows, data, meta).
    pub fn resize_image_data_u8(&self, data: &[u8], cols: usize, rows: usize, target_size: Option<usize>, pad: bool)
        -> Result<(usize, usize, Vec<u8>, sarpro_hip_resize_meta), HipError> {
        let (mut fc, mut fr) = (0usize, 0usize);
        self.chk(unsafe { sarpro_hip_resize_output_dims(cols, rows, target_size.unwrap_or(0), pad as c_int, &mut fc, &mut fr) })?;
        let mut out = vec![0u8; fc * fr];
        let mut meta = sarpro_hip_resize_meta::default();
        self.chk(unsafe { sarpro_hip_resize_image_data(self.ctx, data.as_ptr() as *const c_void, cols, rows,
            target_size.unwrap_or(0), 0, pad as c_int, out.as_mut_ptr() as *mut c_void, &mut meta) })?;
        Ok((fc, fr, out, meta))
    }

    /// The whole JPEG branch of `save_processed_multiband_image_sequential` (save.rs:317-367) for u16 GRD bands:
    /// autoscale -> resize -> pad -> synRGB on the device; returns (final_cols, final_rows, rgb, meta).
    pub fn dualpol_synrgb_resized_u16(&self, band1: &[u16], band2: &[u16], rows: usize, cols: usize, strategy: i32,
        mode: i32, target_size: Option<usize>, pad: bool) -> Result<(usize, usize, Vec<u8>, sarpro_hip_resize_meta), HipError> {
        assert_eq!(band1.len(), rows * cols);
        assert_eq!(band2.len(), rows * cols);
        let (mut fc, mut fr) = (0usize, 0usize);
        self.chk(unsafe { sarpro_hip_resize_output_dims(cols, rows, target_size.unwrap_or(0), pad as c_int, &mut fc, &mut fr) })?;
        let mut rgb = vec![0u8; fc * fr * 3];
        let mut meta = sarpro_hip_resize_meta::default();
        self.chk(unsafe { sarpro_hip_dualpol_synrgb_resized_u16(self.ctx, band1.as_ptr(), band2.as_ptr(), rows, cols, strategy,
            mode, target_size.unwrap_or(0), pad as c_int, rgb.as_mut_ptr(), &mut meta) })?;
        Ok((fc, fr, rgb, meta))
    }
}

impl Drop for RasterCore {
    fn drop(&mut self) { unsafe { sarpro_hip_ctx_destroy(self.ctx) } }
}

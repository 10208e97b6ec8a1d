//! Safe surface of `libsarpro_hip.so` with the shapes sarpro's own code uses, so that `src/core/processing` can be
//! swapped for this crate module by module:
//!
//! | sarpro item | here |
//! |---|---|
//! | `types::{AutoscaleStrategy, BitDepth, PolarizationOperation, SyntheticRgbMode, OutputFormat, ProcessingOperation}` (types.rs:8-14,40-45,115-123,162-182) | [`types`] -- same names, same order, `#[repr(i32)]` = the C ABI's discriminants |
//! | `pipeline::process_scalar_data_inplace` (pipeline.rs:8) | [`process_scalar_data_inplace`] |
//! | `pipeline::process_scalar_data_pipeline` (pipeline.rs:42) | [`process_scalar_data_pipeline`] -- 4-tuple; the first element is a [`DbImage`] that derefs to `Array2<f64>` |
//! | `autoscale::autoscale_db_image_tamed_synrgb_u8` (autoscale.rs:710) | [`autoscale_db_image_tamed_synrgb_u8`] -- takes the `DbImage` (it carries the band the dB came from) |
//! | `ops::{sum,difference,ratio,normalized_diff,log_ratio}_arrays` (ops.rs:4-44) | [`ops`] |
//! | `synthetic_rgb::create_synthetic_rgb_by_mode_and_strategy` (synthetic_rgb.rs:182) | [`create_synthetic_rgb_by_mode_and_strategy`] |
//! | `resize::resize_image_data_with_meta` / `resize_image_data` (resize.rs:91,238) | [`resize_image_data_with_meta`], [`resize_image_data`] |
//! | `save::save_processed_image` / `api::save_image` (save.rs:23, api/mod.rs:803) | [`save_image`] + [`render_image`] (the raster the writer receives) |
//! | `save::save_processed_multiband_image_sequential` / `api::save_multiband_image` (save.rs:172, api/mod.rs:826) | [`save_multiband_image`] + [`render_multiband_image`] |
//! | `api::ProcessedImage`, `api::BatchReport` (api/mod.rs:51,453) | [`ProcessedImage`], [`BatchReport`] |
//!
//! The free functions use one context per thread on the device `SARPRO_HIP_DEVICE` names (default 0); [`RasterCore`] is
//! the explicit handle (device-pointer entry points, row stripes, the RCCL communicator).  A GPU failure inside a function
//! whose reference signature is infallible panics with the library's message; every function has a `RasterCore::try_*`
//! twin that returns [`Result`].
#![allow(clippy::too_many_arguments, clippy::type_complexity)]

use ndarray::Array2;
use sarpro_hip_sys as sys;
use std::cell::OnceCell;
use std::ffi::CStr;
use std::os::raw::{c_int, c_uint, c_void};
use std::path::Path;

pub mod types {
    //! `src/types.rs`: the same variants in the same order; `as i32` is what the C ABI takes.

    /// types.rs:8-14
    #[repr(i32)]
    #[derive(Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Debug, Hash)]
    pub enum PolarizationOperation { Sum = 0, Diff = 1, Ratio = 2, NDiff = 3, LogRatio = 4 }

    /// types.rs:115-123
    #[repr(i32)]
    #[derive(Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Debug, Hash)]
    pub enum AutoscaleStrategy { Standard = 0, Robust = 1, Adaptive = 2, Equalized = 3, Clahe = 4, Tamed = 5, Default = 6 }

    /// types.rs:170-173
    #[repr(i32)]
    #[derive(Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Debug, Hash)]
    pub enum BitDepth { U8 = 0, U16 = 1 }

    /// types.rs:177-182
    #[repr(i32)]
    #[derive(Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Debug, Hash)]
    pub enum SyntheticRgbMode { Default = 0, RgbRatio = 1, SarUrban = 2, Enhanced = 3 }

    /// types.rs:162-165
    #[derive(Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Debug, Hash)]
    pub enum OutputFormat { TIFF, JPEG }

    /// types.rs:40-45
    #[derive(Copy, Clone, PartialEq, Eq, Debug)]
    pub enum ProcessingOperation { SingleBand, MultibandVvVh, MultibandHhHv, PolarOp(PolarizationOperation) }

    impl ProcessingOperation {
        /// the label save.rs:34-47 embeds in the metadata
        pub fn label(&self) -> Option<&'static str> {
            match self {
                ProcessingOperation::SingleBand => None,
                ProcessingOperation::MultibandVvVh => Some("multiband_vv_vh"),
                ProcessingOperation::MultibandHhHv => Some("multiband_hh_hv"),
                ProcessingOperation::PolarOp(PolarizationOperation::Sum) => Some("sum"),
                ProcessingOperation::PolarOp(PolarizationOperation::Diff) => Some("difference"),
                ProcessingOperation::PolarOp(PolarizationOperation::Ratio) => Some("ratio"),
                ProcessingOperation::PolarOp(PolarizationOperation::NDiff) => Some("normalized_diff"),
                ProcessingOperation::PolarOp(PolarizationOperation::LogRatio) => Some("log_ratio"),
            }
        }
    }
}
pub use types::{AutoscaleStrategy, BitDepth, OutputFormat, PolarizationOperation, ProcessingOperation, SyntheticRgbMode};

/// What a sarpro integration maps onto `Error::Processing` (src/error.rs:39-46).
#[derive(Debug, Clone)]
pub struct HipError { pub code: i32, pub message: String }
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "sarpro_hip {}: {}", self.code, self.message) }
}
impl std::error::Error for HipError {}
pub type Result<T> = std::result::Result<T, HipError>;

pub use sys::sarpro_hip_f32_partial as F32Partial;
pub use sys::sarpro_hip_resize_meta as ResizeMeta;
pub use sys::sarpro_hip_stats as HistogramStats;

/// `BatchReport` (api/mod.rs:453-458)
#[derive(Debug, Clone, Copy, Default)]
pub struct BatchReport { pub processed: usize, pub skipped: usize, pub errors: usize }

/// `ProcessedImage` (api/mod.rs:51-63) without the SAFE metadata (the reader stays sarpro's)
#[derive(Debug, Clone)]
pub struct ProcessedImage {
    pub width: usize,
    pub height: usize,
    pub bit_depth: BitDepth,
    pub format: OutputFormat,
    pub gray: Option<Vec<u8>>,
    pub gray16: Option<Vec<u16>>,
    pub rgb: Option<Vec<u8>>,
    pub gray_band2: Option<Vec<u8>>,
    pub gray16_band2: Option<Vec<u16>>,
    /// scale and padding bookkeeping of resize.rs:98-108 (save.rs:71-81 turns it into the geotransform override)
    pub resize: ResizeMeta,
}

fn zeroed_stats() -> HistogramStats { unsafe { std::mem::zeroed() } }
fn zeroed_meta() -> ResizeMeta { unsafe { std::mem::zeroed() } }

/// The communicator of the contexts of ONE process (`sarpro_hip_local_group`): threads instead of ranks of an RCCL job.
pub struct LocalGroup { raw: *mut sys::sarpro_hip_local_group }
unsafe impl Send for LocalGroup {}
unsafe impl Sync for LocalGroup {}
impl LocalGroup {
    pub fn new(nranks: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { sys::sarpro_hip_local_group_create(nranks as c_int, &mut raw) };
        if rc != sys::SARPRO_HIP_OK { return Err(HipError { code: rc, message: "sarpro_hip_local_group_create failed".into() }); }
        Ok(Self { raw })
    }
}
impl Drop for LocalGroup {
    fn drop(&mut self) { unsafe { sys::sarpro_hip_local_group_destroy(self.raw) } }
}

/// One library context (one HIP stream, its workspaces).  One per host thread; contexts are independent.
pub struct RasterCore { ctx: *mut sys::sarpro_hip_ctx }
unsafe impl Send for RasterCore {}

impl Drop for RasterCore {
    fn drop(&mut self) { unsafe { sys::sarpro_hip_ctx_destroy(self.ctx) } }
}

impl RasterCore {
    pub fn new(device: i32) -> Result<Self> { Self::with_flags(device, 0) }

    /// flags: `sys::SARPRO_HIP_CTX_TIMING | SARPRO_HIP_CTX_ASYNC_DEV`
    pub fn with_flags(device: i32, flags: c_uint) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::sarpro_hip_ctx_create(device as c_int, flags, &mut ctx) };
        if rc != sys::SARPRO_HIP_OK { return Err(Self::err(std::ptr::null(), rc)); }
        Ok(Self { ctx })
    }

    fn err(ctx: *const sys::sarpro_hip_ctx, code: c_int) -> HipError {
        let p = unsafe { sys::sarpro_hip_last_error(ctx) };
        let message = if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() };
        HipError { code, message }
    }

    fn chk(&self, rc: c_int) -> Result<()> { if rc == sys::SARPRO_HIP_OK { Ok(()) } else { Err(Self::err(self.ctx, rc)) } }

    /// the context's HIP stream (`hipStream_t`); see the stream-ordering contract in include/sarpro_hip.h
    pub fn stream(&self) -> *mut c_void { unsafe { sys::sarpro_hip_ctx_stream(self.ctx) } }
    pub fn synchronize(&self) -> Result<()> { self.chk(unsafe { sys::sarpro_hip_ctx_synchronize(self.ctx) }) }

    // ------------------------------------------------------------------ context attributes (route switches)
    /// A route switch of this context ("NO_SPEC", "SAMPLE_STRIDE", ...: include/sarpro_hip.h, "context attributes").  The environment
    /// variable `SARPRO_HIP_<NAME>` only supplies the attribute's initial value when the context is created; nothing reads it later.
    pub fn set_attr(&self, name: &str, value: i64) -> Result<()> {
        let n = std::ffi::CString::new(name).map_err(|_| HipError { code: sys::SARPRO_HIP_ERR_INVALID_ARG, message: "attribute name holds a NUL".into() })?;
        self.chk(unsafe { sys::sarpro_hip_ctx_set_attr(self.ctx, n.as_ptr(), value) })
    }
    /// back to "unset" (the default route)
    pub fn reset_attr(&self, name: &str) -> Result<()> {
        let n = std::ffi::CString::new(name).map_err(|_| HipError { code: sys::SARPRO_HIP_ERR_INVALID_ARG, message: "attribute name holds a NUL".into() })?;
        self.chk(unsafe { sys::sarpro_hip_ctx_reset_attr(self.ctx, n.as_ptr()) })
    }
    /// `Some(value)` when the attribute is set
    pub fn attr(&self, name: &str) -> Result<Option<i64>> {
        let n = std::ffi::CString::new(name).map_err(|_| HipError { code: sys::SARPRO_HIP_ERR_INVALID_ARG, message: "attribute name holds a NUL".into() })?;
        let (mut v, mut set) = (0i64, 0 as c_int);
        self.chk(unsafe { sys::sarpro_hip_ctx_get_attr(self.ctx, n.as_ptr(), &mut v, &mut set) })?;
        Ok(if set != 0 { Some(v) } else { None })
    }
    /// the names `set_attr` accepts
    pub fn attr_names() -> Vec<String> {
        (0..).map(|i| unsafe { sys::sarpro_hip_attr_name(i) }).take_while(|p| !p.is_null())
            .map(|p| unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()).collect()
    }

    // ------------------------------------------------------------------ pipeline.rs
    /// pipeline.rs:8 -- `(db, valid_mask)`
    pub fn try_process_scalar_data_inplace(&self, processed: &Array2<f32>) -> Result<(Array2<f64>, Vec<bool>)> {
        let (rows, cols) = processed.dim();
        let src = processed.as_standard_layout();
        let mut db = Array2::<f64>::zeros((rows, cols));
        let mut mask = vec![0u8; rows * cols];
        self.chk(unsafe { sys::sarpro_hip_db_mask_f32(self.ctx, src.as_ptr(), rows, cols, db.as_mut_ptr(), mask.as_mut_ptr()) })?;
        Ok((db, mask.into_iter().map(|m| m != 0).collect()))
    }

    fn mask_only(&self, processed: &Array2<f32>) -> Result<Vec<bool>> {
        let (rows, cols) = processed.dim();
        let src = processed.as_standard_layout();
        let mut mask = vec![0u8; rows * cols];
        self.chk(unsafe { sys::sarpro_hip_db_mask_f32(self.ctx, src.as_ptr(), rows, cols, std::ptr::null_mut(), mask.as_mut_ptr()) })?;
        Ok(mask.into_iter().map(|m| m != 0).collect())
    }

    /// pipeline.rs:42 -- `(scaled_u8, scaled_u16, stats)`: the part of the 4-tuple the device computes in one call
    pub fn try_autoscale_band(&self, processed: &Array2<f32>, bit_depth: BitDepth, strategy: AutoscaleStrategy)
        -> Result<(Vec<u8>, Option<Vec<u16>>, HistogramStats)> {
        let (rows, cols) = processed.dim();
        let src = processed.as_standard_layout();
        let mut st = zeroed_stats();
        match bit_depth {
            BitDepth::U8 => {
                let mut out = vec![0u8; rows * cols];
                self.chk(unsafe { sys::sarpro_hip_autoscale_band_f32(self.ctx, src.as_ptr(), rows, cols, strategy as c_int,
                    bit_depth as c_int, out.as_mut_ptr(), std::ptr::null_mut(), &mut st) })?;
                Ok((out, None, st))
            }
            BitDepth::U16 => {
                let mut out = vec![0u16; rows * cols];
                self.chk(unsafe { sys::sarpro_hip_autoscale_band_f32(self.ctx, src.as_ptr(), rows, cols, strategy as c_int,
                    bit_depth as c_int, std::ptr::null_mut(), out.as_mut_ptr(), &mut st) })?;
                Ok((Vec::new(), Some(out), st)) // pipeline.rs:57-66: the u8 vector is empty for U16
            }
        }
    }

    /// the same for a u16 DN band (Sentinel-1 GRD measurement rasters are u16; 2 B/px over PCIe instead of 4)
    pub fn try_autoscale_band_u16(&self, dn: &Array2<u16>, bit_depth: BitDepth, strategy: AutoscaleStrategy)
        -> Result<(Vec<u8>, Option<Vec<u16>>, HistogramStats)> {
        let (rows, cols) = dn.dim();
        let src = dn.as_standard_layout();
        let mut st = zeroed_stats();
        let mut out8 = if bit_depth == BitDepth::U8 { vec![0u8; rows * cols] } else { Vec::new() };
        let mut out16 = if bit_depth == BitDepth::U16 { vec![0u16; rows * cols] } else { Vec::new() };
        self.chk(unsafe { sys::sarpro_hip_autoscale_band_u16(self.ctx, src.as_ptr(), rows, cols, strategy as c_int, bit_depth as c_int,
            if out8.is_empty() { std::ptr::null_mut() } else { out8.as_mut_ptr() },
            if out16.is_empty() { std::ptr::null_mut() } else { out16.as_mut_ptr() }, &mut st) })?;
        Ok((out8, if bit_depth == BitDepth::U16 { Some(out16) } else { None }, st))
    }

    /// autoscale.rs:710 from the band (the device recomputes dB: the u8 raster equals the reference's on `(db, mask)` of
    /// that band)
    pub fn try_tamed_synrgb_u8(&self, band: &Array2<f32>, is_copol: bool) -> Result<Vec<u8>> {
        let (rows, cols) = band.dim();
        let src = band.as_standard_layout();
        let mut out = vec![0u8; rows * cols];
        self.chk(unsafe { sys::sarpro_hip_tamed_synrgb_u8_f32(self.ctx, src.as_ptr(), rows, cols, is_copol as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }

    // ------------------------------------------------------------------ ops.rs
    /// ops.rs:4-44
    pub fn try_polop(&self, op: PolarizationOperation, a: &Array2<f32>, b: &Array2<f32>) -> Result<Array2<f32>> {
        if a.dim() != b.dim() { return Err(HipError { code: sys::SARPRO_HIP_ERR_SHAPE_MISMATCH, message: "band shapes differ".into() }); }
        let (a_, b_) = (a.as_standard_layout(), b.as_standard_layout());
        let mut out = Array2::<f32>::zeros(a.dim());
        self.chk(unsafe { sys::sarpro_hip_polop_f32(self.ctx, op as c_int, a_.as_ptr(), b_.as_ptr(), a.len(), out.as_mut_ptr()) })?;
        Ok(out)
    }

    /// ops.rs:4-44 followed by pipeline.rs:42 on the result (io/sentinel1.rs:1501-1578), fused: the f32 pol-op raster is
    /// never materialised
    pub fn try_polop_autoscale(&self, op: PolarizationOperation, a: &Array2<f32>, b: &Array2<f32>, bit_depth: BitDepth,
        strategy: AutoscaleStrategy) -> Result<(Vec<u8>, Option<Vec<u16>>, HistogramStats)> {
        if a.dim() != b.dim() { return Err(HipError { code: sys::SARPRO_HIP_ERR_SHAPE_MISMATCH, message: "band shapes differ".into() }); }
        let (rows, cols) = a.dim();
        let (a_, b_) = (a.as_standard_layout(), b.as_standard_layout());
        let mut st = zeroed_stats();
        let mut out8 = if bit_depth == BitDepth::U8 { vec![0u8; rows * cols] } else { Vec::new() };
        let mut out16 = if bit_depth == BitDepth::U16 { vec![0u16; rows * cols] } else { Vec::new() };
        self.chk(unsafe { sys::sarpro_hip_polop_autoscale_band_f32(self.ctx, op as c_int, a_.as_ptr(), b_.as_ptr(), rows, cols,
            strategy as c_int, bit_depth as c_int,
            if out8.is_empty() { std::ptr::null_mut() } else { out8.as_mut_ptr() },
            if out16.is_empty() { std::ptr::null_mut() } else { out16.as_mut_ptr() }, &mut st) })?;
        Ok((out8, if bit_depth == BitDepth::U16 { Some(out16) } else { None }, st))
    }

    // ------------------------------------------------------------------ synthetic_rgb.rs
    /// synthetic_rgb.rs:182
    pub fn try_create_synthetic_rgb(&self, mode: SyntheticRgbMode, strategy: AutoscaleStrategy, band1: &[u8], band2: &[u8]) -> Result<Vec<u8>> {
        if band1.len() != band2.len() { return Err(HipError { code: sys::SARPRO_HIP_ERR_SHAPE_MISMATCH, message: "band lengths differ".into() }); }
        let mut rgb = vec![0u8; band1.len() * 3];
        self.chk(unsafe { sys::sarpro_hip_synrgb_u8(self.ctx, mode as c_int, strategy as c_int, band1.as_ptr(), band2.as_ptr(), band1.len(), rgb.as_mut_ptr()) })?;
        Ok(rgb)
    }

    // ------------------------------------------------------------------ resize.rs + padding.rs
    /// resize.rs:91 -- `(final_cols, final_rows, u8, u16, scale_x, scale_y, pad_left, pad_top)`
    pub fn try_resize_image_data_with_meta(&self, u8_data: &[u8], u16_data: Option<&[u16]>, original_cols: usize, original_rows: usize,
        target_size: Option<usize>, bit_depth: BitDepth, pad: bool)
        -> Result<(usize, usize, Vec<u8>, Option<Vec<u16>>, f64, f64, usize, usize)> {
        let (mut fc, mut fr) = (0usize, 0usize);
        let ts = target_size.unwrap_or(0);
        self.chk(unsafe { sys::sarpro_hip_resize_output_dims(original_cols, original_rows, ts, pad as c_int, &mut fc, &mut fr) })?;
        let mut meta = zeroed_meta();
        match (bit_depth, u16_data) {
            (BitDepth::U16, Some(src)) => {
                let mut out = vec![0u16; fc * fr];
                self.chk(unsafe { sys::sarpro_hip_resize_image_data(self.ctx, src.as_ptr() as *const c_void, original_cols, original_rows, ts,
                    bit_depth as c_int, pad as c_int, out.as_mut_ptr() as *mut c_void, &mut meta) })?;
                Ok((fc, fr, Vec::new(), Some(out), meta.scale_x, meta.scale_y, meta.pad_left, meta.pad_top))
            }
            (BitDepth::U16, None) => Err(HipError { code: sys::SARPRO_HIP_ERR_INVALID_ARG, message: "U16 resize without u16 data".into() }),
            (BitDepth::U8, _) => {
                let mut out = vec![0u8; fc * fr];
                self.chk(unsafe { sys::sarpro_hip_resize_image_data(self.ctx, u8_data.as_ptr() as *const c_void, original_cols, original_rows, ts,
                    bit_depth as c_int, pad as c_int, out.as_mut_ptr() as *mut c_void, &mut meta) })?;
                Ok((fc, fr, out, None, meta.scale_x, meta.scale_y, meta.pad_left, meta.pad_top))
            }
        }
    }

    // ------------------------------------------------------------------ save.rs at raster level
    /// save.rs:23-170 up to the writer call: autoscale -> Lanczos3 resize -> pad, all on the device, one band over PCIe
    pub fn try_render_image(&self, processed: &Array2<f32>, format: OutputFormat, bit_depth: BitDepth, target_size: Option<usize>, pad: bool,
        autoscale: AutoscaleStrategy) -> Result<ProcessedImage> {
        let (rows, cols) = processed.dim();
        let bd = if format == OutputFormat::JPEG { BitDepth::U8 } else { bit_depth }; // save.rs:112-113
        let (mut fc, mut fr) = (0usize, 0usize);
        let ts = target_size.unwrap_or(0);
        self.chk(unsafe { sys::sarpro_hip_resize_output_dims(cols, rows, ts, pad as c_int, &mut fc, &mut fr) })?;
        let src = processed.as_standard_layout();
        let mut meta = zeroed_meta();
        let mut img = ProcessedImage { width: fc, height: fr, bit_depth: bd, format, gray: None, gray16: None, rgb: None, gray_band2: None,
            gray16_band2: None, resize: meta };
        match bd {
            BitDepth::U8 => {
                let mut out = vec![0u8; fc * fr];
                self.chk(unsafe { sys::sarpro_hip_process_band_resized_f32(self.ctx, src.as_ptr(), rows, cols, autoscale as c_int, bd as c_int, ts,
                    pad as c_int, out.as_mut_ptr() as *mut c_void, &mut meta) })?;
                img.gray = Some(out);
            }
            BitDepth::U16 => {
                let mut out = vec![0u16; fc * fr];
                self.chk(unsafe { sys::sarpro_hip_process_band_resized_f32(self.ctx, src.as_ptr(), rows, cols, autoscale as c_int, bd as c_int, ts,
                    pad as c_int, out.as_mut_ptr() as *mut c_void, &mut meta) })?;
                img.gray16 = Some(out);
            }
        }
        img.resize = meta;
        Ok(img)
    }

    /// save.rs:172-400 up to the writer call.  TIFF: two autoscaled, resized, padded bands.  JPEG: both pipelines, the Tamed
    /// re-autoscale (save.rs:323-327,345-349), resize, pad and the synthetic RGB composition.
    pub fn try_render_multiband_image(&self, processed1: &Array2<f32>, processed2: &Array2<f32>, format: OutputFormat, bit_depth: BitDepth,
        target_size: Option<usize>, pad: bool, autoscale: AutoscaleStrategy, syn_mode: SyntheticRgbMode) -> Result<ProcessedImage> {
        if processed1.dim() != processed2.dim() {
            return Err(HipError { code: sys::SARPRO_HIP_ERR_SHAPE_MISMATCH, message: "band shapes differ".into() });
        }
        match format {
            OutputFormat::TIFF => {
                let mut a = self.try_render_image(processed1, format, bit_depth, target_size, pad, autoscale)?;
                let b = self.try_render_image(processed2, format, bit_depth, target_size, pad, autoscale)?;
                a.gray_band2 = b.gray;
                a.gray16_band2 = b.gray16;
                Ok(a)
            }
            OutputFormat::JPEG => {
                let (rows, cols) = processed1.dim();
                let (mut fc, mut fr) = (0usize, 0usize);
                let ts = target_size.unwrap_or(0);
                self.chk(unsafe { sys::sarpro_hip_resize_output_dims(cols, rows, ts, pad as c_int, &mut fc, &mut fr) })?;
                let mut meta = zeroed_meta();
                let mut rgb = vec![0u8; fc * fr * 3];
                // one call either way: pipeline x2 (Tamed: band-specific re-autoscale, save.rs:323-327,345-349) -> resize -> pad -> synRGB
                let (b1, b2) = (processed1.as_standard_layout(), processed2.as_standard_layout());
                self.chk(unsafe { sys::sarpro_hip_dualpol_synrgb_resized_f32(self.ctx, b1.as_ptr(), b2.as_ptr(), rows, cols, autoscale as c_int,
                    syn_mode as c_int, 0, ts, pad as c_int, rgb.as_mut_ptr(), &mut meta) })?;
                Ok(ProcessedImage { width: fc, height: fr, bit_depth: BitDepth::U8, format, gray: None, gray16: None, rgb: Some(rgb),
                    gray_band2: None, gray16_band2: None, resize: meta })
            }
        }
    }

    /// process_safe_to_buffer_with_mode, multiband JPEG branch (api/mod.rs:404-437): BOTH bands through process_scalar_data_pipeline with
    /// the caller's strategy (no Tamed re-autoscale), resize, pad, synRGB by mode and strategy -- the reference's default flow, its
    /// bands resampled on read (non-integer f32).
    pub fn try_process_multiband_to_rgb(&self, band1: &Array2<f32>, band2: &Array2<f32>, target_size: Option<usize>, pad: bool,
        autoscale: AutoscaleStrategy, syn_mode: SyntheticRgbMode) -> Result<ProcessedImage> {
        if band1.dim() != band2.dim() {
            return Err(HipError { code: sys::SARPRO_HIP_ERR_SHAPE_MISMATCH, message: "band shapes differ".into() });
        }
        let (rows, cols) = band1.dim();
        let (b1, b2) = (band1.as_standard_layout(), band2.as_standard_layout());
        let (mut fc, mut fr) = (0usize, 0usize);
        let ts = target_size.unwrap_or(0);
        self.chk(unsafe { sys::sarpro_hip_resize_output_dims(cols, rows, ts, pad as c_int, &mut fc, &mut fr) })?;
        let mut rgb = vec![0u8; fc * fr * 3];
        let mut meta = zeroed_meta();
        self.chk(unsafe { sys::sarpro_hip_dualpol_synrgb_resized_f32(self.ctx, b1.as_ptr(), b2.as_ptr(), rows, cols, autoscale as c_int,
            syn_mode as c_int, sys::SARPRO_HIP_DUALPOL_PLAIN_PIPELINE, ts, pad as c_int, rgb.as_mut_ptr(), &mut meta) })?;
        Ok(ProcessedImage { width: fc, height: fr, bit_depth: BitDepth::U8, format: OutputFormat::JPEG, gray: None, gray16: None,
            rgb: Some(rgb), gray_band2: None, gray16_band2: None, resize: meta })
    }

    /// The JPEG branch of save.rs:317-367 for u16 DN bands in ONE call: autoscale x2 -> resize -> pad -> synRGB on the device,
    /// 2 x 2 B/px up, final_cols x final_rows x 3 B down (BASELINE.json configs 1 and 2).
    pub fn try_render_multiband_image_u16(&self, band1: &Array2<u16>, band2: &Array2<u16>, target_size: Option<usize>, pad: bool,
        autoscale: AutoscaleStrategy, syn_mode: SyntheticRgbMode) -> Result<ProcessedImage> {
        if band1.dim() != band2.dim() {
            return Err(HipError { code: sys::SARPRO_HIP_ERR_SHAPE_MISMATCH, message: "band shapes differ".into() });
        }
        let (rows, cols) = band1.dim();
        let (b1, b2) = (band1.as_standard_layout(), band2.as_standard_layout());
        let (mut fc, mut fr) = (0usize, 0usize);
        let ts = target_size.unwrap_or(0);
        self.chk(unsafe { sys::sarpro_hip_resize_output_dims(cols, rows, ts, pad as c_int, &mut fc, &mut fr) })?;
        let mut rgb = vec![0u8; fc * fr * 3];
        let mut meta = zeroed_meta();
        self.chk(unsafe { sys::sarpro_hip_dualpol_synrgb_resized_u16(self.ctx, b1.as_ptr(), b2.as_ptr(), rows, cols, autoscale as c_int,
            syn_mode as c_int, ts, pad as c_int, rgb.as_mut_ptr(), &mut meta) })?;
        Ok(ProcessedImage { width: fc, height: fr, bit_depth: BitDepth::U8, format: OutputFormat::JPEG, gray: None, gray16: None,
            rgb: Some(rgb), gray_band2: None, gray16_band2: None, resize: meta })
    }

    // ------------------------------------------------------------------ device-pointer entry points (rasters resident in HBM)
    /// `d_*` are device addresses; pitches in elements.  See include/sarpro_hip.h for the stream-ordering contract.
    pub unsafe fn dev_dualpol_synrgb_u16(&self, d_band1: *const u16, d_band2: *const u16, rows: usize, cols: usize, in_pitch: usize,
        strategy: AutoscaleStrategy, mode: SyntheticRgbMode, d_rgb: *mut u8, rgb_pitch_px: usize, stats_out: Option<&mut [HistogramStats; 2]>) -> Result<()> {
        let st = stats_out.map_or(std::ptr::null_mut(), |s| s.as_mut_ptr());
        self.chk(sys::sarpro_hip_dualpol_synrgb_u16_dev(self.ctx, d_band1, d_band2, rows, cols, in_pitch, strategy as c_int, mode as c_int,
            d_rgb, rgb_pitch_px, std::ptr::null_mut(), std::ptr::null_mut(), 0, st))
    }
    pub unsafe fn dev_autoscale_band_u16(&self, d_in: *const u16, rows: usize, cols: usize, in_pitch: usize, strategy: AutoscaleStrategy,
        bit_depth: BitDepth, d_out: *mut c_void, out_pitch: usize, stats_out: Option<&mut HistogramStats>) -> Result<()> {
        let st = stats_out.map_or(std::ptr::null_mut(), |s| s as *mut HistogramStats);
        self.chk(sys::sarpro_hip_autoscale_band_u16_dev(self.ctx, d_in, rows, cols, in_pitch, strategy as c_int, bit_depth as c_int, d_out, out_pitch, st))
    }
    pub unsafe fn dev_autoscale_band_f32(&self, d_in: *const f32, rows: usize, cols: usize, in_pitch: usize, strategy: AutoscaleStrategy,
        bit_depth: BitDepth, d_out: *mut c_void, out_pitch: usize, stats_out: Option<&mut HistogramStats>) -> Result<()> {
        let st = stats_out.map_or(std::ptr::null_mut(), |s| s as *mut HistogramStats);
        self.chk(sys::sarpro_hip_autoscale_band_f32_dev(self.ctx, d_in, rows, cols, in_pitch, strategy as c_int, bit_depth as c_int, d_out, out_pitch, st))
    }
    pub unsafe fn dev_polop_autoscale_band_u16(&self, op: PolarizationOperation, d_a: *const u16, d_b: *const u16, rows: usize, cols: usize,
        in_pitch: usize, strategy: AutoscaleStrategy, bit_depth: BitDepth, d_out: *mut c_void, out_pitch: usize, stats_out: Option<&mut HistogramStats>) -> Result<()> {
        let st = stats_out.map_or(std::ptr::null_mut(), |s| s as *mut HistogramStats);
        self.chk(sys::sarpro_hip_polop_autoscale_band_u16_dev(self.ctx, op as c_int, d_a, d_b, rows, cols, in_pitch, strategy as c_int,
            bit_depth as c_int, d_out, out_pitch, st))
    }
    pub unsafe fn dev_polop_autoscale_band_f32(&self, op: PolarizationOperation, d_a: *const f32, d_b: *const f32, rows: usize, cols: usize,
        in_pitch: usize, strategy: AutoscaleStrategy, bit_depth: BitDepth, d_out: *mut c_void, out_pitch: usize, stats_out: Option<&mut HistogramStats>) -> Result<()> {
        let st = stats_out.map_or(std::ptr::null_mut(), |s| s as *mut HistogramStats);
        self.chk(sys::sarpro_hip_polop_autoscale_band_f32_dev(self.ctx, op as c_int, d_a, d_b, rows, cols, in_pitch, strategy as c_int,
            bit_depth as c_int, d_out, out_pitch, st))
    }
    pub unsafe fn dev_polop_f32(&self, op: PolarizationOperation, d_a: *const f32, d_b: *const f32, n: usize, d_out: *mut f32) -> Result<()> {
        self.chk(sys::sarpro_hip_polop_f32_dev(self.ctx, op as c_int, d_a, d_b, n, d_out))
    }
    pub unsafe fn dev_synrgb_u8(&self, mode: SyntheticRgbMode, strategy: AutoscaleStrategy, d_band1: *const u8, d_band2: *const u8, n: usize,
        d_rgb: *mut u8) -> Result<()> {
        self.chk(sys::sarpro_hip_synrgb_u8_dev(self.ctx, mode as c_int, strategy as c_int, d_band1, d_band2, n, d_rgb))
    }
    pub unsafe fn dev_resize_image_data(&self, d_data: *const c_void, cols: usize, rows: usize, pitch: usize, target_size: Option<usize>,
        bit_depth: BitDepth, pad: bool, d_out: *mut c_void, out_pitch: usize) -> Result<ResizeMeta> {
        let mut meta = zeroed_meta();
        self.chk(sys::sarpro_hip_resize_image_data_dev(self.ctx, d_data, cols, rows, pitch, target_size.unwrap_or(0), bit_depth as c_int,
            pad as c_int, d_out, out_pitch, &mut meta))?;
        Ok(meta)
    }

    // ------------------------------------------------------------------ RCCL communicator + row stripes
    /// rank 0 makes the id, every rank receives it out of band (MPI, a file, torch.distributed ...)
    pub fn comm_unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        let rc = unsafe { sys::sarpro_hip_comm_unique_id(id.as_mut_ptr()) };
        if rc != sys::SARPRO_HIP_OK { return Err(Self::err(std::ptr::null(), rc)); }
        Ok(id)
    }
    pub fn comm_init(&self, nranks: i32, rank: i32, uid: &[u8; 128]) -> Result<()> {
        self.chk(unsafe { sys::sarpro_hip_comm_init(self.ctx, nranks as c_int, rank as c_int, uid.as_ptr()) })
    }
    pub unsafe fn comm_allreduce_sum_u64(&self, d_buf: *mut u64, count: usize) -> Result<()> {
        self.chk(sys::sarpro_hip_comm_allreduce_sum_u64(self.ctx, d_buf, count))
    }
    pub fn comm_destroy(&self) { unsafe { sys::sarpro_hip_comm_destroy(self.ctx) } }
    /// Join the in-process communicator `group` as `rank`: one `RasterCore` and one thread per rank, no RCCL (the ranks' device
    /// buffers must be mutually accessible: one device, or peer access enabled).  The group must outlive the contexts that joined it.
    pub fn comm_init_local(&self, group: &LocalGroup, rank: i32) -> Result<()> {
        self.chk(unsafe { sys::sarpro_hip_comm_init_local(self.ctx, group.raw, rank as c_int) })
    }

    /// tile-aligned row stripes of a `rows`-row scene for `nranks` ranks: `(row0, nrows)` per rank
    pub fn stripe_plan(rows: usize, nranks: usize) -> Result<Vec<(usize, usize)>> {
        let (mut r0, mut nr) = (vec![0usize; nranks], vec![0usize; nranks]);
        let rc = unsafe { sys::sarpro_hip_host_stripe_plan(rows, nranks as c_int, r0.as_mut_ptr(), nr.as_mut_ptr()) };
        if rc != sys::SARPRO_HIP_OK { return Err(Self::err(std::ptr::null(), rc)); }
        Ok(r0.into_iter().zip(nr).collect())
    }

    /// this rank's stripe of a dual-pol u16 scene -> its stripe of the RGB raster, reductions over the library's communicator
    pub unsafe fn stripe_run_u16(&self, d_band1: *const u16, d_band2: *const u16, rows_total: usize, cols: usize, row0: usize, rows_local: usize,
        in_pitch: usize, strategy: AutoscaleStrategy, mode: SyntheticRgbMode, d_rgb: *mut u8, rgb_pitch_px: usize) -> Result<[HistogramStats; 2]> {
        let mut st = [zeroed_stats(), zeroed_stats()];
        self.chk(sys::sarpro_hip_stripe_run_u16(self.ctx, d_band1, d_band2, rows_total, cols, row0, rows_local, in_pitch, strategy as c_int,
            mode as c_int, d_rgb, rgb_pitch_px, st.as_mut_ptr()))?;
        Ok(st)
    }
    /// this rank's stripe of a dual-pol u16 scene -> its rows of the RESIZED, padded RGB product (save.rs:317-367 over row stripes: the
    /// rows the vertical Lanczos windows need from the neighbouring ranks travel in one small all-reduce).  `d_rgb_slice` receives
    /// `out_rows * final_cols * 3` bytes (size it with [`stripe_resized_rows`]); returns (out_row0, out_rows, meta).
    pub unsafe fn stripe_run_resized_u16(&self, d_band1: *const u16, d_band2: *const u16, rows_total: usize, cols: usize, row0: usize,
        rows_local: usize, in_pitch: usize, strategy: AutoscaleStrategy, mode: SyntheticRgbMode, target_size: Option<usize>, pad: bool,
        d_rgb_slice: *mut u8) -> Result<(usize, usize, ResizeMeta)> {
        let (mut r0, mut n, mut m) = (0usize, 0usize, zeroed_meta());
        self.chk(sys::sarpro_hip_stripe_run_resized_u16(self.ctx, d_band1, d_band2, rows_total, cols, row0, rows_local, in_pitch, strategy as c_int,
            mode as c_int, target_size.unwrap_or(0), pad as c_int, d_rgb_slice, &mut r0, &mut n, &mut m))?;
        Ok((r0, n, m))
    }
    /// [`stripe_run_resized_u16`] for f32 bands (the reference's resampled-on-read flow); Tamed needs `plain_pipeline`
    pub unsafe fn stripe_run_resized_f32(&self, d_band1: *const f32, d_band2: *const f32, rows_total: usize, cols: usize, row0: usize,
        rows_local: usize, in_pitch: usize, strategy: AutoscaleStrategy, mode: SyntheticRgbMode, plain_pipeline: bool, target_size: Option<usize>,
        pad: bool, d_rgb_slice: *mut u8) -> Result<(usize, usize, ResizeMeta)> {
        let (mut r0, mut n, mut m) = (0usize, 0usize, zeroed_meta());
        self.chk(sys::sarpro_hip_stripe_run_resized_f32(self.ctx, d_band1, d_band2, rows_total, cols, row0, rows_local, in_pitch, strategy as c_int,
            mode as c_int, if plain_pipeline { sys::SARPRO_HIP_DUALPOL_PLAIN_PIPELINE } else { 0 }, target_size.unwrap_or(0), pad as c_int, d_rgb_slice,
            &mut r0, &mut n, &mut m))?;
        Ok((r0, n, m))
    }
    /// this rank's stripe of an f32 band -> its stripe of the level raster
    pub unsafe fn stripe_run_f32(&self, d_in: *const f32, rows_total: usize, cols: usize, row0: usize, rows_local: usize, in_pitch: usize,
        strategy: AutoscaleStrategy, bit_depth: BitDepth, d_out: *mut c_void, out_pitch: usize) -> Result<HistogramStats> {
        let mut st = zeroed_stats();
        self.chk(sys::sarpro_hip_stripe_run_f32(self.ctx, d_in, rows_total, cols, row0, rows_local, in_pitch, strategy as c_int,
            bit_depth as c_int, d_out, out_pitch, &mut st))?;
        Ok(st)
    }
    /// this rank's stripes of two bands -> its stripe of the autoscaled pol-op raster
    pub unsafe fn stripe_run_polop(&self, op: PolarizationOperation, d_a: *const c_void, d_b: *const c_void, elem_u16: bool, rows_total: usize,
        cols: usize, row0: usize, rows_local: usize, in_pitch: usize, strategy: AutoscaleStrategy, bit_depth: BitDepth, d_out: *mut c_void,
        out_pitch: usize) -> Result<HistogramStats> {
        let mut st = zeroed_stats();
        self.chk(sys::sarpro_hip_stripe_run_polop(self.ctx, op as c_int, d_a, d_b, elem_u16 as c_int, rows_total, cols, row0, rows_local,
            in_pitch, strategy as c_int, bit_depth as c_int, d_out, out_pitch, &mut st))?;
        Ok(st)
    }

    /// phase-by-phase stripe for callers that reduce with their own communicator
    pub unsafe fn stripe_begin_u16(&self, d_band1: *const u16, d_band2: *const u16, rows_total: usize, cols: usize, row0: usize,
        rows_local: usize, in_pitch: usize, strategy: AutoscaleStrategy, mode: SyntheticRgbMode) -> Result<Stripe<'_>> {
        let mut h = std::ptr::null_mut();
        self.chk(sys::sarpro_hip_stripe_begin_u16(self.ctx, d_band1, d_band2, rows_total, cols, row0, rows_local, in_pitch, strategy as c_int,
            mode as c_int, &mut h))?;
        Ok(Stripe { core: self, h })
    }
    pub unsafe fn stripe_begin_f32(&self, d_in: *const f32, rows_total: usize, cols: usize, row0: usize, rows_local: usize, in_pitch: usize,
        strategy: AutoscaleStrategy, bit_depth: BitDepth, d_out: *mut c_void, out_pitch: usize) -> Result<StripeF32<'_>> {
        let mut h = std::ptr::null_mut();
        self.chk(sys::sarpro_hip_stripe_begin_f32(self.ctx, d_in, rows_total, cols, row0, rows_local, in_pitch, strategy as c_int,
            bit_depth as c_int, d_out, out_pitch, &mut h))?;
        Ok(StripeF32 { core: self, h })
    }
    pub unsafe fn stripe_begin_polop(&self, op: PolarizationOperation, d_a: *const c_void, d_b: *const c_void, elem_u16: bool,
        rows_total: usize, cols: usize, row0: usize, rows_local: usize, in_pitch: usize, strategy: AutoscaleStrategy, bit_depth: BitDepth,
        d_out: *mut c_void, out_pitch: usize) -> Result<StripeF32<'_>> {
        let mut h = std::ptr::null_mut();
        self.chk(sys::sarpro_hip_stripe_begin_polop(self.ctx, op as c_int, d_a, d_b, elem_u16 as c_int, rows_total, cols, row0, rows_local,
            in_pitch, strategy as c_int, bit_depth as c_int, d_out, out_pitch, &mut h))?;
        Ok(StripeF32 { core: self, h })
    }
}

/// `sarpro_hip_stripe`: each phase returns the device buffer `(ptr, count)` of u64 the caller all-reduces (sum) before the next
pub struct Stripe<'a> { core: &'a RasterCore, h: *mut sys::sarpro_hip_stripe }
impl Stripe<'_> {
    fn phase(&mut self, f: unsafe extern "C" fn(*mut sys::sarpro_hip_stripe, *mut *mut u64, *mut usize) -> c_int) -> Result<(*mut u64, usize)> {
        let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
        self.core.chk(unsafe { f(self.h, &mut p, &mut n) })?;
        Ok((p, n))
    }
    pub fn phase1(&mut self) -> Result<(*mut u64, usize)> { self.phase(sys::sarpro_hip_stripe_phase1) }
    pub fn phase2(&mut self) -> Result<(*mut u64, usize)> { self.phase(sys::sarpro_hip_stripe_phase2) }
    pub fn phase3(&mut self) -> Result<(*mut u64, usize)> { self.phase(sys::sarpro_hip_stripe_phase3) }
    pub unsafe fn phase4(&mut self, d_rgb: *mut u8, rgb_pitch_px: usize) -> Result<[HistogramStats; 2]> {
        let mut st = [zeroed_stats(), zeroed_stats()];
        self.core.chk(sys::sarpro_hip_stripe_phase4(self.h, d_rgb, rgb_pitch_px, st.as_mut_ptr()))?;
        Ok(st)
    }
}
impl Drop for Stripe<'_> {
    fn drop(&mut self) { unsafe { sys::sarpro_hip_stripe_end(self.h) } }
}

/// `sarpro_hip_stripe_f32`: count / min / max first (gathered on the host), then three u64 histograms
pub struct StripeF32<'a> { core: &'a RasterCore, h: *mut sys::sarpro_hip_stripe_f32 }
impl StripeF32<'_> {
    pub fn phase1(&mut self) -> Result<F32Partial> {
        let mut p: F32Partial = unsafe { std::mem::zeroed() };
        self.core.chk(unsafe { sys::sarpro_hip_stripe_f32_phase1(self.h, &mut p) })?;
        Ok(p)
    }
    /// the merge of all ranks' partials, in rank order
    pub fn merge(parts: &[F32Partial]) -> F32Partial {
        let mut g: F32Partial = unsafe { std::mem::zeroed() };
        unsafe { sys::sarpro_hip_host_f32_merge_partials(parts.as_ptr(), parts.len(), &mut g) };
        g
    }
    pub fn phase2(&mut self, global: &F32Partial) -> Result<(*mut u64, usize)> {
        let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
        self.core.chk(unsafe { sys::sarpro_hip_stripe_f32_phase2(self.h, global, &mut p, &mut n) })?;
        Ok((p, n))
    }
    pub fn phase3(&mut self) -> Result<(*mut u64, usize)> {
        let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
        self.core.chk(unsafe { sys::sarpro_hip_stripe_f32_phase3(self.h, &mut p, &mut n) })?;
        Ok((p, n))
    }
    pub fn phase4(&mut self) -> Result<(*mut u64, usize)> {
        let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
        self.core.chk(unsafe { sys::sarpro_hip_stripe_f32_phase4(self.h, &mut p, &mut n) })?;
        Ok((p, n))
    }
    pub fn phase5(&mut self) -> Result<HistogramStats> {
        let mut st = zeroed_stats();
        self.core.chk(unsafe { sys::sarpro_hip_stripe_f32_phase5(self.h, &mut st) })?;
        Ok(st)
    }
}
impl Drop for StripeF32<'_> {
    fn drop(&mut self) { unsafe { sys::sarpro_hip_stripe_f32_end(self.h) } }
}

// ---------------------------------------------------------------------------------------------------------------------
// Free functions with the reference's signatures, on a per-thread context.
thread_local! {
    static CORE: RasterCore = {
        let dev = std::env::var("SARPRO_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        RasterCore::new(dev).unwrap_or_else(|e| panic!("sarpro_hip: no context on device {dev}: {e}"))
    };
}

/// Runs `f` on this thread's context.
pub fn with_core<T>(f: impl FnOnce(&RasterCore) -> T) -> T { CORE.with(|c| f(c)) }

/// The dB image of one band (pipeline.rs:8-36).  Derefs to `Array2<f64>`, so `db.dim()`, `&*db`, `db[[r, c]]` read as
/// before; the values are filled on first dereference unless the crate is built with `eager-db` (save.rs only reads the
/// dims: 8 B/px over PCIe that nobody looks at).  It keeps the band it came from, which is what
/// [`autoscale_db_image_tamed_synrgb_u8`] hands to the device (dB -> linear is not invertible bit for bit).
pub struct DbImage { band: Array2<f32>, db: OnceCell<Array2<f64>> }

impl DbImage {
    pub fn band(&self) -> &Array2<f32> { &self.band }
    pub fn dim(&self) -> (usize, usize) { self.band.dim() }
    pub fn into_array(self) -> Array2<f64> { let _ = &*self; self.db.into_inner().expect("filled above") }
}

impl std::ops::Deref for DbImage {
    type Target = Array2<f64>;
    fn deref(&self) -> &Array2<f64> {
        self.db.get_or_init(|| with_core(|c| c.try_process_scalar_data_inplace(&self.band)).unwrap_or_else(|e| panic!("{e}")).0)
    }
}

/// pipeline.rs:8
pub fn process_scalar_data_inplace(processed: &Array2<f32>) -> (Array2<f64>, Vec<bool>) {
    with_core(|c| c.try_process_scalar_data_inplace(processed)).unwrap_or_else(|e| panic!("{e}"))
}

/// pipeline.rs:42-67: `(db, valid_mask, scaled_u8, scaled_u16)`
pub fn process_scalar_data_pipeline(processed: &Array2<f32>, bit_depth: BitDepth, strategy: AutoscaleStrategy)
    -> (DbImage, Vec<bool>, Vec<u8>, Option<Vec<u16>>) {
    with_core(|c| {
        let (u8s, u16s, _) = c.try_autoscale_band(processed, bit_depth, strategy).unwrap_or_else(|e| panic!("{e}"));
        let db = DbImage { band: processed.clone(), db: OnceCell::new() };
        let mask = if cfg!(feature = "eager-db") {
            let (d, m) = c.try_process_scalar_data_inplace(processed).unwrap_or_else(|e| panic!("{e}"));
            let _ = db.db.set(d);
            m
        } else {
            c.mask_only(processed).unwrap_or_else(|e| panic!("{e}"))
        };
        (db, mask, u8s, u16s)
    })
}

/// autoscale.rs:710.  `valid_mask` is the mask `process_scalar_data_pipeline` returned with `db` (the device derives the
/// same mask from the band; a different mask is not supported and panics in debug builds).
pub fn autoscale_db_image_tamed_synrgb_u8(db: &DbImage, valid_mask: &[bool], is_copol: bool) -> Vec<u8> {
    debug_assert_eq!(valid_mask.len(), db.band.len());
    with_core(|c| c.try_tamed_synrgb_u8(&db.band, is_copol)).unwrap_or_else(|e| panic!("{e}"))
}

pub mod ops {
    //! ops.rs:4-44
    use super::{with_core, Array2, PolarizationOperation as Op};
    fn run(op: Op, a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32> { with_core(|c| c.try_polop(op, a, b)).unwrap_or_else(|e| panic!("{e}")) }
    pub fn sum_arrays(a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32> { run(Op::Sum, a, b) }
    pub fn difference_arrays(a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32> { run(Op::Diff, a, b) }
    pub fn ratio_arrays(a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32> { run(Op::Ratio, a, b) }
    pub fn normalized_diff_arrays(a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32> { run(Op::NDiff, a, b) }
    pub fn log_ratio_arrays(a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32> { run(Op::LogRatio, a, b) }
}

/// synthetic_rgb.rs:182
pub fn create_synthetic_rgb_by_mode_and_strategy(mode: SyntheticRgbMode, strategy: AutoscaleStrategy, band1_data: &[u8], band2_data: &[u8]) -> Vec<u8> {
    with_core(|c| c.try_create_synthetic_rgb(mode, strategy, band1_data, band2_data)).unwrap_or_else(|e| panic!("{e}"))
}

/// resize.rs:91
pub fn resize_image_data_with_meta(u8_data: &[u8], u16_data: Option<&[u16]>, original_cols: usize, original_rows: usize,
    target_size: Option<usize>, bit_depth: BitDepth, pad: bool)
    -> std::result::Result<(usize, usize, Vec<u8>, Option<Vec<u16>>, f64, f64, usize, usize), Box<dyn std::error::Error>> {
    Ok(with_core(|c| c.try_resize_image_data_with_meta(u8_data, u16_data, original_cols, original_rows, target_size, bit_depth, pad))?)
}

/// resize.rs:238
pub fn resize_image_data(u8_data: &[u8], u16_data: Option<&[u16]>, original_cols: usize, original_rows: usize, target_size: Option<usize>,
    bit_depth: BitDepth, pad: bool) -> std::result::Result<(usize, usize, Vec<u8>, Option<Vec<u16>>), Box<dyn std::error::Error>> {
    let r = resize_image_data_with_meta(u8_data, u16_data, original_cols, original_rows, target_size, bit_depth, pad)?;
    Ok((r.0, r.1, r.2, r.3))
}

/// save.rs:71-81 / 141-151: the geotransform of the resized, padded product
/// Which rows of the resized, padded product the holder of input rows [row0, row0 + rows_local) produces:
/// (out_row0, out_rows, final_cols, final_rows) -- pure host arithmetic, the same answer on every rank.
pub fn stripe_resized_rows(rows_total: usize, cols: usize, row0: usize, rows_local: usize, target_size: Option<usize>, pad: bool)
    -> Option<(usize, usize, usize, usize)> {
    let (mut a, mut b, mut c, mut d) = (0usize, 0usize, 0usize, 0usize);
    let rc = unsafe { sys::sarpro_hip_stripe_resized_rows(rows_total, cols, row0, rows_local, target_size.unwrap_or(0), pad as c_int, &mut a, &mut b, &mut c, &mut d) };
    if rc == 0 { Some((a, b, c, d)) } else { None }
}

pub fn update_geotransform(gt: &mut [f64; 6], cols: usize, rows: usize, meta: &ResizeMeta) {
    unsafe { sys::sarpro_hip_host_update_geotransform(gt.as_mut_ptr(), cols, rows, meta) }
}

/// The raster `save_image` hands to its writer (save.rs:23-170 without the encoder).
pub fn render_image(processed: &Array2<f32>, format: OutputFormat, bit_depth: BitDepth, target_size: Option<usize>, pad: bool,
    autoscale: AutoscaleStrategy) -> Result<ProcessedImage> {
    with_core(|c| c.try_render_image(processed, format, bit_depth, target_size, pad, autoscale))
}

/// The raster(s) `save_multiband_image` hands to its writer (save.rs:172-400 without the encoder).
pub fn render_multiband_image(processed1: &Array2<f32>, processed2: &Array2<f32>, format: OutputFormat, bit_depth: BitDepth,
    target_size: Option<usize>, pad: bool, autoscale: AutoscaleStrategy, syn_mode: SyntheticRgbMode) -> Result<ProcessedImage> {
    with_core(|c| c.try_render_multiband_image(processed1, processed2, format, bit_depth, target_size, pad, autoscale, syn_mode))
}

/// The encoders stay sarpro's (io/writers/*: GDAL GeoTIFF, JPEG + sidecars): implement this for them.  `metadata` is the
/// caller's `SafeMetadata`, passed through untouched; `operation` carries the label save.rs:34-47 embeds.
pub trait RasterWriter<M: ?Sized> {
    fn write(&mut self, output: &Path, image: &ProcessedImage, metadata: Option<&M>, operation: ProcessingOperation)
        -> std::result::Result<(), Box<dyn std::error::Error>>;
}

/// api/mod.rs:803-824 with the writer as the last argument.
pub fn save_image<M: ?Sized, W: RasterWriter<M>>(processed: &Array2<f32>, output: &Path, format: OutputFormat, bit_depth: BitDepth,
    target_size: Option<usize>, metadata: Option<&M>, pad: bool, autoscale: AutoscaleStrategy, operation: ProcessingOperation, writer: &mut W)
    -> std::result::Result<(), Box<dyn std::error::Error>> {
    let image = render_image(processed, format, bit_depth, target_size, pad, autoscale)?;
    writer.write(output, &image, metadata, operation)
}

/// api/mod.rs:826-857 with the writer as the last argument (`SyntheticRgbMode::Default`, as there).
pub fn save_multiband_image<M: ?Sized, W: RasterWriter<M>>(processed1: &Array2<f32>, processed2: &Array2<f32>, output: &Path,
    format: OutputFormat, bit_depth: BitDepth, target_size: Option<usize>, metadata: Option<&M>, pad: bool, autoscale: AutoscaleStrategy,
    operation: ProcessingOperation, writer: &mut W) -> std::result::Result<(), Box<dyn std::error::Error>> {
    let image = render_multiband_image(processed1, processed2, format, bit_depth, target_size, pad, autoscale, SyntheticRgbMode::Default)?;
    writer.write(output, &image, metadata, operation)
}

/// A writer that needs no GDAL: uncompressed strip (Big)TIFF through the library's own writer (gray, 2-band, or RGB).
pub struct PlainTiffWriter { pub geotransform: Option<[f64; 6]> }

impl<M: ?Sized> RasterWriter<M> for PlainTiffWriter {
    fn write(&mut self, output: &Path, image: &ProcessedImage, _metadata: Option<&M>, _operation: ProcessingOperation)
        -> std::result::Result<(), Box<dyn std::error::Error>> {
        let path = std::ffi::CString::new(output.to_string_lossy().as_bytes())?;
        let (w, h) = (image.width, image.height);
        let (samples, bits, data): (u32, u32, Vec<u8>) = if let Some(rgb) = &image.rgb {
            (3, 8, rgb.clone())
        } else if let (Some(a), Some(b)) = (&image.gray, &image.gray_band2) {
            (2, 8, a.iter().zip(b).flat_map(|(x, y)| [*x, *y]).collect())
        } else if let (Some(a), Some(b)) = (&image.gray16, &image.gray16_band2) {
            (2, 16, a.iter().zip(b).flat_map(|(x, y)| { let (p, q) = (x.to_ne_bytes(), y.to_ne_bytes()); [p[0], p[1], q[0], q[1]] }).collect())
        } else if let Some(a) = &image.gray16 {
            (1, 16, a.iter().flat_map(|x| x.to_ne_bytes()).collect())
        } else if let Some(a) = &image.gray {
            (1, 8, a.clone())
        } else {
            return Err("empty image".into());
        };
        let gt = self.geotransform.map(|mut g| { update_geotransform(&mut g, w, h, &image.resize); g });
        let mut tw = std::ptr::null_mut();
        let fail = || -> Box<dyn std::error::Error> {
            let p = unsafe { sys::sarpro_hip_tiff_last_error() };
            (if p.is_null() { "tiff error".to_string() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }).into()
        };
        let rc = unsafe { sys::sarpro_hip_tiff_create(path.as_ptr(), w as u64, h as u64, samples, bits,
            gt.as_ref().map_or(std::ptr::null(), |g| g.as_ptr()), std::ptr::null(), &mut tw) };
        if rc != sys::SARPRO_HIP_OK { return Err(fail()); }
        let row_bytes = w * samples as usize * (bits as usize / 8);
        let rc = unsafe { sys::sarpro_hip_tiff_write_rows(tw, 0, h, data.as_ptr() as *const c_void, row_bytes) };
        let rc2 = unsafe { sys::sarpro_hip_tiff_finish(tw) };
        if rc != sys::SARPRO_HIP_OK || rc2 != sys::SARPRO_HIP_OK { return Err(fail()); }
        Ok(())
    }
}

/// One scene of a resident batch: device pointers of its two u16 bands and of its RGB raster (`RasterCore::try_batch_dualpol_synrgb_u16_dev`).
#[derive(Clone, Copy, Debug)]
pub struct ResidentScene { pub d_band1: *const u16, pub d_band2: *const u16, pub d_rgb: *mut u8 }
/// What a scene of a resident batch did: the fused CLAHE -> RGB pass's raster stood, or the exact kernels produced it.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum SceneRoute { NotSpeculative, Accepted, Refuted, Unproven, PoolOverflow }

impl RasterCore {
    /// The batch loop of api/mod.rs:484-533 for scenes that are already in HBM (all of one shape): save.rs:317-367 at native
    /// resolution per scene, pipelined over `lanes` internal streams of this context (0 = the library's default).  Returns when every
    /// raster is complete, with the report and each scene's (status, route).  The rasters are those of one
    /// `dev_dualpol_synrgb_u16` call per scene, bit for bit.
    pub fn try_batch_dualpol_synrgb_u16_dev(&self, scenes: &[ResidentScene], rows: usize, cols: usize, in_pitch: usize, strategy: AutoscaleStrategy,
        mode: SyntheticRgbMode, rgb_pitch_px: usize, lanes: usize, continue_on_error: bool) -> Result<(BatchReport, Vec<(i32, SceneRoute)>)> {
        let mut descs: Vec<sys::sarpro_hip_resident_scene> = scenes.iter().map(|s| sys::sarpro_hip_resident_scene {
            d_band1: s.d_band1, d_band2: s.d_band2, d_rgb: s.d_rgb, status: 0, route: sys::SARPRO_HIP_ROUTE_NONE }).collect();
        let mut rep = sys::sarpro_hip_batch_report { processed: 0, skipped: 0, errors: 0 };
        let rc = unsafe { sys::sarpro_hip_batch_dualpol_synrgb_u16_dev(self.ctx, descs.as_mut_ptr(), descs.len(), rows, cols, in_pitch,
            strategy as c_int, mode as c_int, rgb_pitch_px, lanes as c_int, continue_on_error as c_int, &mut rep) };
        if rc != sys::SARPRO_HIP_OK && !continue_on_error { return Err(Self::err(self.ctx, rc)); }
        let route = |r: c_int| match r { 0 => SceneRoute::Accepted, 1 => SceneRoute::Refuted, 2 => SceneRoute::Unproven, 3 => SceneRoute::PoolOverflow, _ => SceneRoute::NotSpeculative };
        Ok((BatchReport { processed: rep.processed, skipped: rep.skipped, errors: rep.errors }, descs.iter().map(|d| (d.status as i32, route(d.route))).collect()))
    }
}

/// `process_directory_to_path`'s hot loop (api/mod.rs:474-536) for scenes already decoded to u16 DN: `workers_per_device` host
/// threads (0 = the library's default, 2: scene i's download beside scene i + 1's upload), each with a context of its own, per listed
/// device; scenes dealt dynamically.  `out[i]` receives scene i's RGB (`final_rows * final_cols * 3`).
pub fn batch_dualpol_synrgb_resized_u16(devices: &[i32], workers_per_device: usize, scenes: &[(&Array2<u16>, &Array2<u16>)], strategy: AutoscaleStrategy,
    mode: SyntheticRgbMode, target_size: Option<usize>, pad: bool, continue_on_error: bool, out: &mut [Vec<u8>]) -> Result<(BatchReport, Vec<i32>)> {
    assert_eq!(scenes.len(), out.len());
    let ts = target_size.unwrap_or(0);
    let mut status = vec![0 as c_int; scenes.len()];
    let std_bands: Vec<_> = scenes.iter().map(|(a, b)| (a.as_standard_layout(), b.as_standard_layout())).collect();
    let mut descs = Vec::with_capacity(scenes.len());
    for (i, (a, b)) in std_bands.iter().enumerate() {
        let (rows, cols) = a.dim();
        let (mut fc, mut fr) = (0usize, 0usize);
        unsafe { sys::sarpro_hip_resize_output_dims(cols, rows, ts, pad as c_int, &mut fc, &mut fr) };
        out[i].resize(fc * fr * 3, 0);
        descs.push(sys::sarpro_hip_batch_scene { band1: a.as_ptr(), band2: b.as_ptr(), rows, cols, rgb_out: out[i].as_mut_ptr(),
            status_out: &mut status[i], reader: None, reader_user: std::ptr::null_mut() });
    }
    let devs: Vec<c_int> = devices.iter().map(|d| *d as c_int).collect();
    let mut rep = sys::sarpro_hip_batch_report { processed: 0, skipped: 0, errors: 0 };
    let rc = unsafe { sys::sarpro_hip_batch_dualpol_synrgb_resized_u16(devs.as_ptr(), devs.len() as c_int, workers_per_device as c_int, descs.as_ptr(), descs.len(),
        strategy as c_int, mode as c_int, ts, pad as c_int, continue_on_error as c_int, &mut rep) };
    let report = BatchReport { processed: rep.processed, skipped: rep.skipped, errors: rep.errors };
    if rc != sys::SARPRO_HIP_OK && !continue_on_error { return Err(HipError { code: rc, message: "batch aborted on the first failing scene".into() }); }
    Ok((report, status.into_iter().map(|s| s as i32).collect()))
}

/*
 * sarpro_hip.h -- C ABI of libsarpro_hip.so: the MI355X (gfx950) replacement for
 * sarpro's per-pixel raster core (reference: bogwi/sarpro v0.3.0, src/core/processing).
 *
 * The reference has no FFI layer: its seam is ordinary Rust calls from save.rs /
 * api/mod.rs / io/sentinel1.rs into core::processing::{pipeline,autoscale,ops,
 * synthetic_rgb}.  Each entry point below names the Rust function it replaces
 * (file:line in the reference tree) -- a Rust `extern "C"` crate binding these
 * (rust/sarpro-hip-sys) gives save.rs the same call surface.
 *
 * Conventions
 *   - every call returns a status (0 = ok, <0 = error); message via sarpro_hip_last_error
 *   - rasters are row-major (rows, cols), the layout of ndarray::Array2 / GDAL read_as
 *     (io/gdal.rs:123-131); all sizes in ELEMENTS; outputs are caller-allocated
 *   - enum integers are the declaration order of src/types.rs
 *   - host entry points take HOST pointers (pageable or pinned) and are synchronous;
 *     `_dev` entry points take DEVICE pointers (+ pitch in elements) and run on the
 *     context's stream; they return after the result is complete on the device
 *   - STREAM ORDERING of `_dev` entry points: they run on the context's own hipStreamNonBlocking stream
 *     (sarpro_hip_ctx_stream), which is NOT ordered against any other stream, the NULL stream included.  The caller
 *     must order that stream behind the producers of its input rasters and behind earlier users of its output
 *     rasters -- hipStreamWaitEvent(sarpro_hip_ctx_stream(ctx), event recorded on the producer's stream), or a
 *     device / stream synchronisation before the call -- and order consumers behind the context's stream the same
 *     way unless the call was synchronous (every call is, except on a SARPRO_HIP_CTX_ASYNC_DEV context)
 *   - one context per host thread; contexts are independent (no global state)
 *   - nothing here falls back to a CPU implementation: without a usable HIP device
 *     ctx_create fails with SARPRO_HIP_ERR_NO_DEVICE
 *   - `sarpro_hip_host_*` functions are the host half of the path (statistics, LUT and
 *     CDF construction from device-produced histograms); they need no GPU and are
 *     exported so the multi-GPU driver and the tests can call them directly
 */
#ifndef SARPRO_HIP_H
#define SARPRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (reference: infallible fns; callers wrap panics/errors as
 *      Error::External / Error::Processing, src/error.rs:39-46) ---- */
#define SARPRO_HIP_OK 0
#define SARPRO_HIP_ERR_INVALID_ARG (-1)
#define SARPRO_HIP_ERR_SHAPE_MISMATCH (-2)
#define SARPRO_HIP_ERR_UNSUPPORTED_SHAPE (-3) /* CLAHE tile underflow sizes where the reference panics (autoscale.rs:250,254) */
#define SARPRO_HIP_ERR_HIP (-4)
#define SARPRO_HIP_ERR_RCCL (-5)
#define SARPRO_HIP_ERR_OOM (-6)
#define SARPRO_HIP_ERR_NO_DEVICE (-7)
#define SARPRO_HIP_ERR_IO (-8) /* a row reader / sink / TIFF file failed */

/* ---- src/types.rs discriminants ---- */
typedef enum { /* types.rs:115-123 */
    SARPRO_STRATEGY_STANDARD = 0,
    SARPRO_STRATEGY_ROBUST = 1,
    SARPRO_STRATEGY_ADAPTIVE = 2,
    SARPRO_STRATEGY_EQUALIZED = 3,
    SARPRO_STRATEGY_CLAHE = 4,
    SARPRO_STRATEGY_TAMED = 5,
    SARPRO_STRATEGY_DEFAULT = 6
} sarpro_strategy;
typedef enum { SARPRO_BITDEPTH_U8 = 0, SARPRO_BITDEPTH_U16 = 1 } sarpro_bitdepth; /* types.rs:170-173 */
typedef enum { /* types.rs:8-14 */
    SARPRO_OP_SUM = 0,
    SARPRO_OP_DIFF = 1,
    SARPRO_OP_RATIO = 2,
    SARPRO_OP_NDIFF = 3,
    SARPRO_OP_LOGRATIO = 4
} sarpro_polop;
typedef enum { /* types.rs:177-182 */
    SARPRO_SYNRGB_DEFAULT = 0,
    SARPRO_SYNRGB_RGB_RATIO = 1,
    SARPRO_SYNRGB_SAR_URBAN = 2,
    SARPRO_SYNRGB_ENHANCED = 3
} sarpro_synrgb_mode;

/* HistogramStats (autoscale.rs:7-24) + the window the reference only logs
 * (autoscale.rs:398-401,431-434,485-488,566-569). */
typedef struct {
    uint64_t valid_count;
    double min_db, max_db, mean_db, std_db, median_db;
    double p01, p02, p05, p10, p25, p75, p90, p95, p98, p99;
    double low_clip, high_clip, gamma;
    double skew_factor, tail_heaviness; /* Adaptive only (autoscale.rs:503-504) */
} sarpro_hip_stats;

typedef struct sarpro_hip_ctx sarpro_hip_ctx;

/* ---- context ---- */
#define SARPRO_HIP_CTX_TIMING 1u    /* HIP-event pairs around the kernels (sarpro_hip_last_kernel_times) */
/* Stream-ordered device entry points: sarpro_hip_dualpol_synrgb_u16_dev called with u8 outputs, stats_out == NULL
 * and a scene the device-resident chain takes (16-byte aligned rasters, pitch % 8 == 0) returns once the chain is
 * ENQUEUED on the context's stream: the rasters are complete after sarpro_hip_ctx_synchronize (or after work the
 * caller orders behind sarpro_hip_ctx_stream).  Scenes enqueued back to back run back to back: the GPU does not
 * idle for the ~30 us a host round trip per scene costs.  Every other call stays synchronous.  The reference's
 * functions are synchronous: this is an opt-in for callers that keep rasters in HBM. */
#define SARPRO_HIP_CTX_ASYNC_DEV 2u
int sarpro_hip_ctx_create(int device, unsigned flags, sarpro_hip_ctx **ctx_out);
void sarpro_hip_ctx_destroy(sarpro_hip_ctx *ctx);
const char *sarpro_hip_last_error(const sarpro_hip_ctx *ctx); /* ctx may be NULL: last ctx_create error */
const char *sarpro_hip_version(void);
/* raw hipStream_t of the context (for callers that enqueue their own work / events) */
void *sarpro_hip_ctx_stream(sarpro_hip_ctx *ctx);
int sarpro_hip_ctx_synchronize(sarpro_hip_ctx *ctx);

/* ================= host-pointer entry points (the drop-in seam) ================= */

/* process_scalar_data_pipeline (pipeline.rs:42-67) for an f32 band as GDAL hands it over
 * (io/gdal.rs:123-131).  U8: fills out_u8 (out_u16 may be NULL); U16: fills out_u16.
 * The dB buffer / mask the reference also returns are not materialised (callers only
 * use their dims, save.rs:331; see sarpro_hip_db_mask_f32 for the buffers themselves).
 * stats_out may be NULL (the f32 flavour then skips the per-sample dB moments, which only the statistics and the
 * Adaptive strategy read). */
int sarpro_hip_autoscale_band_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols,
                                  int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16,
                                  sarpro_hip_stats *stats_out);
/* Same function for the u16 DN samples of a full-resolution GRD read (io/gdal.rs:107-141
 * reads u16 as f32; values are integral), the fast flavour. */
int sarpro_hip_autoscale_band_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols,
                                  int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16,
                                  sarpro_hip_stats *stats_out);

/* process_scalar_data_inplace (pipeline.rs:8-40): db = 10*log10(max(v,1e-10)) as f64,
 * mask = db > -50 (exact).  Either output may be NULL.  db is within 2 ulp(f64) of glibc's
 * value (so within 1 ulp when held as f32). */
int sarpro_hip_db_mask_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols,
                           double *db_out, uint8_t *mask_out);

/* autoscale_db_image_tamed_synrgb_u8 (autoscale.rs:710-742); takes the band itself
 * (the dB buffer is recomputed on the device, never shipped). */
int sarpro_hip_tamed_synrgb_u8_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols,
                                   int is_copol, uint8_t *out_u8);
int sarpro_hip_tamed_synrgb_u8_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols,
                                   int is_copol, uint8_t *out_u8);

/* sum_arrays / difference_arrays / ratio_arrays / normalized_diff_arrays /
 * log_ratio_arrays (ops.rs:4-44). */
int sarpro_hip_polop_f32(sarpro_hip_ctx *ctx, int op, const float *a, const float *b, size_t n,
                         float *out);

/* Pol-op band straight to its autoscaled raster: sum / difference / ratio / normalized_diff / log_ratio_arrays
 * (ops.rs:4-44) followed by process_scalar_data_pipeline (pipeline.rs:42-67) on the result -- what
 * io/sentinel1.rs:1501-1578 + save.rs do for a `PolarOp` product -- with the operation computed in registers inside
 * every pass: the f32 pol-op raster is never written or read (12 B/px of traffic less than polop_f32 + autoscale_band_f32,
 * same raster bit for bit).  a, b: the two co-registered bands, as f32 or as the u16 DN of a full-resolution GRD read. */
int sarpro_hip_polop_autoscale_band_f32(sarpro_hip_ctx *ctx, int op, const float *a, const float *b, size_t rows, size_t cols,
                                        int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16,
                                        sarpro_hip_stats *stats_out);
int sarpro_hip_polop_autoscale_band_u16(sarpro_hip_ctx *ctx, int op, const uint16_t *a, const uint16_t *b, size_t rows, size_t cols,
                                        int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16,
                                        sarpro_hip_stats *stats_out);

/* create_synthetic_rgb_by_mode_and_strategy (synthetic_rgb.rs:182-197): interleaved RGB,
 * rgb_out holds 3*n bytes. */
int sarpro_hip_synrgb_u8(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *band1,
                         const uint8_t *band2, size_t n, uint8_t *rgb_out);

/* The JPEG/multiband branch of save_processed_multiband_image_sequential at native
 * resolution (save.rs:317-367): pipeline(band1,U8) -> [Tamed: tamed_synrgb(copol)] ->
 * pipeline(band2,U8) -> [Tamed: tamed_synrgb(cross-pol)] -> synRGB by strategy, fused on the
 * device.  rgb_out holds 3*rows*cols bytes; u8_band1/u8_band2 (optional) receive the
 * per-band u8 rasters that feed the composition; stats_out (optional) is [2]. */
int sarpro_hip_dualpol_synrgb_u16(sarpro_hip_ctx *ctx, const uint16_t *band1, const uint16_t *band2,
                                  size_t rows, size_t cols, int strategy, int mode,
                                  uint8_t *rgb_out, uint8_t *u8_band1, uint8_t *u8_band2,
                                  sarpro_hip_stats *stats_out);
int sarpro_hip_dualpol_synrgb_f32(sarpro_hip_ctx *ctx, const float *band1, const float *band2,
                                  size_t rows, size_t cols, int strategy, int mode,
                                  uint8_t *rgb_out, uint8_t *u8_band1, uint8_t *u8_band2,
                                  sarpro_hip_stats *stats_out);
/* f32 bands are what the reference's DEFAULT flow hands the raster core: `--size N` resamples on read (sentinel1.rs:1074-1108), so
 * the bands arrive as non-integer f32.  flags for the f32 dual-pol entry points below:
 *   SARPRO_HIP_DUALPOL_PLAIN_PIPELINE  api/mod.rs:404-437 (process_safe_to_buffer_with_mode): both bands go through
 *                                      process_scalar_data_pipeline with the caller's strategy; without it save.rs:324-351 applies:
 *                                      under Tamed the bands are re-autoscaled band-specifically (autoscale.rs:710-742). */
#define SARPRO_HIP_DUALPOL_PLAIN_PIPELINE 1u
/* save.rs:317-367 / api/mod.rs:404-437 with resize -> pad between the per-band autoscale and the composition, host f32 bands in,
 * final_rows * final_cols * 3 bytes out (sarpro_hip_resize_output_dims); the reference's order, so the suppressed floor sees the
 * padding.  The resize happens on the u8 level rasters, as in the reference. */
/* (declared with the resize entry points below: sarpro_hip_dualpol_synrgb_resized_f32 / _f32_dev) */

/* ================= resize + pad (first "next" row) ================= */
/* Bookkeeping the reference returns next to the raster (resize.rs:98-108). */
typedef struct {
    size_t final_cols, final_rows;
    double scale_x, scale_y;
    size_t pad_left, pad_top;
} sarpro_hip_resize_meta;
/* Output shape of resize_image_data_with_meta for (cols, rows, target_size (0 = None), pad):
 * calculate_resize_dimensions (resize.rs:6-30) + add_padding_to_square (padding.rs:12). */
int sarpro_hip_resize_output_dims(size_t cols, size_t rows, size_t target_size, int pad, size_t *final_cols,
                                  size_t *final_rows);
/* resize_image_data_with_meta (resize.rs:91-236): Lanczos3 resample to target_size on the long side
 * (skipped when already there), then optional centre zero-pad to a square.  bit_depth selects u8 / u16
 * elements.  `out` holds final_rows * final_cols elements.  Dimension and padding rules are exact; the
 * Lanczos3 arithmetic restates the fast_image_resize crate's convolution and its parity with the crate
 * is unpinned (crate source absent, version not locked). */
int sarpro_hip_resize_image_data(sarpro_hip_ctx *ctx, const void *data, size_t cols, size_t rows, size_t target_size,
                                 int bit_depth, int pad, void *out, sarpro_hip_resize_meta *meta);
int sarpro_hip_resize_image_data_dev(sarpro_hip_ctx *ctx, const void *d_data, size_t cols, size_t rows, size_t pitch,
                                     size_t target_size, int bit_depth, int pad, void *d_out, size_t out_pitch,
                                     sarpro_hip_resize_meta *meta);
/* The JPEG/multiband branch with its resize and pad steps (save.rs:317-367): per-band u8 autoscale at
 * native resolution -> resize -> pad -> synRGB, all on the device; only the final RGB
 * (final_rows * final_cols * 3 bytes, see sarpro_hip_resize_output_dims) crosses PCIe. */
int sarpro_hip_dualpol_synrgb_resized_u16(sarpro_hip_ctx *ctx, const uint16_t *band1, const uint16_t *band2, size_t rows,
                                          size_t cols, int strategy, int mode, size_t target_size, int pad,
                                          uint8_t *rgb_out, sarpro_hip_resize_meta *meta);
int sarpro_hip_dualpol_synrgb_resized_f32(sarpro_hip_ctx *ctx, const float *band1, const float *band2, size_t rows, size_t cols,
                                          int strategy, int mode, unsigned flags, size_t target_size, int pad, uint8_t *rgb_out,
                                          sarpro_hip_resize_meta *meta);
int sarpro_hip_dualpol_synrgb_resized_f32_dev(sarpro_hip_ctx *ctx, const float *d_band1, const float *d_band2, size_t rows, size_t cols,
                                              size_t in_pitch, int strategy, int mode, unsigned flags, size_t target_size, int pad,
                                              uint8_t *d_rgb_out, sarpro_hip_resize_meta *meta);
/* The same product with both bands and the RGB raster (final_rows * final_cols * 3 bytes, compact) resident in device memory:
 * nothing crosses PCIe.  in_pitch in elements.  Synchronous. */
int sarpro_hip_dualpol_synrgb_resized_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows,
                                              size_t cols, size_t in_pitch, int strategy, int mode, size_t target_size, int pad,
                                              uint8_t *d_rgb_out, sarpro_hip_resize_meta *meta);

/* save_processed_image (save.rs:23-170) up to the raster its writer receives: pipeline at native
 * resolution -> resize -> pad, on the device.  `out` holds final_rows * final_cols u8 / u16 elements.
 * The multiband TIFF branch (save.rs:200-316) is this call once per band. */
int sarpro_hip_process_band_resized_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols, int strategy,
                                        int bit_depth, size_t target_size, int pad, void *out,
                                        sarpro_hip_resize_meta *meta);
int sarpro_hip_process_band_resized_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols, int strategy,
                                        int bit_depth, size_t target_size, int pad, void *out,
                                        sarpro_hip_resize_meta *meta);

/* Batch mode (api/mod.rs:474-536 process_directory_to_path; BatchReport :453-458) for scenes already
 * decoded into host memory: workers_per_device worker threads (each with a context of its own) per listed
 * device, scenes dealt dynamically, no collective.  workers_per_device = 0 picks the default, 2: a scene's
 * call is upload -> chain -> download, one PCIe direction at a time; with two workers on a device scene i's
 * D2H crosses the full-duplex link beside scene i + 1's H2D (50.2 -> 31.8 ms per 400 MP scene, bench.py
 * `secondary.e2e_two_in_flight`).  continue_on_error = 0 stops handing out scenes after the first failure
 * (the rest count as skipped) and returns that failure's status. */
/* row-chunk callbacks of the streaming entry points (described with sarpro_hip_dualpol_synrgb_stream_u16 below) */
typedef int (*sarpro_hip_row_reader)(void *user, int band, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch);
typedef int (*sarpro_hip_row_sink)(void *user, size_t row0, size_t nrows, const uint8_t *src, size_t src_pitch_bytes);
typedef struct {
    const uint16_t *band1, *band2; /* rows x cols each; ignored when `reader` is set */
    size_t rows, cols;
    uint8_t *rgb_out;              /* final_rows * final_cols * 3 (sarpro_hip_resize_output_dims) */
    int *status_out;               /* optional per-scene status */
    sarpro_hip_row_reader reader;  /* optional: the scene's bands come through this row reader (a batch of SAFE scenes
                                      does not fit host memory as arrays: api/mod.rs:484-533 opens them one by one) */
    void *reader_user;
} sarpro_hip_batch_scene;
typedef struct { size_t processed, skipped, errors; } sarpro_hip_batch_report;
int sarpro_hip_batch_dualpol_synrgb_resized_u16(const int *devices, int ndevices, int workers_per_device, const sarpro_hip_batch_scene *scenes,
                                                size_t nscenes, int strategy, int mode, size_t target_size, int pad,
                                                int continue_on_error, sarpro_hip_batch_report *report);

/* f32 bands (the reference's default, resampled-on-read flow); flags: SARPRO_HIP_DUALPOL_* */
typedef struct {
    const float *band1, *band2; /* rows x cols each */
    size_t rows, cols;
    uint8_t *rgb_out;           /* final_rows * final_cols * 3 */
    int *status_out;            /* optional per-scene status */
} sarpro_hip_batch_scene_f32;
int sarpro_hip_batch_dualpol_synrgb_resized_f32(const int *devices, int ndevices, int workers_per_device, const sarpro_hip_batch_scene_f32 *scenes,
                                                size_t nscenes, int strategy, int mode, unsigned flags, size_t target_size, int pad,
                                                int continue_on_error, sarpro_hip_batch_report *report);

/* Resident batch: the batch loop of api/mod.rs:484-533 for scenes whose bands are ALREADY in device memory (decoded and uploaded by
 * the caller, or produced on the device), all of one shape -- the save.rs:317-367 product at native resolution per scene:
 * d_band1, d_band2 -> per-band autoscale -> synRGB -> d_rgb (rgb_pitch_px pixels per row).  One context, `lanes` internal lanes (each
 * its own stream and workspace set, created on first use; 0 = the library's default): scene i runs on lane i mod lanes, so scene
 * i + 1's histogram sweep and its short dependent kernels (statistics, CLAHE bins, CDFs, sample, prediction) are enqueued beside
 * scene i's CLAHE -> RGB pass instead of behind it.  Every scene takes the route it would take alone (speculative, refuted or
 * unproven -> exact kernels): the rasters are those of sarpro_hip_dualpol_synrgb_u16_dev, bit for bit.  The call returns when every
 * raster is complete.  status / route are written per scene (route: SARPRO_HIP_ROUTE_*; -1 when no speculative chain ran for it);
 * continue_on_error = 0 stops enqueuing after the first failure (the rest count as skipped), as the other batch entry points.
 * Stream ordering: the lanes' streams are ordered against nothing of the caller's -- inputs must be complete before the call. */
#define SARPRO_HIP_ROUTE_NONE (-1)      /* no speculative CLAHE chain ran (another strategy, a small or unaligned scene) */
#define SARPRO_HIP_ROUTE_ACCEPTED 0     /* the fused CLAHE -> RGB pass's raster stood */
#define SARPRO_HIP_ROUTE_REFUTED 1      /* the predicted floor / rescale range was refuted in the pass: exact kernels produced the raster */
#define SARPRO_HIP_ROUTE_UNPROVEN 2     /* the speculation never started */
#define SARPRO_HIP_ROUTE_POOL_OVERFLOW 3 /* DN windows beyond the fused pass's LDS pool: exact kernels */
#define SARPRO_HIP_ROUTE_RETRIED 4      /* the first floor was refuted, a second fused pass with the floor its counts point to stood */
typedef struct {
    const uint16_t *d_band1, *d_band2; /* device, rows x in_pitch */
    uint8_t *d_rgb;                    /* device, rows x rgb_pitch_px x 3 */
    int status;                        /* out */
    int route;                         /* out */
} sarpro_hip_resident_scene;
int sarpro_hip_batch_dualpol_synrgb_u16_dev(sarpro_hip_ctx *ctx, sarpro_hip_resident_scene *scenes, size_t nscenes, size_t rows,
                                            size_t cols, size_t in_pitch, int strategy, int mode, size_t rgb_pitch_px, int lanes,
                                            int continue_on_error, sarpro_hip_batch_report *report);

/* The resident batch for f32 bands -- the reference's default flow (api/mod.rs:374-449: bands resampled on read, a few megapixels) looped
 * over the scenes of a directory (api/mod.rs:474-536), for bands that are already in device memory: per scene
 * sarpro_hip_dualpol_synrgb_resized_f32_dev (flags: SARPRO_HIP_DUALPOL_*; d_rgb: final_rows * final_cols * 3 bytes, compact), on one of
 * `lanes` internal lanes of the context (0 = the default), each driven by a host thread of its own for the duration of the call: at this
 * size the chain is bound by its host turns, which now overlap.  The rasters are the single call's, bit for bit.  BatchReport semantics
 * as the other batch entry points; the call returns when every raster is complete. */
typedef struct {
    const float *d_band1, *d_band2;    /* device, rows x in_pitch */
    uint8_t *d_rgb;                    /* device, final_rows x final_cols x 3 */
    int status;                        /* out */
} sarpro_hip_resident_scene_f32;
int sarpro_hip_batch_dualpol_synrgb_resized_f32_dev(sarpro_hip_ctx *ctx, sarpro_hip_resident_scene_f32 *scenes, size_t nscenes, size_t rows, size_t cols,
                                                    size_t in_pitch, int strategy, int mode, unsigned flags, size_t target_size, int pad, int lanes,
                                                    int continue_on_error, sarpro_hip_batch_report *report);

/* ================= device-pointer entry points ================= */
/* Same operations on rasters already resident in HBM.  pitch = row stride in elements
 * (>= cols).  The vectorised kernels need base pointers aligned to 16 bytes and
 * pitch % 8 == 0; other layouts take the scalar kernels (same results). */
int sarpro_hip_autoscale_band_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows,
                                      size_t cols, size_t in_pitch, int strategy, int bit_depth,
                                      void *d_out, size_t out_pitch, sarpro_hip_stats *stats_out);
int sarpro_hip_autoscale_band_f32_dev(sarpro_hip_ctx *ctx, const float *d_in, size_t rows,
                                      size_t cols, size_t in_pitch, int strategy, int bit_depth,
                                      void *d_out, size_t out_pitch, sarpro_hip_stats *stats_out);
/* f32 bands resident in device memory (pitches in elements / pixels); flags: SARPRO_HIP_DUALPOL_*.  Synchronous. */
int sarpro_hip_dualpol_synrgb_f32_dev(sarpro_hip_ctx *ctx, const float *d_band1, const float *d_band2, size_t rows, size_t cols,
                                      size_t in_pitch, int strategy, int mode, unsigned flags, uint8_t *d_rgb, size_t rgb_pitch_px,
                                      uint8_t *d_u8_band1, uint8_t *d_u8_band2, size_t u8_pitch, sarpro_hip_stats *stats_out);
int sarpro_hip_dualpol_synrgb_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_band1,
                                      const uint16_t *d_band2, size_t rows, size_t cols,
                                      size_t in_pitch, int strategy, int mode, uint8_t *d_rgb,
                                      size_t rgb_pitch_px, uint8_t *d_u8_band1, uint8_t *d_u8_band2,
                                      size_t u8_pitch, sarpro_hip_stats *stats_out);
/* device-pointer forms of sarpro_hip_polop_autoscale_band_*: in_pitch (elements) is shared by d_a and d_b; d_out is u8 or
 * u16 by bit_depth, out_pitch in elements */
int sarpro_hip_polop_autoscale_band_f32_dev(sarpro_hip_ctx *ctx, int op, const float *d_a, const float *d_b, size_t rows, size_t cols,
                                            size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                            sarpro_hip_stats *stats_out);
int sarpro_hip_polop_autoscale_band_u16_dev(sarpro_hip_ctx *ctx, int op, const uint16_t *d_a, const uint16_t *d_b, size_t rows, size_t cols,
                                            size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                            sarpro_hip_stats *stats_out);
int sarpro_hip_polop_f32_dev(sarpro_hip_ctx *ctx, int op, const float *d_a, const float *d_b,
                             size_t n, float *d_out);
int sarpro_hip_synrgb_u8_dev(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *d_band1,
                             const uint8_t *d_band2, size_t n, uint8_t *d_rgb);

/* ---- per-kernel timing of the last *_dev / host call on this context (HIP events on the
 * context's stream).  names_out receives up to max_entries pointers to static strings. */
int sarpro_hip_last_kernel_times(sarpro_hip_ctx *ctx, const char **names_out, float *ms_out,
                                 int max_entries);
/* Restrict the event pairs to the kernel of this name (as reported by sarpro_hip_last_kernel_times); NULL or ""
 * = every kernel.  An event pair costs the stream about 10 us of idle time between two kernels: a chain of a
 * dozen kernels timed one by one runs ~6 % slower than untimed, timed on its dominant kernel only ~0.5 %. */
int sarpro_hip_ctx_time_only(sarpro_hip_ctx *ctx, const char *kernel_name);

/* ---- context attributes: the route switches.  Every fast route of the library has a slower twin that must give the same raster
 * ("Cross-check switches"), plus a few test hooks and planner tuning values.  They live on the CONTEXT: each
 * attribute NAME (e.g. "NO_SPEC", "SAMPLE_STRIDE"; an optional "SARPRO_HIP_" prefix is accepted) is either unset (the default
 * behaviour) or set to an integer; a switch is on when set and non-zero.  The environment variable SARPRO_HIP_<NAME> is read ONCE,
 * when the context is created, as the attribute's initial value (a non-numeric word counts as 1; SPEC_FORCE takes
 * "mispredict" = 1 and / or "nospec" = 2, F32_ZONES takes "tiny" = 2); no call path reads the environment afterwards.  A context's
 * twin (the second band of a dual-pol f32 product) follows its parent.  Planner attributes (STRIP_ALIGN, CHUNK_ROWS, ...) act when a
 * shape's plan is first built.  sarpro_hip_attr_name(i) enumerates the names (NULL past the last). */
int sarpro_hip_ctx_set_attr(sarpro_hip_ctx *ctx, const char *name, int64_t value);
int sarpro_hip_ctx_reset_attr(sarpro_hip_ctx *ctx, const char *name); /* back to "unset" */
int sarpro_hip_ctx_get_attr(const sarpro_hip_ctx *ctx, const char *name, int64_t *value, int *is_set);
const char *sarpro_hip_attr_name(int index);

/* Diagnostics of the last speculative CLAHE chain of this context.  A whole dual-pol u8 scene (>= 32 MP) counts its level
 * histogram on sampled rows only; from the sample the chain PROVES that the u8 rescale of autoscale.rs:348-364 is the
 * identity (levels 0 and 255 occur) and PREDICTS the suppressed-synRGB floor (synthetic_rgb.rs:99-113), composes with the
 * prediction, and verifies it exactly inside the compose pass (csrc/chain_kernels.hip, k_chain_predict).  Synchronises the
 * context's stream.  spec_ok: 1 = the proof held and the speculative composition ran; 2 = a band holds no level 0 (a crop
 * without invalid pixels), its lowest level min_pred is a PREDICTION too, the rescale (min_pred, 255) is folded into the tables and
 * the fused pass counts the level bytes below min_pred (n_below_min: any refutes); 0 = no speculation.  verdict 0: the speculative
 * RGB stood, 1: refuted (or never ran) and the exact recount + composition ran; the raster is the reference's either way.
 * A refuted FLOOR gets one second fused pass first (retried = 1): the pass's counts say on which side of the prediction the floor lies. */
typedef struct {
    uint32_t spec_ok, verdict;
    int32_t floor_pred;       /* predicted floor before the +3 cushion; 37 stands for "37 or more" (the cushion caps at 40) */
    uint32_t pool_overflow;   /* 1: spec_ok, but the fused CLAHE -> RGB pass stepped aside (the bands' DN windows exceed its LDS pool): verdict 1 */
    uint64_t n_lt[2];         /* band-pixels with level < floor_pred, < floor_pred + 1 (exact, counted by the compose pass) */
    uint64_t target;          /* synthetic_rgb.rs:99-100 */
    double est_lt[2];         /* the sample's estimate of n_lt */
    uint64_t sample_valid[2]; /* per band: valid pixels on the sampled rows, each work item weighted by rows / sampled rows, 4096 = 1.0 */
    uint64_t n_below_min;     /* spec_ok 2: band-pixels whose level lies below their band's min_pred (exact; 0 or refuted) */
    uint32_t min_pred[2];     /* spec_ok 2: predicted lowest level per band (0: proven) */
    uint32_t retried;         /* 1: the first floor was refuted and a second fused pass ran with the floor the counts pointed to (floor_pred is that floor, verdict its verdict) */
    int32_t floor_first;      /* the floor the first pass tried */
} sarpro_hip_spec_report;
int sarpro_hip_ctx_spec_report(sarpro_hip_ctx *ctx, sarpro_hip_spec_report *out);

/* Diagnostics of the last device-resident CLAHE chain of this context (u16 flavour): what stood between the CLAHE levels and the
 * composition.  rescale[b]: the u8 rescale of band b's levels (autoscale.rs:348-364) as a table; floor_with_cushion: synthetic_rgb.rs:
 * 110-113 (-1: none, no synRGB was composed); level_hist[b]: the exact level histogram the exact kernels built them from (bin 0 implied:
 * pixels - sum of the others) -- all zero when the speculative composition stood (nothing was recounted).  Synchronises the stream. */
typedef struct {
    int32_t floor_with_cushion;
    uint8_t identity[2];      /* 1: band b's rescale is the identity on the levels that occur */
    uint8_t reserved[2];
    uint8_t rescale[512];     /* [band][256] */
    uint64_t level_hist[512]; /* [band][256] */
} sarpro_hip_chain_report;
int sarpro_hip_ctx_chain_report(sarpro_hip_ctx *ctx, sarpro_hip_chain_report *out);

/* ================= row-stripe (multi-GPU) protocol ================= */
/* One scene split into row stripes, one per rank.  Each phase
 * ends in a small integer reduction that the caller merges across ranks (RCCL
 * all-reduce(sum) on the returned DEVICE buffers, or sarpro_hip_comm_* below), so the
 * N-rank result is bit-identical to the 1-rank result.  A phase returns with its buffer
 * complete; the caller's reduction must be complete before it calls the next phase. */
typedef struct sarpro_hip_stripe sarpro_hip_stripe;
/* rows_total x cols scene; this rank owns rows [row0, row0+rows_local).  d_band1/2 are the
 * local stripe (rows_local x cols, pitch in elements). */
int sarpro_hip_stripe_begin_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2,
                                size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                size_t in_pitch, int strategy, int mode, sarpro_hip_stripe **out);
/* phase 1: local DN histograms.  *d_buf: device u64 buffer to all-reduce(sum), *count its length. */
int sarpro_hip_stripe_phase1(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count);
/* phase 2 (CLAHE only; no-op otherwise): local per-tile bin histograms -> all-reduce(sum). */
int sarpro_hip_stripe_phase2(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count);
/* phase 3: apply; local histogram of the pre-rescale u8 levels -> all-reduce(sum). */
int sarpro_hip_stripe_phase3(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count);
/* phase 4: compose the local stripe of the RGB raster (rows_local x cols x 3). */
int sarpro_hip_stripe_phase4(sarpro_hip_stripe *s, uint8_t *d_rgb, size_t rgb_pitch_px,
                             sarpro_hip_stats *stats_out);
void sarpro_hip_stripe_end(sarpro_hip_stripe *s);

/* ================= streaming ingest / egress (north-star: "staged to device via pinned
 * hipMemcpyAsync overlapped on a side stream") =================
 * The caller keeps its decoder (GDAL RasterIO, the strip-TIFF reader below, a socket ...): `reader` fills `nrows`
 * rows of band 0 or 1 starting at `row0` into a PINNED buffer of the library (`dst_pitch` elements per row);
 * `sink` receives finished interleaved RGB rows (`src_pitch_bytes` per row, cols * 3 valid).  Both return 0 or an
 * error code of their own, which aborts the call with SARPRO_HIP_ERR_IO.  While the reader produces chunk k + 1,
 * chunk k crosses PCIe on a side stream and the DN-histogram work items of the rows that have arrived run on the
 * compute stream; the RGB leaves chunk by chunk the same way.  Replaces the read loop + processing + write of
 * save_processed_multiband_image_sequential's JPEG branch at native resolution (save.rs:317-367, io/gdal.rs:107-141).
 * chunk_rows = 0 picks ~32 MiB chunks. */
int sarpro_hip_dualpol_synrgb_stream_u16(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *reader_user, size_t rows,
                                         size_t cols, int strategy, int mode, size_t chunk_rows, sarpro_hip_row_sink sink,
                                         void *sink_user, sarpro_hip_stats *stats_out);

/* save.rs:317-367 with its resize / pad (sarpro_hip_dualpol_synrgb_resized_u16) fed by the row reader: the product is
 * small (target_size^2 x 3 bytes), so it is returned in one piece. */
int sarpro_hip_dualpol_synrgb_resized_stream_u16(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *reader_user, size_t rows,
                                                 size_t cols, int strategy, int mode, size_t target_size, int pad, uint8_t *rgb_out,
                                                 sarpro_hip_resize_meta *meta);

/* ---- file shims for the two callbacks: uncompressed strip TIFF / BigTIFF (what Sentinel-1 GRD measurement rasters
 * are; io/gdal.rs:107-141 and io/writers/tiff.rs:6-78 go through GDAL).  Baseline TIFF 6.0 + BigTIFF, II or MM,
 * Compression = 1, strips, 8- / 16-bit unsigned samples, chunky or planar; anything else fails with
 * SARPRO_HIP_ERR_IO and a message (sarpro_hip_tiff_last_error, per thread).  No GPU involved. ---- */
typedef struct sarpro_hip_tiff sarpro_hip_tiff;
typedef struct sarpro_hip_tiff_writer sarpro_hip_tiff_writer;
typedef struct {
    uint64_t width, height, rows_per_strip;
    uint32_t bits_per_sample, samples_per_pixel, sample_format, planar, compression, big_endian, bigtiff, tiled;
    uint32_t has_geo;        /* bit 0: pixel_scale, bit 1: tiepoint, bit 2: GeoKey directory */
    uint32_t tiepoint_count; /* GRD products carry a GCP grid: only the first tiepoint is returned here */
    double pixel_scale[3], tiepoint[6];
} sarpro_hip_tiff_info;
int sarpro_hip_tiff_open(const char *path, sarpro_hip_tiff **out, sarpro_hip_tiff_info *info_out);
int sarpro_hip_tiff_read_rows_u16(sarpro_hip_tiff *t, int sample, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch);
void sarpro_hip_tiff_close(sarpro_hip_tiff *t);
/* a sarpro_hip_row_reader over two single-band files: user = sarpro_hip_tiff *[2] (band 0, band 1) */
int sarpro_hip_tiff_pair_reader(void *user, int band, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch);
/* geotransform6 (GDAL order, north-up) and geo_keys_from (GeoKey directory carried over verbatim) may be NULL */
int sarpro_hip_tiff_create(const char *path, uint64_t width, uint64_t height, uint32_t samples, uint32_t bits,
                           const double *geotransform6, const sarpro_hip_tiff *geo_keys_from, sarpro_hip_tiff_writer **out);
int sarpro_hip_tiff_write_rows(sarpro_hip_tiff_writer *w, size_t row0, size_t nrows, const void *src, size_t src_pitch_bytes);
/* a sarpro_hip_row_sink: user = sarpro_hip_tiff_writer * */
int sarpro_hip_tiff_row_sink(void *user, size_t row0, size_t nrows, const uint8_t *src, size_t src_pitch_bytes);
int sarpro_hip_tiff_finish(sarpro_hip_tiff_writer *w); /* writes the directory, closes and frees the writer */
const char *sarpro_hip_tiff_last_error(void);
/* sysfs cpulist syntax ("0-31,64-95") -> CPU numbers, at most max of them; returns the count, -1 on a syntax error.  The batch
 * driver binds each worker thread to the CPUs of its GPU's NUMA node with it (csrc/batch.cpp). */
int sarpro_hip_host_parse_cpulist(const char *s, int *cpus, int max);
/* the geotransform of a resized / padded product, save.rs:71-81 (pixel size by cols / final_cols, origin by the padding) */
void sarpro_hip_host_update_geotransform(double gt[6], size_t cols, size_t rows, const sarpro_hip_resize_meta *m);

/* The same stripe in ONE call per rank, reductions over the library's communicator (sarpro_hip_comm_init
 * must have been called): the device-resident chains with their small all-reduces enqueued on the stream,
 * no host synchronisation until the stripe's RGB (rows_local x cols x 3) is complete.  Every rank of the
 * communicator must make the call (a rank may hold an empty stripe). */
int sarpro_hip_stripe_run_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total,
                              size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode,
                              uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out);

/* Row stripes of one scene -> the RESIZED, padded product (save.rs:317-367: per-band u8 autoscale at native resolution -> Lanczos3
 * resize -> pad -> synRGB; SURVEY.md 8e's halo item).  Each rank holds rows [row0, row0 + rows_local) of both DN rasters -- 1 / N of
 * the scene crossed its PCIe link -- and produces a contiguous range of the FINAL raster's rows: output row j of the vertical pass
 * belongs to the rank holding the centre row of j's window, the rows its window needs from the neighbours travel in ONE small
 * all-reduce (sarpro_amd/csrc/resize_path.cpp); the rank owning the first / last resized row also owns the padding above / below.
 * d_rgb_slice receives out_rows x final_cols x 3 bytes, compact; the caller places them at row *out_row0 of the product.  Size the
 * slice with sarpro_hip_stripe_resized_rows (pure host arithmetic, the same answer on every rank).  The stripes must tile the scene in
 * rank order; every rank of the communicator makes the call (a rank may hold an empty stripe).  The assembled raster is that of
 * sarpro_hip_dualpol_synrgb_resized_u16_dev on the one-piece scene, bit for bit. */
int sarpro_hip_stripe_resized_rows(size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t target_size, int pad,
                                   size_t *out_row0, size_t *out_rows, size_t *final_cols, size_t *final_rows);
int sarpro_hip_stripe_run_resized_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total, size_t cols,
                                      size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode, size_t target_size, int pad,
                                      uint8_t *d_rgb_slice, size_t *out_row0, size_t *out_rows, sarpro_hip_resize_meta *meta);
/* the same for f32 bands (the reference's default, resampled-on-read flow; flags: SARPRO_HIP_DUALPOL_*): the levels come from the
 * striped f32 chain (process_scalar_data_pipeline at U8); under Tamed without SARPRO_HIP_DUALPOL_PLAIN_PIPELINE from the band-specific
 * re-autoscale of save.rs:324-351 (autoscale_db_image_tamed_synrgb_u8 over stripes: the same collectives, the Tamed windows from the
 * same all-reduced bins), with the flag as api/mod.rs:404-437 has it (both bands through the pipeline's Tamed arm). */
int sarpro_hip_stripe_run_resized_f32(sarpro_hip_ctx *ctx, const float *d_band1, const float *d_band2, size_t rows_total, size_t cols, size_t row0,
                                      size_t rows_local, size_t in_pitch, int strategy, int mode, unsigned flags, size_t target_size, int pad,
                                      uint8_t *d_rgb_slice, size_t *out_row0, size_t *out_rows, sarpro_hip_resize_meta *meta);

/* Self-test: the division the u16 pol-op kernels use (the Newton core of the IEEE division without its rescaling frame) against
 * the compiler's IEEE division over ALL 2^32 pairs of u16 values, ratio and normalised difference.  *mismatches_out must be 0. */
int sarpro_hip_selftest_polop_division(sarpro_hip_ctx *ctx, uint64_t *mismatches_out);

/* ---- row stripes of the f32 flavour (a calibrated f32 band, or a polarisation operation of two bands computed on the
 * fly): autoscale.rs:35-117 needs the scene's count / min / max before it can bin, and the scene's 4096 bins before it can
 * select the window, so the protocol has one more reduction than the u16 one.  One open f32 stripe per context. ---- */
typedef struct {
    uint64_t count;           /* valid samples                      merge: sum */
    double sum_db, sumsq_db;  /* of dB over the valid samples       merge: sum (in rank order: sarpro_hip_host_f32_merge_partials) */
    float min_v, max_v;       /* of the valid samples (+inf / -inf when there is none)   merge: min / max */
} sarpro_hip_f32_partial;
typedef struct sarpro_hip_stripe_f32 sarpro_hip_stripe_f32;
/* this rank owns rows [row0, row0 + rows_local) of the rows_total x cols scene; d_in / d_out are the local stripe */
int sarpro_hip_stripe_begin_f32(sarpro_hip_ctx *ctx, const float *d_in, size_t rows_total, size_t cols, size_t row0,
                                size_t rows_local, size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                sarpro_hip_stripe_f32 **out);
/* the samples are op(a, b) (ops.rs:4-44) of two f32 (elem_u16 = 0) or u16 DN (elem_u16 = 1) stripes */
int sarpro_hip_stripe_begin_polop(sarpro_hip_ctx *ctx, int op, const void *d_a, const void *d_b, int elem_u16, size_t rows_total,
                                  size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int bit_depth,
                                  void *d_out, size_t out_pitch, sarpro_hip_stripe_f32 **out);
/* phase 1: the stripe's count / min / max / dB moments, on the HOST (32 bytes); the caller gathers and merges them */
int sarpro_hip_stripe_f32_phase1(sarpro_hip_stripe_f32 *s, sarpro_hip_f32_partial *local_out);
int sarpro_hip_host_f32_merge_partials(const sarpro_hip_f32_partial *parts, size_t n, sarpro_hip_f32_partial *out);
/* phase 2: the stripe's 4096-bin histogram over the scene's [min, max] (autoscale.rs:102-117) -> all-reduce(sum) */
int sarpro_hip_stripe_f32_phase2(sarpro_hip_stripe_f32 *s, const sarpro_hip_f32_partial *global, uint64_t **d_buf, size_t *count);
/* phase 3: statistics and window from the merged bins; CLAHE: the stripe's tile histograms -> all-reduce(sum) (else count 0) */
int sarpro_hip_stripe_f32_phase3(sarpro_hip_stripe_f32 *s, uint64_t **d_buf, size_t *count);
/* phase 4: the stripe's levels; u8: histogram of the pre-rescale levels -> all-reduce(sum) (u16: count 0) */
int sarpro_hip_stripe_f32_phase4(sarpro_hip_stripe_f32 *s, uint64_t **d_buf, size_t *count);
/* phase 5: u8 rescale in place (autoscale.rs:348-364); the stripe of the output raster is complete on return */
int sarpro_hip_stripe_f32_phase5(sarpro_hip_stripe_f32 *s, sarpro_hip_stats *stats_out);
void sarpro_hip_stripe_f32_end(sarpro_hip_stripe_f32 *s);
/* the same in one call per rank over the library's communicator (every rank must make the call) */
int sarpro_hip_stripe_run_f32(sarpro_hip_ctx *ctx, const float *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                              size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch, sarpro_hip_stats *stats_out);
int sarpro_hip_stripe_run_polop(sarpro_hip_ctx *ctx, int op, const void *d_a, const void *d_b, int elem_u16, size_t rows_total,
                                size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int bit_depth,
                                void *d_out, size_t out_pitch, sarpro_hip_stats *stats_out);

/* RCCL communicator owned by the library (optional: callers may reduce the phase buffers
 * with their own communicator, e.g. torch.distributed's).  uid is the 128-byte
 * ncclUniqueId produced by sarpro_hip_comm_unique_id on rank 0 and shipped to all ranks. */
int sarpro_hip_comm_unique_id(uint8_t uid_out[128]);
int sarpro_hip_comm_init(sarpro_hip_ctx *ctx, int nranks, int rank, const uint8_t uid[128]);
int sarpro_hip_comm_allreduce_sum_u64(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count);
/* The same communicator for the contexts of ONE process (one context and one host thread per rank -- several GPUs of a node driven by
 * threads as sarpro_hip_batch_* does, or several contexts on one GPU): no RCCL, no rendezvous.  An all-reduce is a barrier, a sum kernel
 * over the ranks' device buffers (they must be mutually accessible: one device, or peer access enabled by the caller) and a second
 * barrier; every rank's thread must be inside the collective for it to complete.  Create the group once, join it from every context,
 * destroy it after the contexts.  The stripe entry points (sarpro_hip_stripe_run_*) use whichever communicator the context has. */
typedef struct sarpro_hip_local_group sarpro_hip_local_group;
int sarpro_hip_local_group_create(int nranks, sarpro_hip_local_group **out);
void sarpro_hip_local_group_destroy(sarpro_hip_local_group *group);
int sarpro_hip_comm_init_local(sarpro_hip_ctx *ctx, sarpro_hip_local_group *group, int rank);
void sarpro_hip_comm_destroy(sarpro_hip_ctx *ctx);

/* ================= host half of the path (no GPU needed) ================= */
/* compute_histogram_stats (autoscale.rs:35-160) from the exact 65536-bin DN histogram of a
 * u16 band (valid <=> DN >= 1, pipeline.rs:19-22). */
int sarpro_hip_host_stats_from_dn_hist(const uint64_t dn_hist[65536], sarpro_hip_stats *out);
/* Strategy window (autoscale.rs:404-429 Standard; :491-564 advanced; :721-729 tamed-synrgb
 * when tamed_synrgb = 1 copol / 2 crosspol) -> fills low_clip/high_clip/gamma in *stats. */
int sarpro_hip_host_window(sarpro_hip_stats *stats, int strategy, int tamed_synrgb);
/* Level of every DN under the window: the u16 the reference's map loop yields
 * (autoscale.rs:437-447 / :647-655; tamed :731-741), max_val 255 or 65535.  lut[0] = 0. */
int sarpro_hip_host_level_lut_u16(const sarpro_hip_stats *stats, int bit_depth, int tamed_synrgb,
                                  uint16_t lut_out[65536]);
/* CLAHE bin of every DN (autoscale.rs:583-591 then :262-265): 0..255. */
int sarpro_hip_host_clahe_bin_lut_u16(const sarpro_hip_stats *stats, uint8_t lut_out[65536]);
/* Clip / redistribute / CDF for all tiles (autoscale.rs:271-302).  tile_hists is
 * [8*8][256] counts in tile order ty*8+tx; cdfs_out is [64][256] f64. */
int sarpro_hip_host_clahe_cdfs(const uint64_t *tile_hists, size_t rows, size_t cols,
                               double *cdfs_out);
/* scale_u16_to_u8 (autoscale.rs:348-364) as a 256-entry map for levels 0..255 given the
 * global min / max of the level raster. */
int sarpro_hip_host_u8_rescale_lut(unsigned min_level, unsigned max_level, uint8_t lut_out[256]);
/* LUTs of create_synthetic_rgb (synthetic_rgb.rs:20-51) / _suppressed (:92-155, from the
 * combined 256-bin histogram of both u8 bands).  luts_out = lut_r[256] | lut_g[256] |
 * lut_b[65536]; floor_out = floor_with_cushion (suppressed) or -1. */
int sarpro_hip_host_synrgb_luts(int strategy, const uint64_t combined_hist[256], uint64_t n_per_band,
                                uint8_t *luts_out, int *floor_out);
/* CLAHE shape check: 0 when the reference's tile arithmetic underflows (autoscale.rs:250,254). */
int sarpro_hip_host_clahe_shape_ok(size_t rows, size_t cols);
/* autoscale.rs:327-329 for a bin whose four tile CDFs are all 1.0 (every sample from p99 up): level 255 or -- in the first half tile
 * row / column, where a blend weight is negative and (1 - d) + d can round below 1.0 -- 254, by the pixel's row and column alone.
 * col_class[cols] in {0, 1, 2}; bit k of row_bits[rows]: a saturated pixel of that row in a column of class k gets 255 (else 254).
 * What the fused CLAHE -> RGB pass uses instead of the f64 blend for such samples.  SARPRO_HIP_ERR_UNSUPPORTED_SHAPE: no table. */
int sarpro_hip_host_clahe_saturated_levels(size_t rows, size_t cols, uint8_t *col_class, uint8_t *row_bits);
/* Row-stripe plan: stripes aligned to CLAHE tile rows (tile_h = ceil(rows/8)).
 * row0_out / nrows_out hold nranks entries. */
int sarpro_hip_host_stripe_plan(size_t rows, int nranks, size_t *row0_out, size_t *nrows_out);

/* f32-input flavour: the per-pixel decisions as sorted f32 threshold tables (exact w.r.t. the
 * reference's f64/libm evaluation; the device only compares).  thr[0] is unused;
 * thr[k] = smallest valid f32 sample whose index/level is >= k (+inf if unreachable). */
/* percentiles of compute_histogram_stats (autoscale.rs:81-159) from its 4096-bin histogram */
int sarpro_hip_host_stats_from_bins4096(uint64_t valid_count, double min_db, double max_db, double mean_db,
                                        double std_db, const uint64_t hist4096[4096], sarpro_hip_stats *out);
float sarpro_hip_host_f32_valid_threshold(void); /* smallest f32 with 10*log10(v) > -50 (pipeline.rs:22) */
int sarpro_hip_host_f32_bin4096_thresholds(double min_db, double max_db, float thr_out[4096]); /* autoscale.rs:113-115 */
/* levels (autoscale.rs:440-442) under stats->low_clip/high_clip/gamma: 256 (U8) or 65536 (U16) entries */
int sarpro_hip_host_f32_level_thresholds(const sarpro_hip_stats *stats, int bit_depth, float *thr_out);
int sarpro_hip_host_f32_clahe_bin_thresholds(const sarpro_hip_stats *stats, float thr_out[256]); /* autoscale.rs:585-586,262-265 */

/* ================= synthetic scene generator (bench / tests) ================= */
/* counter-based (splitmix64) dual-pol GRD-like scene written
 * straight into HBM.  q_tables: 2 bands x 4 classes x 65536 u16 inverse-CDF tables
 * (host pointer).  band in {0,1}. */
int sarpro_hip_synth_scene_u16_dev(sarpro_hip_ctx *ctx, uint64_t seed, int band,
                                   const uint16_t *q_tables_host, size_t rows_total, size_t cols,
                                   size_t row0, size_t rows_local, uint16_t *d_out, size_t pitch);
/* The same generator with the scene's structure selectable (bench.py cycles its timed steps over scenes that differ in
 * distribution, not only in seed).  flags: bit 0 no no-data wedges (no invalid pixel anywhere), bit 1 no bright targets, bit 2 an all-invalid second band,
 * bits 4-7 class map (0: (r/b + 3 c/b) mod 4, the generator above; 1: (r/b xor c/b) mod 4; 2: diagonal bands ((r + c) / b) mod 4;
 * 3: one class -- table 1 -- everywhere), bits 8-15 class blocks per side (0 = 16: b = ceil(rows / 16)). */
#define SARPRO_HIP_SYNTH_NO_WEDGE 1u
#define SARPRO_HIP_SYNTH_NO_BRIGHT 2u
#define SARPRO_HIP_SYNTH_NO_BAND2 4u   /* band 1 (the second) is no-data everywhere */
#define SARPRO_HIP_SYNTH_MAP(m) (((unsigned)(m) & 15u) << 4)
#define SARPRO_HIP_SYNTH_BLOCKS(n) (((unsigned)(n) & 255u) << 8)
int sarpro_hip_synth_scene_u16_dev_ex(sarpro_hip_ctx *ctx, uint64_t seed, int band,
                                      const uint16_t *q_tables_host, size_t rows_total, size_t cols,
                                      size_t row0, size_t rows_local, uint16_t *d_out, size_t pitch, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* SARPRO_HIP_H */

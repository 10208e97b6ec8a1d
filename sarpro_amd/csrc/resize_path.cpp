// resize_path.cpp -- resize_image_data_with_meta (resize.rs:91-236) and the resized dual-pol
// composition (save.rs:317-367 with target_size / pad) on the device: the 400 MP level rasters never
// leave HBM, only the small RGB does.  Lanczos3 arithmetic: see resize_kernels.hip (parity with the
// third-party crate is unpinned); dimension rules and centre padding are exact restatements
// (resize.rs:6-30, padding.rs:5-49).
#include <algorithm>
#include <cstring>

#include "internal.h"
#include "resize_kernels.h"

using namespace sarpro;

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
            return e__ == hipErrorOutOfMemory ? SARPRO_HIP_ERR_OOM : SARPRO_HIP_ERR_HIP;          \
        }                                                                                         \
    } while (0)
#define RETCHK(expr)                                   \
    do {                                               \
        int rc__ = (expr);                             \
        if (rc__ != SARPRO_HIP_OK) return rc__;        \
    } while (0)

static int fail(sarpro_hip_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

extern "C" int sarpro_hip_resize_output_dims(size_t cols, size_t rows, size_t target_size, int pad, size_t *final_cols,
                                             size_t *final_rows) {
    if (!final_cols || !final_rows) return SARPRO_HIP_ERR_INVALID_ARG;
    size_t c = cols, r = rows;
    if (target_size && std::max(cols, rows) != target_size) resize_dimensions(cols, rows, target_size, &c, &r);
    if (pad) c = r = std::max(c, r);
    *final_cols = c; *final_rows = r;
    return SARPRO_HIP_OK;
}

namespace {

// Coefficient tables of one axis on the device, cached on the context by (in, out, element size): the
// two bands of a scene (and every scene of a batch with the same shape) reuse them.
int get_coeffs(sarpro_hip_ctx *ctx, int slot, uint32_t in_size, uint32_t out_size, int elem_size, ResizePassArgs *a) {
    auto &key = ctx->resize_key[slot];
    DevBuf &buf = ctx->resize_coef[slot];
    const size_t n = out_size;
    if (key[0] == in_size && key[1] == out_size && key[2] == (uint32_t)elem_size && buf.p) {
        a->start = buf.as<uint32_t>();
        a->size = buf.as<uint32_t>() + n;
        a->k = reinterpret_cast<const int32_t *>(buf.as<uint8_t>() + n * 8);
        a->in_size = in_size; a->out_size = out_size; a->precision = (int)key[3];
        a->window = key[4]; a->block_span = key[5] & 0x7FFFFFFFu; a->k_small = key[5] >> 31;
        return SARPRO_HIP_OK;
    }
    ResizeCoeffs c;
    build_resize_coeffs(in_size, out_size, elem_size, &c);
    const size_t kb = c.k.size() * sizeof(int32_t);
    const size_t bytes = n * 8 + kb;
    HIPCHK(ctx, buf.reserve(bytes));
    HIPCHK(ctx, ctx->h_upload.reserve(std::max<size_t>(bytes, 2 * 131072 + 2 * 64 * 256 * 8 + 66048 + 1024)));
    uint8_t *h = ctx->h_upload.as<uint8_t>();
    std::memcpy(h, c.start.data(), n * 4);
    std::memcpy(h + n * 4, c.size.data(), n * 4);
    std::memcpy(h + n * 8, c.k.data(), kb);
    HIPCHK(ctx, hipMemcpyAsync(buf.p, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the pinned stage is reused by the next table
    a->start = buf.as<uint32_t>();
    a->size = buf.as<uint32_t>() + n;
    a->k = reinterpret_cast<const int32_t *>(buf.as<uint8_t>() + n * 8);
    a->in_size = c.in_size; a->out_size = c.out_size; a->precision = c.precision;
    uint32_t span = 0;
    for (size_t g = 0; g < n; g += kResizeHBlock) span = std::max(span, c.start[std::min(n, g + kResizeHBlock) - 1] - c.start[g]);
    int32_t kmax = 0;
    for (int32_t v : c.k) kmax = std::max(kmax, v < 0 ? -v : v);
    a->window = c.window; a->block_span = span; a->k_small = kmax < 32768 - 128 ? 1u : 0u;
    key[0] = in_size; key[1] = out_size; key[2] = (uint32_t)elem_size; key[3] = (uint32_t)c.precision; key[4] = c.window; key[5] = span | (a->k_small << 31);
    return SARPRO_HIP_OK;
}

} // namespace

namespace sarpro {

// d_out must hold final_rows x out_pitch elements (see sarpro_hip_resize_output_dims)
int resize_pad_dev(sarpro_hip_ctx *ctx, const void *d_in, size_t cols, size_t rows, size_t in_pitch, size_t target_size,
                   int elem_size, int pad, void *d_out, size_t out_pitch, sarpro_hip_resize_meta *meta, const ResizeLutSrc *lut_src) {
    if (elem_size != 1 && elem_size != 2) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad element size");
    if (lut_src && elem_size != 1) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "a DN table gives u8 levels");
    if (in_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t nc = cols, nr = rows;
    double sx = 1.0, sy = 1.0;
    const bool do_resize = target_size && std::max(cols, rows) != target_size && cols && rows; // resize.rs:110-145
    if (do_resize) resize_dimensions(cols, rows, target_size, &nc, &nr);
    size_t fc = nc, fr = nr, pad_left = 0, pad_top = 0;
    if (pad) { fc = fr = std::max(nc, nr); pad_left = (fc - nc) / 2; pad_top = (fr - nr) / 2; } // padding.rs:12-14
    if (out_pitch < fc) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "out_pitch < final columns");
    if (meta) { meta->final_cols = fc; meta->final_rows = fr; meta->scale_x = sx; meta->scale_y = sy; meta->pad_left = pad_left; meta->pad_top = pad_top; }
    if (!fc || !fr) return SARPRO_HIP_OK;
    if (lut_src && !do_resize) return kResizeLutUnsupported; // (a copy, not a pass: the caller materialises the levels)
    if (lut_src && (!nc || !nr)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "target size collapses a dimension to zero");
    ResizePassArgs ah{}, av{};
    ah.generic = av.generic = ctx->attrs.on(A_RESIZE_GENERIC) ? 1u : 0u;
    if (lut_src) { // probe BEFORE anything is enqueued: the register-resident horizontal pass must take this shape
        RETCHK(get_coeffs(ctx, 0, (uint32_t)cols, (uint32_t)nc, elem_size, &ah));
        if (!resize_h_dot_fits(ah.block_span, ah.window, true, lut_src->lut_cap) || (reinterpret_cast<uintptr_t>(d_in) & 15) != 0 || in_pitch % 16 != 0 ||
            ctx->attrs.on(A_RESIZE_GENERIC))
            return kResizeLutUnsupported;
    }
    uint8_t *out = reinterpret_cast<uint8_t *>(d_out);
    if (pad) HIPCHK(ctx, hipMemset2DAsync(out, out_pitch * elem_size, 0, fc * elem_size, fr, ctx->stream));
    uint8_t *dst = out + (pad_top * out_pitch + pad_left) * elem_size;
    if (!do_resize) {
        if (nc && nr)
            HIPCHK(ctx, hipMemcpy2DAsync(dst, out_pitch * elem_size, d_in, in_pitch * elem_size, nc * elem_size, nr, hipMemcpyDeviceToDevice, ctx->stream));
        return SARPRO_HIP_OK;
    }
    if (!nc || !nr) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "target size collapses a dimension to zero");
    sx = (double)nc / (double)cols; sy = (double)nr / (double)rows; // resize.rs:168-169
    if (meta) { meta->scale_x = sx; meta->scale_y = sy; }
    // horizontal pass -> intermediate (rows x nc), vertical pass -> destination window
    const size_t tmp_pitch = round_up(nc, 64);
    HIPCHK(ctx, ctx->resize_tmp.reserve(rows * tmp_pitch * elem_size));
    RETCHK(get_coeffs(ctx, 0, (uint32_t)cols, (uint32_t)nc, elem_size, &ah));
    RETCHK(get_coeffs(ctx, 1, (uint32_t)rows, (uint32_t)nr, elem_size, &av));
    ah.src = d_in; ah.src_pitch = in_pitch; ah.dst = ctx->resize_tmp.p; ah.dst_pitch = tmp_pitch;
    ah.max_val = elem_size == 1 ? 255u : 65535u;
    av.src = ctx->resize_tmp.p; av.src_pitch = tmp_pitch; av.dst = dst; av.dst_pitch = out_pitch;
    av.width = (uint32_t)nc; av.max_val = ah.max_val;
    {
        KernelTimer t(ctx, "resize_h");
        if (lut_src) {
            const hipError_t e = launch_resize_h_lut(ah, *lut_src, (uint32_t)rows, ctx->stream);
            if (e == hipErrorNotSupported) return fail(ctx, SARPRO_HIP_ERR_HIP, "resize: the table form was probed and then refused"); // (the probe above mirrors the launcher)
            HIPCHK(ctx, e);
        } else HIPCHK(ctx, launch_resize_h(ah, (uint32_t)rows, elem_size, ctx->stream));
    }
    {
        KernelTimer t(ctx, "resize_v");
        HIPCHK(ctx, launch_resize_v(av, elem_size, ctx->stream));
    }
    return SARPRO_HIP_OK; // enqueued: the callers synchronise once, after their last step
}

} // namespace sarpro

extern "C" int sarpro_hip_resize_image_data_dev(sarpro_hip_ctx *ctx, const void *d_data, size_t cols, size_t rows, size_t pitch,
                                                size_t target_size, int bit_depth, int pad, void *d_out, size_t out_pitch,
                                                sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if ((!d_data || !d_out) && cols * rows) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    timing_reset(ctx);
    RETCHK(resize_pad_dev(ctx, d_data, cols, rows, pitch, target_size, bit_depth == SARPRO_BITDEPTH_U8 ? 1 : 2, pad, d_out, out_pitch, meta));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_resize_image_data(sarpro_hip_ctx *ctx, const void *data, size_t cols, size_t rows, size_t target_size,
                                            int bit_depth, int pad, void *out, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if ((!data || !out) && cols * rows) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    timing_reset(ctx);
    const size_t esz = bit_depth == SARPRO_BITDEPTH_U8 ? 1 : 2;
    size_t pitch = 0, fc = 0, fr = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], data, rows, cols, esz, &pitch));
    RETCHK(sarpro_hip_resize_output_dims(cols, rows, target_size, pad, &fc, &fr));
    const size_t opitch = round_up(std::max<size_t>(fc, 1), 64);
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(fr, 1) * opitch * esz));
    sarpro_hip_resize_meta m{};
    RETCHK(resize_pad_dev(ctx, ctx->stage_in[0].p, cols, rows, pitch, target_size, (int)esz, pad, ctx->stage_out[0].p, opitch, &m));
    if (meta) *meta = m;
    return fetch_out_2d(ctx, out, ctx->stage_out[0].p, opitch * esz, fc * esz, fr);
}

// save.rs:317-367 including the resize / pad steps between the per-band autoscale and the composition
// (the reference's order: autoscale -> resize -> pad -> synRGB; the suppressed floor therefore sees
// the zero padding, synthetic_rgb.rs:92-99).  Host u16 bands in, final_rows x final_cols RGB out.
// The same product with the bands arriving through a row reader (streaming ingest): reading, the PCIe upload of
// the previous chunk and -- across bands -- nothing else overlap; the 2048^2 result is small enough for one copy.
static int dualpol_resized_impl(sarpro_hip_ctx *ctx, const uint16_t *const host_bands[2], sarpro_hip_row_reader reader, void *reader_user,
                                size_t rows, size_t cols, int strategy, int mode, size_t target_size, int pad, uint8_t *rgb_out,
                                sarpro_hip_resize_meta *meta, const uint16_t *const dev_bands[2] = nullptr, size_t dev_pitch = 0);

// The same product with the bands and the RGB raster resident in device memory (nothing crosses PCIe): d_rgb_out holds
// final_rows * final_cols * 3 bytes, compact.  Synchronous like the host form.
extern "C" int sarpro_hip_dualpol_synrgb_resized_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows,
                                                         size_t cols, size_t in_pitch, int strategy, int mode, size_t target_size, int pad,
                                                         uint8_t *d_rgb_out, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!d_band1 || !d_band2 || !d_rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (in_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    const uint16_t *none[2] = {nullptr, nullptr}, *dev[2] = {d_band1, d_band2};
    return dualpol_resized_impl(ctx, none, nullptr, nullptr, rows, cols, strategy, mode, target_size, pad, d_rgb_out, meta, dev, in_pitch);
}

extern "C" int sarpro_hip_dualpol_synrgb_resized_stream_u16(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *reader_user,
                                                            size_t rows, size_t cols, int strategy, int mode, size_t target_size,
                                                            int pad, uint8_t *rgb_out, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!reader || (rows * cols && !rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null reader / raster");
    const uint16_t *none[2] = {nullptr, nullptr};
    return dualpol_resized_impl(ctx, none, reader, reader_user, rows, cols, strategy, mode, target_size, pad, rgb_out, meta);
}

extern "C" int sarpro_hip_dualpol_synrgb_resized_u16(sarpro_hip_ctx *ctx, const uint16_t *band1, const uint16_t *band2, size_t rows,
                                                     size_t cols, int strategy, int mode, size_t target_size, int pad,
                                                     uint8_t *rgb_out, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!band1 || !band2 || !rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    const uint16_t *bands[2] = {band1, band2};
    return dualpol_resized_impl(ctx, bands, nullptr, nullptr, rows, cols, strategy, mode, target_size, pad, rgb_out, meta);
}

static int dualpol_resized_impl(sarpro_hip_ctx *ctx, const uint16_t *const bands[2], sarpro_hip_row_reader reader, void *reader_user,
                                size_t rows, size_t cols, int strategy, int mode, size_t target_size, int pad, uint8_t *rgb_out,
                                sarpro_hip_resize_meta *meta, const uint16_t *const dev_bands[2], size_t dev_pitch) {
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    size_t fc = 0, fr = 0;
    RETCHK(sarpro_hip_resize_output_dims(cols, rows, target_size, pad, &fc, &fr));
    const size_t r1 = std::max<size_t>(rows, 1);
    const size_t opitch = round_up(std::max<size_t>(fc, 1), 64);
    sarpro_hip_resize_meta m{};
    TimingHold hold(ctx); // last_kernel_times: every kernel of both bands, the resize passes and the composition
    // Device-resident bands, percentile strategy: BOTH bands' tables from one chain (one histogram launch, one statistics chain, one
    // synchronisation instead of two of each), then the two horizontal passes through them
    ResizeLutSrc both[2]{};
    bool both_done[2] = {false, false};
    if (dev_bands && !ctx->attrs.on(A_NO_RESIZE_LUT) && rows && cols) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        RETCHK(bands_u8_table_dev(ctx, dev_bands, 2, rows, cols, dev_pitch, strategy, 0, both));
        if (both[0].lut && both[1].lut) {
            for (int b = 0; b < 2; ++b) {
                HIPCHK(ctx, ctx->resized[b].reserve(std::max<size_t>(fr, 1) * opitch));
                const int rc = resize_pad_dev(ctx, dev_bands[b], cols, rows, dev_pitch, target_size, 1, pad, ctx->resized[b].p, opitch, &m, &both[b]);
                if (rc == kResizeLutUnsupported) break; // (the same answer for both bands: same shape)
                RETCHK(rc);
                both_done[b] = true;
            }
        }
    }
    for (int b = 0; b < 2; ++b) {
        if (both_done[b]) continue;
        size_t pitch = 0;
        const uint16_t *d_in = nullptr;
        if (dev_bands) {
            HIPCHK(ctx, hipSetDevice(ctx->device));
            pitch = dev_pitch;
            d_in = dev_bands[b];
        } else if (reader) {
            pitch = round_up(std::max<size_t>(cols, 1), 64);
            HIPCHK(ctx, hipSetDevice(ctx->device));
            HIPCHK(ctx, ctx->stage_in[0].reserve(r1 * pitch * 2));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the previous band's kernels have read the staging raster
            RETCHK(stream_upload_band(ctx, reader, reader_user, b, rows, cols, ctx->stage_in[0].as<uint16_t>(), pitch, 0));
        } else {
            RETCHK(stage_in_2d(ctx, ctx->stage_in[0], bands[b], rows, cols, 2, &pitch));
        }
        HIPCHK(ctx, ctx->stage_out[0].reserve(r1 * pitch));
        // per-band u8 at native resolution (pipeline.rs:42; Tamed: autoscale.rs:710 with the band's polarisation)
        const int tamed = strategy == SARPRO_STRATEGY_TAMED ? (b == 0 ? 1 : 2) : 0;
        if (!d_in) d_in = ctx->stage_in[0].as<uint16_t>();
        HIPCHK(ctx, ctx->resized[b].reserve(std::max<size_t>(fr, 1) * opitch));
        // The percentile strategies' autoscale is a DN -> u8 table: the horizontal resize pass reads the DN raster through it and the
        // native-resolution level raster (1 B/px written, 1 B/px read again) never exists.  CLAHE, shapes the register-resident pass
        // does not take, SARPRO_HIP_NO_RESIZE_LUT=1: the level raster, then the u8 passes.
        bool done = false;
        if (!ctx->attrs.on(A_NO_RESIZE_LUT)) {
            ResizeLutSrc ls{};
            RETCHK(band_u8_table_dev(ctx, d_in, rows, cols, pitch, strategy, tamed, &ls));
            if (ls.lut) {
                const int rc = resize_pad_dev(ctx, d_in, cols, rows, pitch, target_size, 1, pad, ctx->resized[b].p, opitch, &m, &ls);
                if (rc != kResizeLutUnsupported) { RETCHK(rc); done = true; }
            }
        }
        if (!done) {
            RETCHK(band_u8_dev(ctx, d_in, rows, cols, pitch, strategy, tamed, ctx->stage_out[0].as<uint8_t>(), pitch));
            RETCHK(resize_pad_dev(ctx, ctx->stage_out[0].p, cols, rows, pitch, target_size, 1, pad, ctx->resized[b].p, opitch, &m));
        }
    }
    if (meta) *meta = m;
    if (!fc || !fr) return SARPRO_HIP_OK;
    // composition on the resized, padded bands through the flat entry point: compacted first when their pitch is not their width
    // (2048 columns: it is), straight into the caller's raster when that lives on the device
    const uint8_t *cb[2] = {ctx->resized[0].as<uint8_t>(), ctx->resized[1].as<uint8_t>()};
    if (opitch != fc) {
        HIPCHK(ctx, ctx->stage_out[1].reserve(fc * fr));
        HIPCHK(ctx, ctx->stage_out[2].reserve(fc * fr));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_out[1].p, fc, ctx->resized[0].p, opitch, fc, fr, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_out[2].p, fc, ctx->resized[1].p, opitch, fc, fr, hipMemcpyDeviceToDevice, ctx->stream));
        cb[0] = ctx->stage_out[1].as<uint8_t>(); cb[1] = ctx->stage_out[2].as<uint8_t>();
    }
    if (dev_bands && (reinterpret_cast<uintptr_t>(rgb_out) & 15) == 0) {
        RETCHK(sarpro_hip_synrgb_u8_dev(ctx, mode, strategy, cb[0], cb[1], fc * fr, rgb_out));
    } else {
        HIPCHK(ctx, ctx->stage_out[0].reserve(std::max(fc * fr * 3, r1 * round_up(std::max<size_t>(cols, 1), 64))));
        RETCHK(sarpro_hip_synrgb_u8_dev(ctx, mode, strategy, cb[0], cb[1], fc * fr, ctx->stage_out[0].as<uint8_t>()));
        HIPCHK(ctx, hipMemcpyAsync(rgb_out, ctx->stage_out[0].p, fc * fr * 3, dev_bands ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// Row stripes of ONE scene -> the resized, padded product ("the Lanczos resize needs a +-3 * scale-row halo per
// stripe").  save.rs:317-367 with the scene's rows spread over the ranks of a communicator: each rank holds rows [row0, row0 +
// rows_local) of both DN rasters (1/N of the scene crossed ITS PCIe link) and produces a contiguous range of the FINAL raster's rows.
//   1. per band the stripe's u8 levels (band_u8_stripe_dev: the statistics, CLAHE bins and level histogram are all-reduced integers,
//      so the levels are the one-piece raster's);
//   2. the horizontal pass over the stripe's rows (rows are independent);
//   3. the halo: output row j of the vertical pass belongs to the rank that holds the CENTRE row of j's window; its window reaches
//      at most `window` rows into the neighbours.  Every rank writes the rows it holds within `window` of each stripe boundary into that
//      boundary's zone of one exchange buffer (zero elsewhere), ONE all-reduce(sum) assembles all zones on all ranks -- each byte has
//      exactly one non-zero contributor, so the sum is a gather and the only collective the library has carries it (integer, exact).
//      (N - 1) zones x 2 window rows x 2 bands x the resized width: 1.8 MB for 400 MP -> 2048^2 on 8 ranks, against 41 MB per band for
//      the whole intermediate raster.  The stripe geometry of all ranks travels the same way first (2 words per rank);
//   4. the vertical pass for the rank's output rows, over its own intermediate rows + the two halos;
//   5. padding + composition of the rank's rows of the final raster: the suppressed variant's floor is taken from the combined
//      histogram of the WHOLE padded product (synthetic_rgb.rs:92-113), summed over the ranks (256 words).
// The rank's rows are returned compact (out_rows x final_cols x 3 bytes) with their position; the caller places them (each rank's D2H
// into one host raster, or a gather).  Bit for bit the raster of sarpro_hip_dualpol_synrgb_resized_u16_dev on the one-piece scene.
// ---------------------------------------------------------------------------------------
namespace {

struct StripeResizeGeom {
    bool do_resize = false;
    size_t nc = 0, nr = 0, fc = 0, fr = 0, pad_left = 0, pad_top = 0;
    size_t j0 = 0, j1 = 0, f0 = 0, f1 = 0; // the rank's rows of the resized raster / of the final raster
    uint32_t window = 0;                   // taps of the vertical pass (0: no resize)
};

// which rows of the product does the holder of input rows [row0, row0 + rows_local) produce?  (pure host arithmetic: every rank, and the
// caller sizing its slice buffer, derive the same answer)
void stripe_resize_geom(size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t target_size, int pad, StripeResizeGeom *g) {
    g->nc = cols; g->nr = rows_total;
    g->do_resize = target_size && std::max(cols, rows_total) != target_size && cols && rows_total; // resize.rs:110-145
    if (g->do_resize) resize_dimensions(cols, rows_total, target_size, &g->nc, &g->nr);
    g->fc = g->nc; g->fr = g->nr;
    if (pad) { g->fc = g->fr = std::max(g->nc, g->nr); g->pad_left = (g->fc - g->nc) / 2; g->pad_top = (g->fr - g->nr) / 2; } // padding.rs:12-14
    const size_t row1 = row0 + rows_local;
    if (g->do_resize && g->nr && g->nc) {
        std::vector<uint32_t> start, size; // (the windows without their weights: 2048 x 60 Lanczos evaluations per call otherwise)
        resize_bounds((uint32_t)rows_total, (uint32_t)g->nr, &g->window, &start, &size);
        auto centre = [&](size_t j) { return (size_t)start[j] + size[j] / 2; };
        size_t j = 0;
        while (j < g->nr && centre(j) < row0) ++j;
        g->j0 = j;
        while (j < g->nr && centre(j) < row1) ++j;
        g->j1 = j;
    } else {
        g->j0 = std::min(row0, g->nr); g->j1 = std::min(row1, g->nr);
    }
    const bool any = g->j1 > g->j0;
    g->f0 = any && g->j0 == 0 ? 0 : g->pad_top + g->j0;             // the holder of the first / last resized row also holds the padding above / below
    g->f1 = any && g->j1 == g->nr ? g->fr : g->pad_top + g->j1;
    if (!any) g->f1 = g->f0;
}

} // namespace

extern "C" int sarpro_hip_stripe_resized_rows(size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t target_size, int pad,
                                              size_t *out_row0, size_t *out_rows, size_t *final_cols, size_t *final_rows) {
    if (row0 + rows_local > rows_total) return SARPRO_HIP_ERR_INVALID_ARG;
    StripeResizeGeom g;
    stripe_resize_geom(rows_total, cols, row0, rows_local, target_size, pad, &g);
    if (out_row0) *out_row0 = g.f0;
    if (out_rows) *out_rows = g.f1 - g.f0;
    if (final_cols) *final_cols = g.fc;
    if (final_rows) *final_rows = g.fr;
    return SARPRO_HIP_OK;
}

// f32 bands (elem_f32; flags: SARPRO_HIP_DUALPOL_*): the levels come from the striped f32 chain (sarpro_hip_stripe_run_f32 =
// process_scalar_data_pipeline at U8 over stripes); under Tamed without the PLAIN_PIPELINE flag from its band-specific form
// (stripe_run_f32_tamed = autoscale_db_image_tamed_synrgb_u8 over stripes: save.rs:324-351).
static int stripe_run_resized_impl(sarpro_hip_ctx *ctx, const void *const d_bands[2], bool elem_f32, unsigned flags, size_t rows_total, size_t cols, size_t row0,
                                   size_t rows_local, size_t in_pitch, int strategy, int mode, size_t target_size, int pad, uint8_t *d_rgb_slice, size_t *out_row0,
                                   size_t *out_rows, sarpro_hip_resize_meta *meta) {
    if (!ctx->comm && !ctx->local_group && !ctx->attrs.on(A_COMM_REPLAY)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "no communicator on this context (sarpro_hip_comm_init / _init_local)");
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (row0 + rows_local > rows_total) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "stripe outside the scene");
    if (rows_local * cols && (!d_bands[0] || !d_bands[1] || in_pitch < cols)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster / pitch < cols");
    comm_replay_rewind(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TimingHold hold(ctx);
    StripeResizeGeom g;
    stripe_resize_geom(rows_total, cols, row0, rows_local, target_size, pad, &g);
    if (g.do_resize && (!g.nc || !g.nr)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "target size collapses a dimension to zero");
    sarpro_hip_resize_meta m{};
    m.final_cols = g.fc; m.final_rows = g.fr; m.pad_left = g.pad_left; m.pad_top = g.pad_top;
    m.scale_x = g.do_resize ? (double)g.nc / (double)cols : 1.0; m.scale_y = g.do_resize ? (double)g.nr / (double)rows_total : 1.0; // resize.rs:168-169
    if (meta) *meta = m;
    if (out_row0) *out_row0 = g.f0;
    if (out_rows) *out_rows = g.f1 - g.f0;
    const int nranks = std::max(ctx->comm_nranks, 1), rank = ctx->comm_rank;
    const size_t K = g.window, row1 = row0 + rows_local;
    const size_t lvl_pitch = round_up(std::max<size_t>(cols, 1), 64), tmp_pitch = round_up(std::max<size_t>(g.nc, 1), 64);
    const size_t tmp_rows = rows_local + 2 * K, tmp_bytes = std::max<size_t>(tmp_rows, 1) * tmp_pitch; // per band: [K halo rows][the stripe's rows][K halo rows]

    // ---- the ranks' stripes (every rank needs every boundary to lay the exchange buffer out)
    HIPCHK(ctx, ctx->resize_geom.reserve(sizeof(uint64_t) * 2 * (size_t)nranks));
    HIPCHK(ctx, hipMemsetAsync(ctx->resize_geom.p, 0, sizeof(uint64_t) * 2 * (size_t)nranks, ctx->stream));
    const uint64_t mine[2] = {(uint64_t)row0 + 1u, (uint64_t)rows_local}; // (+ 1: a rank that never wrote its entry shows as 0)
    HIPCHK(ctx, hipMemcpyAsync(ctx->resize_geom.as<uint64_t>() + 2 * (size_t)rank, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    {
        KernelTimer t(ctx, "allreduce_stripe_geometry");
        RETCHK(comm_allreduce_sum_u64_async(ctx, ctx->resize_geom.as<uint64_t>(), 2 * (size_t)nranks));
    }
    std::vector<uint64_t> geom(2 * (size_t)nranks);
    HIPCHK(ctx, hipMemcpyAsync(geom.data(), ctx->resize_geom.p, sizeof(uint64_t) * geom.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    {
        size_t next = 0;
        for (int r = 0; r < nranks; ++r) { // (every rank sees the same table: the same answer everywhere, nobody is left in a collective)
            if (geom[2 * (size_t)r] != next + 1) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "row stripes must tile the scene in rank order");
            next += geom[2 * (size_t)r + 1];
        }
        if (next != rows_total) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "row stripes must tile the scene in rank order");
    }

    // ---- 1 + 2: levels of the stripe, horizontal pass into the middle of the band's intermediate raster
    // (u16: which chain a stripe takes -- and with it the sequence of collectives -- follows from the layout of its rasters; a stripe that is
    // not in the aligned form is staged through a library raster, an empty one adopts the library's pitch: every rank takes the same route,
    // as sarpro_hip_stripe_run_u16 does)
    const void *band_ptr[2] = {d_bands[0], d_bands[1]};
    if (!elem_f32) {
        const bool empty = rows_local == 0 || cols == 0;
        const bool aligned = !empty && in_pitch % 16 == 0 && (reinterpret_cast<uintptr_t>(d_bands[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_bands[1]) & 15) == 0;
        if (empty) in_pitch = lvl_pitch;
        else if (!aligned) {
            for (int b = 0; b < 2; ++b) {
                HIPCHK(ctx, ctx->stage_in[b].reserve(rows_local * lvl_pitch * sizeof(uint16_t)));
                HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_in[b].p, lvl_pitch * sizeof(uint16_t), d_bands[b], in_pitch * sizeof(uint16_t), cols * sizeof(uint16_t), rows_local,
                                             hipMemcpyDeviceToDevice, ctx->stream));
                band_ptr[b] = ctx->stage_in[b].p;
            }
            in_pitch = lvl_pitch;
        }
    }
    HIPCHK(ctx, ctx->resize_tmp.reserve(2 * tmp_bytes));
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(rows_local, 1) * lvl_pitch));
    ResizePassArgs ah{}, av{};
    ah.generic = av.generic = ctx->attrs.on(A_RESIZE_GENERIC) ? 1u : 0u;
    if (g.do_resize) {
        RETCHK(get_coeffs(ctx, 0, (uint32_t)cols, (uint32_t)g.nc, 1, &ah));
        RETCHK(get_coeffs(ctx, 1, (uint32_t)rows_total, (uint32_t)g.nr, 1, &av));
        if (av.window != g.window) return fail(ctx, SARPRO_HIP_ERR_HIP, "striped resize: the halo was sized for another window than the vertical pass's");
    }
    for (int b = 0; b < 2; ++b) {
        uint8_t *tmp = ctx->resize_tmp.as<uint8_t>() + (size_t)b * tmp_bytes, *mid = tmp + K * tmp_pitch;
        const int tamed = strategy == SARPRO_STRATEGY_TAMED ? (b == 0 ? 1 : 2) : 0; // save.rs:324-351
        uint8_t *lvl = g.do_resize ? ctx->stage_out[0].as<uint8_t>() : mid;        // (no resize: the levels ARE the intermediate raster)
        if (elem_f32 && tamed && !(flags & SARPRO_HIP_DUALPOL_PLAIN_PIPELINE)) {
            RETCHK(stripe_run_f32_tamed(ctx, reinterpret_cast<const float *>(band_ptr[b]), rows_total, cols, row0, rows_local, in_pitch, tamed, lvl, g.do_resize ? lvl_pitch : tmp_pitch));
        } else if (elem_f32) {
            RETCHK(sarpro_hip_stripe_run_f32(ctx, reinterpret_cast<const float *>(band_ptr[b]), rows_total, cols, row0, rows_local, in_pitch, strategy, SARPRO_BITDEPTH_U8, lvl,
                                             g.do_resize ? lvl_pitch : tmp_pitch, nullptr));
        } else {
            RETCHK(band_u8_stripe_dev(ctx, reinterpret_cast<const uint16_t *>(band_ptr[b]), rows_total, cols, row0, rows_local, in_pitch, strategy, tamed, lvl,
                                      g.do_resize ? lvl_pitch : tmp_pitch));
        }
        if (g.do_resize && rows_local) {
            ah.src = lvl; ah.src_pitch = lvl_pitch; ah.dst = mid; ah.dst_pitch = tmp_pitch; ah.max_val = 255u;
            KernelTimer t(ctx, "resize_h");
            HIPCHK(ctx, launch_resize_h(ah, (uint32_t)rows_local, 1, ctx->stream));
        }
    }

    // ---- 3: the halo rows, through the boundary zones
    if (g.do_resize && nranks > 1) {
        const size_t zone_bytes = 2 * K * tmp_pitch, nz = (size_t)nranks - 1, xbytes = 2 * nz * zone_bytes; // [band][zone][2 K rows][tmp_pitch]
        HIPCHK(ctx, ctx->resize_halo.reserve(xbytes));
        uint8_t *x = ctx->resize_halo.as<uint8_t>();
        HIPCHK(ctx, hipMemsetAsync(x, 0, xbytes, ctx->stream));
        for (size_t z = 0; z < nz; ++z) { // boundary between rank z and z + 1, at row B: the zone is rows [B - K, B + K)
            const long long B = (long long)geom[2 * (z + 1)] - 1, lo = std::max<long long>(B - (long long)K, (long long)row0), hi = std::min<long long>(B + (long long)K, (long long)row1);
            if (hi <= lo) continue; // none of this rank's rows lie in the zone
            for (int b = 0; b < 2; ++b) {
                const uint8_t *mid = ctx->resize_tmp.as<uint8_t>() + (size_t)b * tmp_bytes + K * tmp_pitch;
                HIPCHK(ctx, hipMemcpyAsync(x + ((size_t)b * nz + z) * zone_bytes + (size_t)(lo - (B - (long long)K)) * tmp_pitch, mid + (size_t)(lo - (long long)row0) * tmp_pitch,
                                           (size_t)(hi - lo) * tmp_pitch, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        {
            KernelTimer t(ctx, "allreduce_resize_halo");
            RETCHK(comm_allreduce_sum_u64_async(ctx, reinterpret_cast<uint64_t *>(x), xbytes / 8));
        }
        if (g.j1 > g.j0) // rows [row0 - K, row0) from the zone at this rank's upper boundary, rows [row1, row1 + K) from the one at its lower boundary
            for (int b = 0; b < 2; ++b) {
                uint8_t *tmp = ctx->resize_tmp.as<uint8_t>() + (size_t)b * tmp_bytes;
                if (rank > 0) HIPCHK(ctx, hipMemcpyAsync(tmp, x + ((size_t)b * nz + (size_t)rank - 1) * zone_bytes, K * tmp_pitch, hipMemcpyDeviceToDevice, ctx->stream));
                if (rank + 1 < nranks)
                    HIPCHK(ctx, hipMemcpyAsync(tmp + (K + rows_local) * tmp_pitch, x + ((size_t)b * nz + (size_t)rank) * zone_bytes + K * tmp_pitch, K * tmp_pitch, hipMemcpyDeviceToDevice, ctx->stream));
            }
    }

    // ---- 4: the rank's rows of the resized bands, inside its rows of the padded bands
    const size_t srows = g.f1 - g.f0, opitch = round_up(std::max<size_t>(g.fc, 1), 64);
    for (int b = 0; b < 2; ++b) {
        HIPCHK(ctx, ctx->resized[b].reserve(std::max<size_t>(srows, 1) * opitch));
        if (srows) HIPCHK(ctx, hipMemsetAsync(ctx->resized[b].p, 0, srows * opitch, ctx->stream));
        if (g.j1 <= g.j0) continue;
        const uint8_t *tmp = ctx->resize_tmp.as<uint8_t>() + (size_t)b * tmp_bytes;
        uint8_t *dst = ctx->resized[b].as<uint8_t>() + g.pad_left; // row (pad_top + j - f0) of the slice holds resized row j
        if (g.do_resize) {
            av.src = tmp; av.src_pitch = tmp_pitch; av.src_row0 = (int32_t)((long long)row0 - (long long)K);
            av.dst = dst; av.dst_pitch = opitch; av.dst_row0 = (int32_t)((long long)g.f0 - (long long)g.pad_top);
            av.width = (uint32_t)g.nc; av.max_val = 255u; av.oy0 = (uint32_t)g.j0; av.oy_n = (uint32_t)(g.j1 - g.j0);
            KernelTimer t(ctx, "resize_v");
            HIPCHK(ctx, launch_resize_v(av, 1, ctx->stream));
        } else {
            HIPCHK(ctx, hipMemcpy2DAsync(dst + (g.pad_top + g.j0 - g.f0) * opitch, opitch, tmp + (K + (g.j0 - row0)) * tmp_pitch, tmp_pitch, g.nc, g.j1 - g.j0,
                                         hipMemcpyDeviceToDevice, ctx->stream));
        }
    }

    // ---- 5: composition of the rank's rows; the suppressed floor from the whole padded product's histogram
    const uint8_t *cb[2] = {ctx->resized[0].as<uint8_t>(), ctx->resized[1].as<uint8_t>()};
    if (opitch != g.fc && srows) {
        HIPCHK(ctx, ctx->stage_out[1].reserve(g.fc * srows));
        HIPCHK(ctx, ctx->stage_out[2].reserve(g.fc * srows));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_out[1].p, g.fc, ctx->resized[0].p, opitch, g.fc, srows, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_out[2].p, g.fc, ctx->resized[1].p, opitch, g.fc, srows, hipMemcpyDeviceToDevice, ctx->stream));
        cb[0] = ctx->stage_out[1].as<uint8_t>(); cb[1] = ctx->stage_out[2].as<uint8_t>();
    }
    const size_t n_local = g.fc * srows;
    if (n_local && !d_rgb_slice) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null RGB slice");
    if (!g.fc || !g.fr) return SARPRO_HIP_OK;
    uint8_t *rgb = d_rgb_slice;
    if (n_local && (reinterpret_cast<uintptr_t>(d_rgb_slice) & 15) != 0) { // (the vector composition wants 16-byte alignment: through a library raster then)
        HIPCHK(ctx, ctx->stage_out[0].reserve(std::max(n_local * 3, std::max<size_t>(rows_local, 1) * lvl_pitch)));
        rgb = ctx->stage_out[0].as<uint8_t>();
    }
    RETCHK(synrgb_flat_dev(ctx, mode, strategy, cb[0], cb[1], n_local, g.fc * g.fr, true, rgb));
    if (rgb != d_rgb_slice && n_local) HIPCHK(ctx, hipMemcpyAsync(d_rgb_slice, rgb, n_local * 3, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_run_resized_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total, size_t cols,
                                                 size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode, size_t target_size, int pad,
                                                 uint8_t *d_rgb_slice, size_t *out_row0, size_t *out_rows, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const void *const bands[2] = {d_band1, d_band2};
    const int rc = stripe_run_resized_impl(ctx, bands, false, 0u, rows_total, cols, row0, rows_local, in_pitch, strategy, mode, target_size, pad, d_rgb_slice,
                                           out_row0, out_rows, meta);
    if (rc != SARPRO_HIP_OK) comm_abort_local_group(ctx); // an in-process group: the peers of a rank that failed must not wait for it (comm.cpp)
    return rc;
}

extern "C" int sarpro_hip_stripe_run_resized_f32(sarpro_hip_ctx *ctx, const float *d_band1, const float *d_band2, size_t rows_total, size_t cols, size_t row0,
                                                 size_t rows_local, size_t in_pitch, int strategy, int mode, unsigned flags, size_t target_size, int pad,
                                                 uint8_t *d_rgb_slice, size_t *out_row0, size_t *out_rows, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const void *const bands[2] = {d_band1, d_band2};
    const int rc = stripe_run_resized_impl(ctx, bands, true, flags, rows_total, cols, row0, rows_local, in_pitch, strategy, mode, target_size, pad, d_rgb_slice,
                                           out_row0, out_rows, meta);
    if (rc != SARPRO_HIP_OK) comm_abort_local_group(ctx);
    return rc;
}

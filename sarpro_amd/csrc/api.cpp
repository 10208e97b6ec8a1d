// api.cpp -- C ABI of libsarpro_hip.so: context, work planning and the pass orchestration of
// the u16 (integer-DN) flavour.  See include/sarpro_hip.h for the reference functions each
// entry point replaces.  No CPU fallback exists anywhere in this file: every raster result is
// produced by the kernels in kernels.hip.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>

#include "api_common.h"
#include "chain_kernels.h"
#include "context.h"
#include "internal.h"
#include "resize_kernels.h"
#include "u16_job.h"

using namespace sarpro;

// ---------------------------------------------------------------------------------------
// device-pointer entry points (u16)
// ---------------------------------------------------------------------------------------
extern "C" int sarpro_hip_autoscale_band_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols,
                                                 size_t in_pitch, int strategy, int bit_depth, void *d_out,
                                                 size_t out_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!d_in || !d_out) && rows * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (out_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "out_pitch < cols");
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = d_in;
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = bit_depth;
    void *outs[kMaxBands] = {d_out, nullptr};
    return job_run_all(J, outs, out_pitch, nullptr, 0, stats_out);
}

namespace sarpro {
// nb = 1: one band (tamed: 0 / 1 copol / 2 crosspol, as band_u8_dev).  nb = 2: the two bands of a dual-pol product in ONE chain
// (one histogram pass, one statistics chain, one synchronisation; strategy Tamed: band 0 copol, band 1 crosspol; `tamed` unused).
int bands_u8_table_dev(sarpro_hip_ctx *ctx, const uint16_t *const d_in[], int nb, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed,
                       ResizeLutSrc *out) {
    for (int b = 0; b < nb; ++b) { out[b].lut = nullptr; out[b].dev_state = nullptr; out[b].band = b; out[b].lut_cap = 0; }
    U16Job J;
    J.ctx = ctx; J.nbands = nb;
    for (int b = 0; b < nb; ++b) J.d_in[b] = d_in[b];
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = (nb == 1 && tamed) ? SARPRO_STRATEGY_TAMED : strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.tamed_force = nb == 1 ? tamed : 0;
    J.tables_only = true;
    timing_reset(ctx);
    RETCHK(job_init(J));
    void *outs[kMaxBands] = {nullptr, nullptr};
    if (!rows || !cols || !chain_levels_eligible(J, outs, 0, nullptr, 0)) return SARPRO_HIP_OK;
    HostTimer t(ctx, "host:chain(enqueue+final sync)");
    const int rc = job_run_chain_levels(J, outs, 0, nullptr, 0, nullptr);
    if (rc == kRerunOnHostRoute) return SARPRO_HIP_OK;
    RETCHK(rc);
    for (int b = 0; b < nb; ++b) {
        out[b].lut = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        out[b].dev_state = reinterpret_cast<const ChainBandState *>(ctx->chain_state.as<uint8_t>());
        out[b].lut_cap = (ctx->chain_levels_cap + 15u) & ~15u;
    }
    return SARPRO_HIP_OK;
}
int band_u8_table_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed, ResizeLutSrc *out) {
    const uint16_t *const in[1] = {d_in};
    return bands_u8_table_dev(ctx, in, 1, rows, cols, in_pitch, strategy, tamed, out);
}

int band_u8_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed,
                uint8_t *d_out, size_t out_pitch) {
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = d_in;
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = tamed ? SARPRO_STRATEGY_TAMED : strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.tamed_force = tamed;
    void *outs[kMaxBands] = {d_out, nullptr};
    return job_run_all(J, outs, out_pitch, nullptr, 0, nullptr);
}
int band_u8_stripe_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch,
                       int strategy, int tamed, uint8_t *d_out, size_t out_pitch) {
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = d_in;
    J.rows_total = rows_total; J.row0 = row0; J.rows_local = rows_local; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = tamed ? SARPRO_STRATEGY_TAMED : strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.tamed_force = tamed;
    J.reduce = true;
    void *outs[kMaxBands] = {d_out, nullptr};
    return job_run_all(J, outs, out_pitch, nullptr, 0, nullptr);
}
} // namespace sarpro

extern "C" int sarpro_hip_dualpol_synrgb_u16_dev(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2,
                                                 size_t rows, size_t cols, size_t in_pitch, int strategy, int mode,
                                                 uint8_t *d_rgb, size_t rgb_pitch_px, uint8_t *d_u8_band1,
                                                 uint8_t *d_u8_band2, size_t u8_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!d_band1 || !d_band2 || !d_rgb) && rows * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (rgb_pitch_px < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "rgb_pitch_px < cols");
    if ((d_u8_band1 == nullptr) != (d_u8_band2 == nullptr))
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pass both per-band u8 outputs or neither");
    if (d_u8_band1 && u8_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "u8_pitch < cols");
    U16Job J;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = d_band1; J.d_in[1] = d_band2;
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true;
    J.allow_async = true; // SARPRO_HIP_CTX_ASYNC_DEV applies to this entry point only
    void *outs[kMaxBands] = {d_u8_band1, d_u8_band2};
    return job_run_all(J, outs, u8_pitch, d_rgb, rgb_pitch_px, stats_out);
}

extern "C" int sarpro_hip_polop_f32_dev(sarpro_hip_ctx *ctx, int op, const float *d_a, const float *d_b, size_t n,
                                        float *d_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (op < SARPRO_OP_SUM || op > SARPRO_OP_LOGRATIO) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarization operation");
    if ((!d_a || !d_b || !d_out) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    timing_reset(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        KernelTimer t(ctx, "polop_f32");
        HIPCHK(ctx, launch_polop_f32(op, d_a, d_b, n, d_out, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_synrgb_u8_dev(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *d_b1,
                                        const uint8_t *d_b2, size_t n, uint8_t *d_rgb) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if ((!d_b1 || !d_b2 || !d_rgb) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    timing_reset(ctx);
    if (n == 0) return SARPRO_HIP_OK;
    return sarpro::synrgb_flat_dev(ctx, mode, strategy, d_b1, d_b2, n, n, false, d_rgb);
}

// The composition of n pixels that are a PART of a product of n_total pixels (`reduce`: the other parts are other ranks' -- the
// suppressed variant's combined histogram is summed over the ranks before the floor is taken, synthetic_rgb.rs:92-113; every rank calls,
// also one with n = 0).  n == n_total, reduce false: the whole product (sarpro_hip_synrgb_u8_dev).
int sarpro::synrgb_flat_dev(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *d_b1, const uint8_t *d_b2, size_t n, size_t n_total,
                            bool reduce, uint8_t *d_rgb) {
    (void)mode; // synthetic_rgb.rs:72-79: the mode is ignored
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // view the flat arrays as rows of 4096 px (+ one short row) so the 2-D kernels apply
    const size_t W = 4096;
    const size_t full_rows = n / W, tail = n - full_rows * W;
    const bool suppressed = strategy == SARPRO_STRATEGY_TAMED || strategy == SARPRO_STRATEGY_CLAHE;
    std::vector<uint8_t> luts(66048), tables(66048);
    int fwc = -1;
    if (suppressed) {
        HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands));
        HIPCHK(ctx, hipMemsetAsync(ctx->level_hist.p, 0, sizeof(uint64_t) * 256, ctx->stream));
        for (const uint8_t *p : {d_b1, d_b2}) {
            KernelTimer t(ctx, "hist256_u8");
            if (full_rows) HIPCHK(ctx, launch_hist256_u8(p, W, (uint32_t)full_rows, (uint32_t)W, ctx->level_hist.as<unsigned long long>(), ctx->stream));
            if (tail) HIPCHK(ctx, launch_hist256_u8(p + full_rows * W, W, 1, (uint32_t)tail, ctx->level_hist.as<unsigned long long>(), ctx->stream));
        }
        if (reduce) RETCHK(comm_allreduce_sum_u64_async(ctx, ctx->level_hist.as<uint64_t>(), 256));
        uint64_t h[256];
        HIPCHK(ctx, hipMemcpyAsync(h, ctx->level_hist.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        fwc = synrgb_floor_from_hist(h, n_total);
        synrgb_luts_suppressed(fwc, luts.data());
    } else {
        synrgb_luts_default(luts.data());
    }
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    fold_compose_tables(luts.data(), fwc, ident, ident, tables.data());
    HIPCHK(ctx, ctx->tables.reserve(66048 + 512));
    HIPCHK(ctx, ctx->h_upload.reserve(2 * 131072 + 2 * 64 * 256 * 8 + 66048 + 1024));
    std::memcpy(ctx->h_upload.p, tables.data(), 66048); // pinned and the context's own: the copy needs no wait of its own
    HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, ctx->h_upload.p, 66048, hipMemcpyHostToDevice, ctx->stream));
    const bool al = ptr_aligned16(d_b1) && ptr_aligned16(d_b2) && ptr_aligned16(d_rgb);
    ComposeArgs c{};
    c.tables = ctx->tables.as<uint8_t>();
    c.in_pitch = W; c.rgb_pitch_px = W;
    if (full_rows) {
        c.b1 = d_b1; c.b2 = d_b2; c.rgb = d_rgb; c.rows = (uint32_t)full_rows; c.cols = (uint32_t)W;
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, al ? 16 : 1, ctx->stream));
    }
    if (tail) {
        c.b1 = d_b1 + full_rows * W; c.b2 = d_b2 + full_rows * W; c.rgb = d_rgb + full_rows * W * 3;
        c.rows = 1; c.cols = (uint32_t)tail;
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, al ? 16 : 1, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// host-pointer entry points: stage through pitched device buffers, run the device path
// ---------------------------------------------------------------------------------------
namespace sarpro {

int stage_in_2d(sarpro_hip_ctx *ctx, DevBuf &buf, const void *host, size_t rows, size_t cols, size_t esz, size_t *pitch_elems) {
    const size_t pitch = round_up(std::max<size_t>(cols, 1), 64);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, buf.reserve(std::max<size_t>(rows, 1) * pitch * esz));
    if (rows && cols)
        HIPCHK(ctx, hipMemcpy2DAsync(buf.p, pitch * esz, host, cols * esz, cols * esz, rows, hipMemcpyHostToDevice, ctx->stream));
    *pitch_elems = pitch;
    return SARPRO_HIP_OK;
}

int fetch_out_2d(sarpro_hip_ctx *ctx, void *host, const void *dev, size_t pitch_bytes, size_t row_bytes, size_t rows) {
    if (rows && row_bytes) {
        HIPCHK(ctx, hipMemcpy2DAsync(host, row_bytes, dev, pitch_bytes, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SARPRO_HIP_OK;
}

} // namespace sarpro

static int host_band_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols, int strategy, int bit_depth,
                         int tamed_force, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const bool u8o = tamed_force || bit_depth == SARPRO_BITDEPTH_U8;
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if (rows * cols && (!in || (u8o ? (void *)out_u8 : (void *)out_u16) == nullptr)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    size_t pitch = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], in, rows, cols, 2, &pitch));
    const size_t osz = u8o ? 1 : 2;
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(rows, 1) * pitch * osz));
    U16Job J;
    J.ctx = ctx; J.nbands = 1; J.d_in[0] = ctx->stage_in[0].as<uint16_t>();
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = pitch;
    J.strategy = tamed_force ? SARPRO_STRATEGY_TAMED : strategy;
    J.bit_depth = u8o ? SARPRO_BITDEPTH_U8 : SARPRO_BITDEPTH_U16;
    J.tamed_force = tamed_force;
    if (!tamed_force && (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    void *outs[kMaxBands] = {ctx->stage_out[0].p, nullptr};
    RETCHK(job_run_all(J, outs, pitch, nullptr, 0, stats_out));
    return fetch_out_2d(ctx, u8o ? (void *)out_u8 : (void *)out_u16, ctx->stage_out[0].p, pitch * osz, cols * osz, rows);
}

// ---------------------------------------------------------------------------------------
// Streaming ingest / egress: the two u16 bands arrive through a row-chunk reader (a GDAL
// RasterIO loop, the strip-TIFF reader of tiff_io.cpp, ...) into a ring of pinned buffers; each chunk goes to
// the device with hipMemcpy2DAsync on a side stream while the reader fills the next one, and the DN-histogram
// work items whose rows have arrived run on the compute stream behind an event -- when the last chunk lands
// the first pass of the chain is already done.  The RGB leaves the same way, chunk by chunk, to a row sink.
// ---------------------------------------------------------------------------------------
static int stream_prepare(sarpro_hip_ctx *ctx, size_t ring_bytes) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : ctx->ring_evt)
        if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(ctx, ctx->h_ring.reserve(ring_bytes));
    return SARPRO_HIP_OK;
}

namespace sarpro {
int stream_upload_band(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *user, int band, size_t rows, size_t cols,
                       uint16_t *d_dst, size_t pitch, size_t chunk_rows) {
    if (!reader) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null reader");
    if (rows * cols == 0) return SARPRO_HIP_OK;
    constexpr int kSlots = 3;
    if (!chunk_rows) chunk_rows = std::max<size_t>(16, (32u << 20) / (pitch * 2));
    chunk_rows = std::min(chunk_rows, rows);
    const size_t slot_bytes = chunk_rows * pitch * sizeof(uint16_t);
    RETCHK(stream_prepare(ctx, kSlots * slot_bytes));
    uint8_t *ring = ctx->h_ring.as<uint8_t>();
    size_t chunk = 0;
    for (size_t r0 = 0; r0 < rows; r0 += chunk_rows, ++chunk) {
        const int slot = (int)(chunk % kSlots);
        const size_t n = std::min(chunk_rows, rows - r0);
        if (chunk >= (size_t)kSlots) HIPCHK(ctx, hipEventSynchronize(ctx->ring_evt[slot]));
        uint16_t *h = reinterpret_cast<uint16_t *>(ring + slot * slot_bytes);
        {
            HostTimer t(ctx, "host:reader");
            if (int rc = reader(user, band, r0, n, h, pitch)) {
                (void)hipStreamSynchronize(ctx->copy_stream);
                ctx->err = "row reader failed (code " + std::to_string(rc) + ")";
                return SARPRO_HIP_ERR_IO;
            }
        }
        HIPCHK(ctx, hipMemcpyAsync(d_dst + r0 * pitch, h, n * pitch * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ring_evt[slot], ctx->copy_stream));
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_evt[(chunk - 1) % kSlots], 0)); // the compute stream sees the band
    return SARPRO_HIP_OK;
}
} // namespace sarpro

extern "C" int sarpro_hip_dualpol_synrgb_stream_u16(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *reader_user, size_t rows,
                                                    size_t cols, int strategy, int mode, size_t chunk_rows, sarpro_hip_row_sink sink,
                                                    void *sink_user, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!reader || !sink) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null reader / sink");
    timing_reset(ctx);
    if (rows * cols == 0) return SARPRO_HIP_OK;
    constexpr int kSlots = 3;
    const size_t pitch = round_up(cols, 64);
    if (!chunk_rows) chunk_rows = std::max<size_t>(16, (32u << 20) / (pitch * 2 * 2)); // ~32 MiB per slot (both bands)
    chunk_rows = std::min(chunk_rows, rows);
    const size_t in_slot = 2 * chunk_rows * pitch * sizeof(uint16_t), out_chunk_rows = std::max<size_t>(1, in_slot / (pitch * 3));
    const size_t slot_bytes = std::max(in_slot, out_chunk_rows * pitch * 3);
    RETCHK(stream_prepare(ctx, kSlots * slot_bytes));
    HIPCHK(ctx, ctx->stage_in[0].reserve(rows * pitch * 2));
    HIPCHK(ctx, ctx->stage_in[1].reserve(rows * pitch * 2));
    HIPCHK(ctx, ctx->stage_out[0].reserve(rows * pitch * 3));
    U16Job J;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = ctx->stage_in[0].as<uint16_t>(); J.d_in[1] = ctx->stage_in[1].as<uint16_t>();
    J.rows_total = J.rows_local = rows; J.cols = cols; J.in_pitch = pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true;
    RETCHK(job_init(J));

    // ---- ingest: reader -> pinned slot -> device, histogram items behind each chunk ----
    const std::vector<Rect> &items = J.clahe() ? J.plan->hist_rects_tiled : J.plan->hist_rects_flat;
    bool sorted = true;
    for (size_t i = 1; i < items.size() && sorted; ++i) sorted = items[i - 1].r0 <= items[i].r0;
    int issued = 0;
    bool begun = false;
    uint8_t *ring = ctx->h_ring.as<uint8_t>();
    size_t chunk = 0;
    for (size_t r0 = 0; r0 < rows; r0 += chunk_rows, ++chunk) {
        const int slot = (int)(chunk % kSlots);
        const size_t n = std::min(chunk_rows, rows - r0);
        if (chunk >= (size_t)kSlots) HIPCHK(ctx, hipEventSynchronize(ctx->ring_evt[slot])); // the slot's previous upload has left it
        uint16_t *h[2] = {reinterpret_cast<uint16_t *>(ring + slot * slot_bytes), reinterpret_cast<uint16_t *>(ring + slot * slot_bytes) + chunk_rows * pitch};
        {
            HostTimer t(ctx, "host:reader");
            for (int b = 0; b < 2; ++b)
                if (int rc = reader(reader_user, b, r0, n, h[b], pitch)) {
                    (void)hipStreamSynchronize(ctx->copy_stream);
                    (void)hipStreamSynchronize(ctx->stream);
                    ctx->err = "row reader failed (code " + std::to_string(rc) + ")";
                    return SARPRO_HIP_ERR_IO;
                }
        }
        for (int b = 0; b < 2; ++b)
            HIPCHK(ctx, hipMemcpyAsync(const_cast<uint16_t *>(J.d_in[b]) + r0 * pitch, h[b], n * pitch * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ring_evt[slot], ctx->copy_stream));
        if (sorted) {
            int last = issued;
            while (last < (int)items.size() && (size_t)items[last].r1 <= r0 + n) ++last;
            if (last > issued || !begun) {
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_evt[slot], 0));
                RETCHK(job_phase1(J, !begun, issued, last, false));
                begun = true;
                issued = last;
            }
        }
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_evt[(chunk - 1) % kSlots], 0));
    RETCHK(job_phase1(J, !begun, issued, -1, true)); // whatever is left, then the band histograms
    J.hist_done = true;

    // ---- the rest of the chain, device-resident ----
    void *outs[kMaxBands] = {nullptr, nullptr};
    uint8_t *d_rgb = ctx->stage_out[0].as<uint8_t>();
    {
        HostTimer t(ctx, "host:chain(enqueue+final sync)");
        if (chain_eligible(J)) RETCHK(job_run_chain(J, outs, 0, d_rgb, pitch, stats_out));
        else if (chain_levels_eligible(J, outs, 0, d_rgb, pitch)) RETCHK(job_run_chain_levels(J, outs, 0, d_rgb, pitch, stats_out));
        else return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "streaming ingest needs the device chain (SARPRO_HIP_NO_CHAIN is set)");
    }

    // ---- egress: device -> pinned slot -> sink, the copy of chunk k+1 under the sink of chunk k ----
    size_t ochunk = 0;
    const size_t nout = (rows + out_chunk_rows - 1) / out_chunk_rows;
    auto enqueue_out = [&](size_t k) -> int {
        const size_t r0 = k * out_chunk_rows, n = std::min(out_chunk_rows, rows - r0);
        HIPCHK(ctx, hipMemcpyAsync(ring + (k % kSlots) * slot_bytes, d_rgb + r0 * pitch * 3, n * pitch * 3, hipMemcpyDeviceToHost, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ring_evt[k % kSlots], ctx->copy_stream));
        return SARPRO_HIP_OK;
    };
    for (size_t k = 0; k < std::min<size_t>(kSlots - 1, nout); ++k) RETCHK(enqueue_out(k));
    for (; ochunk < nout; ++ochunk) {
        if (ochunk + kSlots - 1 < nout) RETCHK(enqueue_out(ochunk + kSlots - 1));
        HIPCHK(ctx, hipEventSynchronize(ctx->ring_evt[ochunk % kSlots]));
        const size_t r0 = ochunk * out_chunk_rows, n = std::min(out_chunk_rows, rows - r0);
        HostTimer t(ctx, "host:sink");
        if (int rc = sink(sink_user, r0, n, ring + (ochunk % kSlots) * slot_bytes, pitch * 3)) {
            (void)hipStreamSynchronize(ctx->copy_stream);
            ctx->err = "row sink failed (code " + std::to_string(rc) + ")";
            return SARPRO_HIP_ERR_IO;
        }
    }
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_autoscale_band_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols,
                                             int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16,
                                             sarpro_hip_stats *stats_out) {
    return host_band_u16(ctx, in, rows, cols, strategy, bit_depth, 0, out_u8, out_u16, stats_out);
}

extern "C" int sarpro_hip_tamed_synrgb_u8_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols,
                                              int is_copol, uint8_t *out_u8) {
    return host_band_u16(ctx, in, rows, cols, SARPRO_STRATEGY_TAMED, SARPRO_BITDEPTH_U8,
                         is_copol ? kTamedCopol : kTamedCrosspol, out_u8, nullptr, nullptr);
}

extern "C" int sarpro_hip_dualpol_synrgb_u16(sarpro_hip_ctx *ctx, const uint16_t *band1, const uint16_t *band2,
                                             size_t rows, size_t cols, int strategy, int mode, uint8_t *rgb_out,
                                             uint8_t *u8_band1, uint8_t *u8_band2, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!band1 || !band2 || !rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    size_t pitch = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], band1, rows, cols, 2, &pitch));
    RETCHK(stage_in_2d(ctx, ctx->stage_in[1], band2, rows, cols, 2, &pitch));
    const size_t r1 = std::max<size_t>(rows, 1);
    HIPCHK(ctx, ctx->stage_out[0].reserve(r1 * pitch * 3));
    const bool want_u8 = u8_band1 || u8_band2;
    if (want_u8) { HIPCHK(ctx, ctx->stage_out[1].reserve(r1 * pitch)); HIPCHK(ctx, ctx->stage_out[2].reserve(r1 * pitch)); }
    int rc = sarpro_hip_dualpol_synrgb_u16_dev(ctx, ctx->stage_in[0].as<uint16_t>(), ctx->stage_in[1].as<uint16_t>(), rows, cols,
                                               pitch, strategy, mode, ctx->stage_out[0].as<uint8_t>(), pitch,
                                               want_u8 ? ctx->stage_out[1].as<uint8_t>() : nullptr,
                                               want_u8 ? ctx->stage_out[2].as<uint8_t>() : nullptr, pitch, stats_out);
    if (rc) return rc;
    RETCHK(fetch_out_2d(ctx, rgb_out, ctx->stage_out[0].p, pitch * 3, cols * 3, rows));
    if (u8_band1) RETCHK(fetch_out_2d(ctx, u8_band1, ctx->stage_out[1].p, pitch, cols, rows));
    if (u8_band2) RETCHK(fetch_out_2d(ctx, u8_band2, ctx->stage_out[2].p, pitch, cols, rows));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_polop_f32(sarpro_hip_ctx *ctx, int op, const float *a, const float *b, size_t n, float *out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (op < SARPRO_OP_SUM || op > SARPRO_OP_LOGRATIO) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarization operation");
    if ((!a || !b || !out) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    if (!n) return SARPRO_HIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->stage_in[0].reserve(n * 4));
    HIPCHK(ctx, ctx->stage_in[1].reserve(n * 4));
    HIPCHK(ctx, ctx->stage_out[0].reserve(n * 4));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[0].p, a, n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[1].p, b, n * 4, hipMemcpyHostToDevice, ctx->stream));
    RETCHK(sarpro_hip_polop_f32_dev(ctx, op, ctx->stage_in[0].as<float>(), ctx->stage_in[1].as<float>(), n, ctx->stage_out[0].as<float>()));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_out[0].p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_synrgb_u8(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *band1, const uint8_t *band2,
                                    size_t n, uint8_t *rgb_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!band1 || !band2 || !rgb_out) && n) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null array");
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (!n) return SARPRO_HIP_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->stage_in[0].reserve(n));
    HIPCHK(ctx, ctx->stage_in[1].reserve(n));
    HIPCHK(ctx, ctx->stage_out[0].reserve(n * 3));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[0].p, band1, n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[1].p, band2, n, hipMemcpyHostToDevice, ctx->stream));
    RETCHK(sarpro_hip_synrgb_u8_dev(ctx, mode, strategy, ctx->stage_in[0].as<uint8_t>(), ctx->stage_in[1].as<uint8_t>(), n,
                                    ctx->stage_out[0].as<uint8_t>()));
    HIPCHK(ctx, hipMemcpyAsync(rgb_out, ctx->stage_out[0].p, n * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// synthetic scene
// ---------------------------------------------------------------------------------------
extern "C" int sarpro_hip_synth_scene_u16_dev_ex(sarpro_hip_ctx *ctx, uint64_t seed, int band, const uint16_t *q_tables_host,
                                                 size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                                 uint16_t *d_out, size_t pitch, uint32_t flags) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (band < 0 || band > 1 || !q_tables_host || (!d_out && rows_local * cols) || pitch < cols)
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synthetic scene arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t qbytes = sizeof(uint16_t) * 4 * 65536;
    HIPCHK(ctx, ctx->qtab.reserve(qbytes));
    HIPCHK(ctx, hipMemcpyAsync(ctx->qtab.p, q_tables_host + (size_t)band * 4 * 65536, qbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, launch_synth_scene_u16(seed, band, ctx->qtab.as<uint16_t>(), rows_total, cols, row0, rows_local, d_out, pitch, flags, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}
extern "C" int sarpro_hip_synth_scene_u16_dev(sarpro_hip_ctx *ctx, uint64_t seed, int band, const uint16_t *q_tables_host,
                                              size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                              uint16_t *d_out, size_t pitch) {
    return sarpro_hip_synth_scene_u16_dev_ex(ctx, seed, band, q_tables_host, rows_total, cols, row0, rows_local, d_out, pitch, 0u);
}

// ---------------------------------------------------------------------------------------
// row-stripe protocol
// ---------------------------------------------------------------------------------------
struct sarpro_hip_stripe {
    U16Job job;
    int phase = 0;
};

extern "C" int sarpro_hip_stripe_begin_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2,
                                           size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch,
                                           int strategy, int mode, sarpro_hip_stripe **out) {
    if (!ctx || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    if ((!d_band1 || !d_band2) && rows_local * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    sarpro_hip_stripe *s = new sarpro_hip_stripe();
    U16Job &J = s->job;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = d_band1; J.d_in[1] = d_band2;
    J.rows_total = rows_total; J.cols = cols; J.row0 = row0; J.rows_local = rows_local; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true;
    timing_reset(ctx);
    int rc = job_init(J);
    if (rc) { delete s; return rc; }
    J.plan->refs += 1; // the plan cache never evicts a plan an open stripe handle works on
    *out = s;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase1(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 0) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(job_phase1(s->job));
    HIPCHK(s->job.ctx, hipStreamSynchronize(s->job.ctx->stream)); // the buffer is complete when we return
    *d_buf = s->job.ctx->ghist.as<uint64_t>();
    *count = 65536 * 2;
    s->phase = 1;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase2(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 1) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(job_after_phase1(s->job));
    RETCHK(job_phase2(s->job));
    HIPCHK(s->job.ctx, hipStreamSynchronize(s->job.ctx->stream));
    if (s->job.clahe()) { *d_buf = s->job.ctx->tile_bins.as<uint64_t>(); *count = 64 * 256 * 2; }
    else { *d_buf = nullptr; *count = 0; }
    s->phase = 2;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase3(sarpro_hip_stripe *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 2) return SARPRO_HIP_ERR_INVALID_ARG;
    void *outs[kMaxBands] = {nullptr, nullptr};
    RETCHK(job_phase3(s->job, outs, 0));
    HIPCHK(s->job.ctx, hipStreamSynchronize(s->job.ctx->stream));
    if (s->job.clahe()) { *d_buf = s->job.ctx->level_hist.as<uint64_t>(); *count = 256 * 2; }
    else { *d_buf = nullptr; *count = 0; } // percentile strategies: level histogram follows from the reduced DN histogram
    s->phase = 3;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_phase4(sarpro_hip_stripe *s, uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out) {
    if (!s || s->phase != 3) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!d_rgb && s->job.rows_local * s->job.cols) return fail(s->job.ctx, SARPRO_HIP_ERR_INVALID_ARG, "null rgb raster");
    if (rgb_pitch_px < s->job.cols) return fail(s->job.ctx, SARPRO_HIP_ERR_INVALID_ARG, "rgb_pitch_px < cols");
    void *outs[kMaxBands] = {nullptr, nullptr};
    if (s->job.rows_local * s->job.cols) RETCHK(job_phase4(s->job, outs, 0, d_rgb, rgb_pitch_px, false));
    if (stats_out) { stats_out[0] = s->job.stats[0]; stats_out[1] = s->job.stats[1]; }
    s->phase = 4;
    return SARPRO_HIP_OK;
}

extern "C" void sarpro_hip_stripe_end(sarpro_hip_stripe *s) {
    if (!s) return;
    if (s->job.plan && s->job.plan->refs > 0) s->job.plan->refs -= 1;
    delete s;
}

// One call per rank for one row stripe of a scene, reductions over the library's RCCL communicator
// (sarpro_hip_comm_init): the device-resident chains run unchanged with three (CLAHE) or one (percentile
// strategies) small all-reduces enqueued on the stream between their kernels -- no host synchronisation
// until the stripe's RGB is complete.
namespace sarpro {
// Which route a stripe takes (device-resident chain or host phases, fused pass or apply + compose) follows from the LAYOUT of its
// rasters -- alignment of the pointers, pitches -- and a route fixes the sequence of collectives the rank joins.  Ranks with
// different layouts would therefore join different sequences and wait for each other for ever.  The one-call stripe entry point
// makes the layout a property of the library, not of the caller: a stripe whose rasters are not in the aligned form (16-byte
// aligned pointers, in_pitch % 16 == 0, rgb_pitch_px % 16 == 0) is staged through library-owned rasters that are (one device copy
// in, one out: 11 B/px on that rank only), so every rank of a scene takes the same route whatever it was handed.
static int stripe_run_u16_impl(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total,
                               size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode,
                               uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out) {
    if (!ctx->comm && !ctx->local_group && !ctx->attrs.on(A_COMM_REPLAY)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "no communicator on this context (sarpro_hip_comm_init / _init_local)");
    comm_replay_rewind(ctx);
    if ((!d_band1 || !d_band2 || !d_rgb) && rows_local * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (rgb_pitch_px < cols || in_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    const bool empty = rows_local == 0 || cols == 0;
    const bool aligned = !empty && in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(d_band1) && ptr_aligned16(d_band2) && ptr_aligned16(d_rgb);
    const size_t lib_pitch = round_up(std::max<size_t>(cols, 1), 64);
    uint8_t *rgb_user = nullptr;
    size_t rgb_user_pitch = 0;
    if (empty) { // nothing of this rank is read or written; its pitches still decide the route: the library's
        in_pitch = rgb_pitch_px = lib_pitch;
    } else if (!aligned) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        const uint16_t *src[2] = {d_band1, d_band2};
        for (int b = 0; b < 2; ++b) {
            HIPCHK(ctx, ctx->stage_in[b].reserve(rows_local * lib_pitch * sizeof(uint16_t)));
            HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_in[b].p, lib_pitch * sizeof(uint16_t), src[b], in_pitch * sizeof(uint16_t), cols * sizeof(uint16_t), rows_local,
                                         hipMemcpyDeviceToDevice, ctx->stream));
        }
        HIPCHK(ctx, ctx->stage_out[0].reserve(rows_local * lib_pitch * 3));
        d_band1 = ctx->stage_in[0].as<uint16_t>(); d_band2 = ctx->stage_in[1].as<uint16_t>();
        rgb_user = d_rgb; rgb_user_pitch = rgb_pitch_px;
        d_rgb = ctx->stage_out[0].as<uint8_t>();
        in_pitch = rgb_pitch_px = lib_pitch;
    }
    U16Job J;
    J.ctx = ctx; J.nbands = 2; J.d_in[0] = d_band1; J.d_in[1] = d_band2;
    J.rows_total = rows_total; J.cols = cols; J.row0 = row0; J.rows_local = rows_local; J.in_pitch = in_pitch;
    J.strategy = strategy; J.bit_depth = SARPRO_BITDEPTH_U8; J.mode = mode; J.synrgb = true; J.reduce = true;
    void *outs[kMaxBands] = {nullptr, nullptr};
    RETCHK(job_run_all(J, outs, 0, d_rgb, rgb_pitch_px, stats_out));
    if (rgb_user) {
        HIPCHK(ctx, hipMemcpy2DAsync(rgb_user, rgb_user_pitch * 3, d_rgb, rgb_pitch_px * 3, cols * 3, rows_local, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SARPRO_HIP_OK;
}
} // namespace sarpro

extern "C" int sarpro_hip_stripe_run_u16(sarpro_hip_ctx *ctx, const uint16_t *d_band1, const uint16_t *d_band2, size_t rows_total,
                                         size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int mode,
                                         uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const int rc = sarpro::stripe_run_u16_impl(ctx, d_band1, d_band2, rows_total, cols, row0, rows_local, in_pitch, strategy, mode, d_rgb, rgb_pitch_px, stats_out);
    if (rc != SARPRO_HIP_OK) sarpro::comm_abort_local_group(ctx); // an in-process group: the peers of a rank that failed must not wait for it (comm.cpp)
    return rc;
}

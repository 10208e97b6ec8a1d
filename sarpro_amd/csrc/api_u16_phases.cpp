// api_u16_phases.cpp -- the u16 flavour as explicit phases: the host route (statistics and tables on the host between the passes) and the
// row-stripe protocol's steps (each phase ends in a small integer reduction).
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

namespace sarpro {

// ---------------------------------------------------------------------------------------
// The u16 pipeline as explicit phases (also the row-stripe protocol: each phase ends in a
// small integer reduction that a multi-rank driver all-reduces before the next phase).
// ---------------------------------------------------------------------------------------


int job_init(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (J.strategy < 0 || J.strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (J.bit_depth != SARPRO_BITDEPTH_U8 && J.bit_depth != SARPRO_BITDEPTH_U16)
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if (J.mode < 0 || J.mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (J.in_pitch < J.cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    if (J.rows_total > 0x7FFFFFFFull || J.cols > 0x7FFFFFFFull) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "raster too large");
    if (J.row0 + J.rows_local > J.rows_total) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "stripe outside the scene");
    if (J.clahe() && !clahe_shape_ok(J.rows_total, J.cols))
        return fail(ctx, SARPRO_HIP_ERR_UNSUPPORTED_SHAPE,
                    "CLAHE tile arithmetic underflows for this shape (reference panics: autoscale.rs:250,254)");
    J.vec = J.in_pitch % 8 == 0;
    for (int b = 0; b < J.nbands; ++b) J.vec = J.vec && ptr_aligned16(J.d_in[b]);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return get_plan(ctx, J.rows_total, J.cols, J.row0, J.rows_local, J.vec ? 8 : 1, &J.plan);
}

// the tile histograms were zeroed again by their last reader (enqueued): the next histogram pass may skip its fill
void mark_tile_hist_clean(U16Job &J) {
    if (!J.tile_hist_bytes) return; // the pass was not begun by this job object (cannot know its footprint)
    J.ctx->tile_hist_clean_ptr = J.ctx->tile_hist[0].p;
    J.ctx->tile_hist_clean_bytes = J.tile_hist_bytes;
}

// phase 1: local DN histograms -> ctx->ghist (u64 [nbands][65536]) on the device
uint32_t *tile_hist_of(sarpro_hip_ctx *ctx, int band, int ntiles) { return ctx->tile_hist[0].as<uint32_t>() + (size_t)band * 65536 * (size_t)ntiles; }

// The histogram pass can be issued in pieces (streaming ingest: the work items whose rows have arrived):
// `begin` clears the tile histograms, [first, last) are indices into the plan's item list (sorted by row),
// `end` folds the tile histograms into the band histogram.  The default is the whole pass.
int job_phase1(U16Job &J, bool begin, int first, int last, bool end) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (J.hist_done) return SARPRO_HIP_OK;
    const bool tiled = J.clahe();
    const int ntiles = tiled ? kTiles * kTiles : 1;
    HIPCHK(ctx, ctx->ghist.reserve(sizeof(uint64_t) * 65536 * kMaxBands));
    DnHistArgs a{};
    // both bands' tile histograms in one allocation (band b at tile_hist_of(ctx, b, ntiles)): one fill instead of two
    const size_t band_bytes = sizeof(uint32_t) * 65536 * (size_t)ntiles;
    HIPCHK(ctx, ctx->tile_hist[0].reserve(band_bytes * kMaxBands));
    if (begin && ctx->pipe_wait_before_hist) { // resident batch, PIPE_ORDER = 2: this scene's sweep beside the previous scene's fused pass, not before it
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_wait_before_hist, 0));
        ctx->pipe_wait_before_hist = nullptr;
    }
    if (begin) {
        // the chain's last reader of the tile histograms zeroes what it read: a scene that follows one of the same or a
        // larger footprint on this context starts on clean bins (the fill of 32 MiB and its launch: ~12 us)
        const size_t need = band_bytes * (size_t)J.nbands;
        if (!(ctx->tile_hist_clean_ptr == ctx->tile_hist[0].p && ctx->tile_hist_clean_bytes >= need))
            HIPCHK(ctx, hipMemsetAsync(ctx->tile_hist[0].p, 0, need, ctx->stream));
        ctx->tile_hist_clean_bytes = 0; // dirty from here on
        J.tile_hist_bytes = need;
    }
    for (int b = 0; b < J.nbands; ++b) {
        a.in[b] = J.d_in[b];
        a.tile_hist[b] = tile_hist_of(ctx, b, ntiles);
    }
    a.pitch = J.in_pitch;
    const int nall = (int)(tiled ? J.plan->hist_rects_tiled.size() : J.plan->hist_rects_flat.size());
    if (last < 0 || last > nall) last = nall;
    a.rects = (tiled ? J.plan->d_hist_rects_tiled : J.plan->d_hist_rects_flat).as<Rect>() + first;
    a.lds_bins = 8192;
    const int nrects = last - first;
    if (J.vec && tiled && J.nbands == 2 && J.plan->piece_grid > 0 && first == 0 && last == nall && nall > 0 && !ctx->attrs.on(A_NO_PIECE_HIST)) {
        // whole tiled pass, both bands: persistent workgroups on balanced pieces (piece_kernels.hip k_dn_hist_pieces)
        DnHistPiecesArgs pa{};
        for (int b = 0; b < 2; ++b) { pa.in[b] = a.in[b]; pa.tile_hist[b] = a.tile_hist[b]; }
        pa.pitch = a.pitch; pa.items = J.plan->d_piece_items.as<PieceItem>(); pa.wg_first = J.plan->d_piece_first.as<int32_t>();
        pa.lds_bins = kPieceLdsBins;
        {
            KernelTimer t(ctx, "dn_hist_u16");
            HIPCHK(ctx, launch_dn_hist_pieces(pa, J.plan->piece_grid, ctx->stream));
        }
        if (ctx->pipe_record_after_hist) { HIPCHK(ctx, hipEventRecord(ctx->pipe_record_after_hist, ctx->stream)); ctx->pipe_record_after_hist = nullptr; }
    } else if (J.vec && !tiled && first == 0 && last == nall && nall > 0 && !ctx->attrs.on(A_NO_LINEAR_HIST)) {
        KernelTimer t(ctx, "dn_hist_u16"); // whole untiled pass in one go: the in-order sweep
        HIPCHK(ctx, launch_dn_hist_u16_linear(a, (uint32_t)J.rows_local, (uint32_t)J.cols, J.nbands, ctx->stream));
    } else if (nrects > 0) {
        KernelTimer t(ctx, "dn_hist_u16");
        if (J.vec) HIPCHK(ctx, launch_dn_hist_u16_interior(a, nrects, J.nbands, ctx->stream));
        else HIPCHK(ctx, launch_dn_hist_u16(a, nrects, J.nbands, false, ctx->stream));
    }
    if (!end) return SARPRO_HIP_OK;
    if (J.vec && !(tiled ? J.plan->hist_sliver_tiled : J.plan->hist_sliver_flat).empty()) { // unused unless the planner splits slivers
        const int ns = (int)(tiled ? J.plan->hist_sliver_tiled.size() : J.plan->hist_sliver_flat.size());
        a.rects = (tiled ? J.plan->d_hist_sliver_tiled : J.plan->d_hist_sliver_flat).as<Rect>();
        a.lds_bins = 2048;
        KernelTimer t(ctx, "dn_hist_u16_sliver");
        HIPCHK(ctx, launch_dn_hist_u16(a, ns, J.nbands, false, ctx->stream));
    }
    {
        SumTileHistArgs sa{};
        for (int b = 0; b < J.nbands; ++b) {
            sa.tile_hist[b] = tile_hist_of(ctx, b, ntiles);
            sa.out[b] = ctx->ghist.as<unsigned long long>() + (size_t)b * 65536;
        }
        sa.clear = J.clear_after_sum ? 1u : 0u;
        KernelTimer t(ctx, "sum_tile_hists");
        HIPCHK(ctx, launch_sum_tile_hists(sa, ntiles, J.nbands, ctx->stream));
        if (sa.clear) mark_tile_hist_clean(J);
    }
    return SARPRO_HIP_OK;
}

// after the (optional) all-reduce of ghist: stats, window, DN tables
int job_after_phase1(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    const size_t bytes = sizeof(uint64_t) * 65536 * (size_t)J.nbands;
    HIPCHK(ctx, ctx->h_ghist.reserve(sizeof(uint64_t) * 65536 * kMaxBands));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_ghist.p, ctx->ghist.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, ctx->luts.reserve(2 * 131072));
    HIPCHK(ctx, ctx->h_upload.reserve(2 * 131072 + 2 * 64 * 256 * 8 + 66048 + 1024));
    // the bands are independent: band 1 is worked on by a helper thread while this thread does band 0
    auto band_work = [&J, ctx](int b) -> int {
        uint64_t *h = ctx->h_ghist.as<uint64_t>() + (size_t)b * 65536;
        { // the interior kernel does not count DN = 0: it is what is left of the scene
            uint64_t others = 0;
            for (uint32_t dn = 1; dn < 65536; ++dn) others += h[dn];
            h[0] = (uint64_t)J.rows_total * J.cols - others;
        }
        RETCHK(stats_from_dn_hist(h, &J.stats[b]));
        RETCHK(select_window(&J.stats[b], J.strategy, J.tamed_kind(b)));
        if (J.clahe()) build_clahe_bin_lut_u16(J.stats[b], &J.lut[b]);
        else build_level_lut_u16(J.stats[b], J.u8_out() ? SARPRO_BITDEPTH_U8 : SARPRO_BITDEPTH_U16, J.tamed_kind(b), &J.lut[b]);
        if (!J.clahe() && J.u8_out()) {
            // levels are a function of DN: their histogram, min and max follow from the DN histogram
            std::memset(J.level_hist_h[b], 0, sizeof(J.level_hist_h[b]));
            for (uint32_t dn = 0; dn < 65536; ++dn)
                if (h[dn]) J.level_hist_h[b][dn ? J.lut[b].full[dn] : 0] += h[dn];
        }
        return SARPRO_HIP_OK;
    };
    if (J.nbands == 2) {
        std::future<int> other = std::async(std::launch::async, band_work, 1);
        const int rc0 = band_work(0), rc1 = other.get();
        if (rc0) return rc0;
        if (rc1) return rc1;
    } else {
        RETCHK(band_work(0));
    }
    return SARPRO_HIP_OK;
}

// u8 rescale (autoscale.rs:348-364) from the level histogram; tamed_synrgb has none (:731-741)
void job_rescale_from_level_hist(U16Job &J, int b) {
    if (J.tamed_kind(b) != kNotTamedSynrgb) {
        for (int i = 0; i < 256; ++i) J.resc[b][i] = (uint8_t)i;
        J.resc_identity[b] = true;
        return;
    }
    unsigned mn = 0, mx = 0;
    bool any = false;
    for (unsigned i = 0; i < 256; ++i)
        if (J.level_hist_h[b][i]) { if (!any) mn = i; mx = i; any = true; }
    u8_rescale_lut(mn, mx, J.resc[b]);
    J.resc_identity[b] = true;
    for (unsigned i = 0; i < 256; ++i)
        if (J.level_hist_h[b][i] && J.resc[b][i] != i) J.resc_identity[b] = false;
}

// phase 2 (CLAHE): per-tile bin histograms -> ctx->tile_bins (u64 [nbands][64][256])
int job_phase2(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (!J.clahe()) return SARPRO_HIP_OK;
    HIPCHK(ctx, ctx->tile_bins.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands));
    TileBinHistArgs ta{};
    for (int b = 0; b < J.nbands; ++b) {
        uint8_t *stage = ctx->h_upload.as<uint8_t>() + (size_t)b * 65536;
        for (int i = 0; i < 65536; ++i) stage[i] = (uint8_t)J.lut[b].full[i];
        HIPCHK(ctx, hipMemcpyAsync(ctx->luts.as<uint8_t>() + (size_t)b * 131072, stage, 65536, hipMemcpyHostToDevice, ctx->stream));
        ta.tile_hist[b] = tile_hist_of(ctx, b, kTiles * kTiles);
        ta.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        ta.out[b] = ctx->tile_bins.as<unsigned long long>() + (size_t)b * 64 * 256;
    }
    KernelTimer t(ctx, "tile_bin_hist");
    HIPCHK(ctx, launch_tile_bin_hist(ta, kTiles * kTiles, J.nbands, ctx->stream));
    return SARPRO_HIP_OK;
}

int ensure_levels(U16Job &J) {
    sarpro_hip_ctx *ctx = J.ctx;
    J.lvl_pitch = round_up(J.cols, 64);
    for (int b = 0; b < J.nbands; ++b) {
        HIPCHK(ctx, ctx->levels[b].reserve(J.lvl_pitch * std::max<size_t>(J.rows_local, 1)));
        J.d_levels[b] = ctx->levels[b].as<uint8_t>();
    }
    return SARPRO_HIP_OK;
}

// phase 3: apply.  d_out[b] (+ out_pitch) receive the per-band raster when the caller wants it
// (single-band entry points: the final raster; dual-pol: optional u8 copies, may be null).
// Leaves the u8 level histogram in ctx->level_hist (CLAHE u8) for the reduction.
int job_phase3(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch) {
    sarpro_hip_ctx *ctx = J.ctx;
    const bool u8o = J.u8_out();
    HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands));
    if (J.clahe()) {
        // CDFs from the (reduced) tile histograms
        const size_t tb_bytes = sizeof(uint64_t) * 64 * 256 * (size_t)J.nbands;
        HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_small.p, ctx->tile_bins.p, tb_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, ctx->cdfs.reserve(sizeof(double) * 64 * 256 * kMaxBands));
        double *h_cdfs = reinterpret_cast<double *>(ctx->h_upload.as<uint8_t>() + 2 * 131072);
        for (int b = 0; b < J.nbands; ++b)
            RETCHK(clahe_cdfs(ctx->h_small.as<uint64_t>() + (size_t)b * 64 * 256, J.rows_total, J.cols, h_cdfs + (size_t)b * 64 * 256));
        HIPCHK(ctx, hipMemcpyAsync(ctx->cdfs.p, h_cdfs, sizeof(double) * 64 * 256 * (size_t)J.nbands, hipMemcpyHostToDevice, ctx->stream));

        ClaheApplyArgs a{};
        const bool direct = !J.synrgb && d_out[0] != nullptr; // single band: write the caller's raster
        if (!direct) RETCHK(ensure_levels(J));
        size_t win_max = 0;
        for (int b = 0; b < J.nbands; ++b) {
            a.in[b] = J.d_in[b];
            a.out[b] = direct ? d_out[b] : (void *)J.d_levels[b];
            a.cdfs[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
            a.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            a.win_lo[b] = J.lut[b].win_lo;
            a.win_hi[b] = J.lut[b].win_hi;
            win_max = std::max<size_t>(win_max, J.lut[b].win_hi - J.lut[b].win_lo + 1);
            a.level_hist[b] = u8o ? ctx->level_hist.as<unsigned long long>() + (size_t)b * 256 : nullptr;
        }
        a.in_pitch = J.in_pitch;
        a.out_pitch = direct ? out_pitch : J.lvl_pitch;
        a.rects = J.plan->d_apply_rects.as<Rect>();
        a.lut_in_lds = win_max <= kLutLdsMaxBytes;
        a.row_w = J.plan->d_row_w.as<RowWeight>();
        a.col_w = J.plan->d_col_w.as<RowWeight>();
        a.row_off = (int32_t)J.row0;
        a.max_val = u8o ? 255.0 : 65535.0;
        if (u8o) HIPCHK(ctx, hipMemsetAsync(ctx->level_hist.p, 0, sizeof(uint64_t) * 256 * kMaxBands, ctx->stream));
        const bool vec = J.vec && a.out_pitch % 8 == 0 && ptr_aligned16(a.out[0]) && (J.nbands < 2 || ptr_aligned16(a.out[1]));
        if (vec != J.vec) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "output raster must be 16-byte aligned with pitch % 8 == 0 when the input is");
        if (J.vec && u8o && clahe_apply_spec_ok(a, J.nbands) && !ctx->attrs.on(A_NO_SPEC)) {
            HIPCHK(ctx, ctx->spec_dump.reserve(kSpecDumpBytes));
            a.dump = ctx->spec_dump.as<uint8_t>();
            KernelTimer t(ctx, "clahe_apply_u8_spec");
            HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)J.plan->apply_rects.size(), J.nbands, ctx->stream));
        } else {
            KernelTimer t(ctx, "clahe_apply_u16");
            HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_rects.size(), J.nbands, J.vec, !u8o, ctx->stream));
        }
        if (J.vec && !J.plan->apply_sliver.empty()) { // < 8-column leftovers at cell edges: scalar exact kernel
            a.rects = J.plan->d_apply_sliver.as<Rect>();
            KernelTimer t(ctx, "clahe_apply_sliver");
            HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_sliver.size(), J.nbands, false, !u8o, ctx->stream));
        }
        return SARPRO_HIP_OK;
    }
    // percentile strategies: the level histogram is known on the host already; publish it on the
    // device too so the stripe protocol reduces the same buffer in both modes
    if (u8o) {
        uint64_t *stage = reinterpret_cast<uint64_t *>(ctx->h_upload.as<uint8_t>() + 2 * 131072);
        for (int b = 0; b < J.nbands; ++b) std::memcpy(stage + (size_t)b * 256, J.level_hist_h[b], sizeof(uint64_t) * 256);
        HIPCHK(ctx, hipMemcpyAsync(ctx->level_hist.p, stage, sizeof(uint64_t) * 256 * (size_t)J.nbands, hipMemcpyHostToDevice, ctx->stream));
    }
    return SARPRO_HIP_OK;
}

// phase 4: finish.  Single band: final raster into d_out[0].  Dual-pol: RGB into d_rgb and
// (optionally) the per-band u8 rasters into d_out[b].
int job_phase4(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                      bool level_hist_reduced_on_device) {
    sarpro_hip_ctx *ctx = J.ctx;
    const bool u8o = J.u8_out();
    const uint32_t rows = (uint32_t)J.rows_local, cols = (uint32_t)J.cols;

    if (u8o && (J.clahe() || level_hist_reduced_on_device)) {
        HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
        uint64_t *h = ctx->h_small.as<uint64_t>() + 64 * 256 * kMaxBands;
        HIPCHK(ctx, hipMemcpyAsync(h, ctx->level_hist.p, sizeof(uint64_t) * 256 * (size_t)J.nbands, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < J.nbands; ++b) {
            std::memcpy(J.level_hist_h[b], h + (size_t)b * 256, sizeof(uint64_t) * 256);
            if (J.clahe()) { // the speculative kernel does not count level 0: it is what is left of the scene
                uint64_t others = 0;
                for (int i = 1; i < 256; ++i) others += J.level_hist_h[b][i];
                J.level_hist_h[b][0] = (uint64_t)J.rows_total * J.cols - others;
            }
        }
    }
    if (u8o) for (int b = 0; b < J.nbands; ++b) job_rescale_from_level_hist(J, b);

    uint8_t *up = ctx->h_upload.as<uint8_t>();
    // dual-pol percentile strategies without per-band outputs: ONE fused pass DN,DN -> RGB (7 B/px), no
    // intermediate u8 rasters
    bool fused = false;
    if (!J.clahe() && J.synrgb && !d_out[0] && !d_out[1] && J.vec && J.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 &&
        ptr_aligned16(d_rgb) && !ctx->attrs.on(A_NO_FUSED)) {
        LutComposeArgs probe{};
        probe.win_hi[0] = J.lut[0].win_hi; probe.win_hi[1] = J.lut[1].win_hi;
        fused = lut_compose_fits(probe);
    }
    if (!J.clahe()) {
        // table apply: final = resc[level[DN]] (u8) or level[DN] (u16)
        const bool need_levels = !fused && J.synrgb && (d_out[0] == nullptr || d_out[1] == nullptr);
        if (need_levels) RETCHK(ensure_levels(J));
        for (int b = 0; b < J.nbands; ++b) {
            LutApplyArgs a{};
            a.in = J.d_in[b];
            a.in_pitch = J.in_pitch;
            a.rows = rows; a.cols = cols;
            if (d_out[b]) { a.out = d_out[b]; a.out_pitch = out_pitch; }
            else { a.out = J.d_levels[b]; a.out_pitch = J.lvl_pitch; }
            a.win_lo = J.lut[b].win_lo; a.win_hi = J.lut[b].win_hi;
            const size_t esz = u8o ? 1 : 2;
            a.lut_in_lds = (size_t)(a.win_hi - a.win_lo + 1) * esz <= kLutLdsMaxBytes;
            uint8_t *stage = up + (size_t)b * 131072;
            if (u8o) for (int i = 0; i < 65536; ++i) stage[i] = J.resc[b][J.lut[b].full[i] & 0xFF];
            else std::memcpy(stage, J.lut[b].full.data(), 131072);
            if (u8o) stage[0] = J.resc[b][0];
            void *d_lut = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            HIPCHK(ctx, hipMemcpyAsync(d_lut, stage, 65536 * esz, hipMemcpyHostToDevice, ctx->stream));
            a.lut = d_lut;
            if (fused) continue; // the tables are consumed by the fused pass below
            const bool vec = J.vec && a.out_pitch % 8 == 0 && ptr_aligned16(a.out);
            KernelTimer t(ctx, "lut_apply_u16");
            HIPCHK(ctx, launch_lut_apply_u16(a, vec, !u8o, ctx->stream));
        }
    }
    if (!J.synrgb) {
        if (J.clahe() && u8o && !J.resc_identity[0]) { // rare: CLAHE levels did not span 0..255
            HIPCHK(ctx, ctx->tables.reserve(66048));
            std::memcpy(up, J.resc[0], 256);
            HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, up, 256, hipMemcpyHostToDevice, ctx->stream));
            KernelTimer t(ctx, "remap_u8");
            HIPCHK(ctx, launch_remap_u8(reinterpret_cast<uint8_t *>(d_out[0]), out_pitch, rows, cols, ctx->tables.as<uint8_t>(), ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return SARPRO_HIP_OK;
    }

    // ---- dual-pol composition (synthetic_rgb.rs:182-197) ----
    // combined histogram of the FINAL u8 bands = level histograms pushed through each band's rescale
    uint64_t combined[256];
    std::memset(combined, 0, sizeof(combined));
    for (int b = 0; b < 2; ++b)
        for (int i = 0; i < 256; ++i) combined[J.resc[b][i]] += J.level_hist_h[b][i];
    std::vector<uint8_t> luts(66048), tables(66048);
    const bool suppressed = J.strategy == SARPRO_STRATEGY_TAMED || J.strategy == SARPRO_STRATEGY_CLAHE;
    if (suppressed) {
        J.floor_with_cushion = synrgb_floor_from_hist(combined, (uint64_t)J.rows_total * J.cols);
        synrgb_luts_suppressed(J.floor_with_cushion, luts.data());
    } else {
        J.floor_with_cushion = -1;
        synrgb_luts_default(luts.data());
    }
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    // CLAHE: the compose kernel reads LEVELS, so the rescale is folded into the tables.
    // Percentile strategies: the table-apply pass already wrote final u8 values.
    const uint8_t *r1 = J.clahe() ? J.resc[0] : ident, *r2 = J.clahe() ? J.resc[1] : ident;
    fold_compose_tables(luts.data(), J.floor_with_cushion, r1, r2, tables.data());
    HIPCHK(ctx, ctx->tables.reserve(66048 + 512));
    uint8_t *tstage = up + 2 * 131072 + 2 * 64 * 256 * 8;
    std::memcpy(tstage, tables.data(), 66048);
    HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, tstage, 66048, hipMemcpyHostToDevice, ctx->stream));

    if (fused) {
        LutComposeArgs f{};
        for (int b = 0; b < 2; ++b) {
            f.in[b] = J.d_in[b];
            f.lut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            f.win_hi[b] = J.lut[b].win_hi;
        }
        f.rgb = d_rgb; f.in_pitch = J.in_pitch; f.rgb_pitch_px = rgb_pitch_px; f.rows = rows; f.cols = cols;
        f.tables = ctx->tables.as<uint8_t>();
        {
            KernelTimer t(ctx, "lut_compose_u16");
            HIPCHK(ctx, launch_lut_compose_u16(f, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return SARPRO_HIP_OK;
    }
    ComposeArgs c{};
    if (J.clahe()) { c.b1 = J.d_levels[0]; c.b2 = J.d_levels[1]; c.in_pitch = J.lvl_pitch; }
    else if (d_out[0] && d_out[1]) { c.b1 = (const uint8_t *)d_out[0]; c.b2 = (const uint8_t *)d_out[1]; c.in_pitch = out_pitch; }
    else {
        // percentile strategies wrote into d_out[b] when given, else into the level rasters; compose
        // needs one pitch for both bands, so mixed destinations are not offered by the entry points
        c.b1 = J.d_levels[0]; c.b2 = J.d_levels[1]; c.in_pitch = J.lvl_pitch;
    }
    c.rgb = d_rgb; c.rgb_pitch_px = rgb_pitch_px; c.rows = rows; c.cols = cols;
    c.tables = ctx->tables.as<uint8_t>();
    const int cvec = (c.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(c.b1) && ptr_aligned16(c.b2) && ptr_aligned16(d_rgb)) ? 16 : 1;
    {
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, cvec, ctx->stream));
    }
    if (J.clahe()) { // optional per-band u8 rasters: levels pushed through the rescale
        for (int b = 0; b < 2; ++b) {
            if (!d_out[b]) continue;
            HIPCHK(ctx, hipMemcpy2DAsync(d_out[b], out_pitch, J.d_levels[b], J.lvl_pitch, cols, rows, hipMemcpyDeviceToDevice, ctx->stream));
            if (!J.resc_identity[b]) {
                uint8_t *m = tstage + 66048 + (size_t)b * 256;
                std::memcpy(m, J.resc[b], 256);
                HIPCHK(ctx, hipMemcpyAsync(ctx->tables.as<uint8_t>() + 66048 + (size_t)b * 256, m, 256, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, launch_remap_u8(reinterpret_cast<uint8_t *>(d_out[b]), out_pitch, rows, cols,
                                            ctx->tables.as<uint8_t>() + 66048 + (size_t)b * 256, ctx->stream));
            }
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}


} // namespace sarpro

// api_plan.cpp -- work planning of the u16 flavour: a stripe of a scene -> the work items of every pass (cached per shape in the context).
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
// planning: split a stripe into work items
// ---------------------------------------------------------------------------------------
namespace sarpro {

// Column strips of the vector kernels start on a multiple of this many pixels (read when a plan is built): with 64, the 1-KiB
// row segment a wave reads (64 lanes x 8 u16) is 128-byte aligned and covers 8 memory lines instead of 9, the 512 bytes of
// levels it writes cover 4.5 instead of 5.  Cells and tiles start at multiples of tile_w / 2 (1250 px on the headline scene),
// so with the vector width alone (8) nearly every segment straddled a line at both ends: +13 % of HBM traffic on the apply
// pass by the PMC counters (profiles/r2_traffic.json).  The leading lanes of a cell's first strip are masked instead.
constexpr size_t kRgbItemRows = 256, kSampleItemRows = 1024, kU16ItemRows = 1024;
constexpr size_t kRgbTailRows = 1024, kRgbTailItemRows = 96; // fused CLAHE -> RGB pass: the stripe's last rows in small items (see get_plan)
constexpr size_t kRgbItemRowsLarge = 512, kRgbTailRowsLarge = 2500, kRgbTailItemRowsLarge = 128;
static size_t strip_align(const StripePlan &P, int vecw) { // (STRIP_ALIGN: planner tuning, read when the plan is built)
    if (vecw != 8 && vecw != 4) return (size_t)vecw;
    return std::max<size_t>(vecw, P.strip_align_px / vecw * vecw);
}

static void push_strips(std::vector<Rect> &out, const StripePlan &P, size_t lo, size_t hi, size_t c0, size_t c1,
                        const int ids[4], size_t chunk_rows, int vecw, int flags = 0) {
    if (lo >= hi || c0 >= c1) return;
    const size_t strip = 64 * (size_t)vecw, align = strip_align(P, vecw);
    for (size_t cs = c0 / align * align; cs < c1; cs += strip) {
        Rect r{};
        r.c0 = (int32_t)std::max(c0, cs);
        r.c1 = (int32_t)std::min(c1, cs + strip);
        r.cstart = (int32_t)cs;
        for (int k = 0; k < 4; ++k) r.id[k] = ids[k];
        r.pad[0] = flags;
        for (size_t rr = lo; rr < hi; rr += chunk_rows) {
            r.r0 = (int32_t)(rr - P.row0);
            r.r1 = (int32_t)(std::min(rr + chunk_rows, hi) - P.row0);
            out.push_back(r);
        }
    }
}

// Global rows [gr0, gr1) x columns [c0, c1) clipped to the local stripe -> work items.  With
// `sliver` given (vecw == 8) the aligned interior goes to `out`, the edge leftovers to `sliver`.
static void add_rects(std::vector<Rect> &out, std::vector<Rect> *sliver, const StripePlan &P, size_t gr0, size_t gr1,
                      size_t c0, size_t c1, const int ids[4], size_t chunk_rows, int vecw, int flags = 0) {
    const size_t lo = std::max(gr0, P.row0), hi = std::min(gr1, P.row0 + P.rows_local);
    if (lo >= hi || c0 >= c1) return;
    if (!sliver) { push_strips(out, P, lo, hi, c0, c1, ids, chunk_rows, vecw, flags); return; }
    const size_t a = (c0 + vecw - 1) / vecw * vecw, b = c1 / vecw * vecw; // aligned interior [a, b)
    if (a < b) {
        push_strips(out, P, lo, hi, a, b, ids, chunk_rows, vecw);
        push_strips(*sliver, P, lo, hi, c0, a, ids, chunk_rows * 4, 1);
        push_strips(*sliver, P, lo, hi, b, c1, ids, chunk_rows * 4, 1);
    } else {
        push_strips(*sliver, P, lo, hi, c0, c1, ids, chunk_rows * 4, 1);
    }
}

static int upload_vec(sarpro_hip_ctx *ctx, DevBuf &d, const void *src, size_t bytes) {
    if (!bytes) return SARPRO_HIP_OK;
    HIPCHK(ctx, d.reserve(bytes));
    HIPCHK(ctx, hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // src may be a temporary
    return SARPRO_HIP_OK;
}

// Pieces of a whole scene for persistent workgroups (piece_kernels.hip): strips of 1, 2, 4, 8 or 16 wave columns (256 px each)
// inside one interpolation cell, cut into row ranges so that each of the `grid` workgroups gets the same cost (rows x wave
// columns, the ragged last column included).  Cell-major order: a workgroup's pieces are neighbours.
static void build_pieces(StripePlan *P, int grid) {
    const ClaheGeometry &g = P->geom;
    struct Strip { PieceItem it; double cost_per_row; };
    std::vector<Strip> strips;
    double total = 0.0;
    // cells = ranges of constant (t0, t1), cut again where the weight changes sign -- the first half tile extrapolates (d < 0,
    // autoscale.rs:308-313) and takes the wider speculation margin, the half tile after it has the same tiles but d >= 0 --
    // and at the tile boundaries (a cell is offset by half a tile: the sample pass's strata are per tile)
    auto cuts = [](const std::vector<size_t> &starts, const std::vector<RowWeight> &w, size_t tile) {
        std::vector<size_t> out;
        for (size_t i = 0; i + 1 < starts.size(); ++i) {
            out.push_back(starts[i]);
            for (size_t k = starts[i] + 1; k < starts[i + 1]; ++k)
                if ((w[k - 1].d < 0.0) != (w[k].d < 0.0) || k % tile == 0) out.push_back(k);
        }
        out.push_back(starts.empty() ? 0 : starts.back());
        return out;
    };
    const std::vector<size_t> rcut = cuts(g.row_cell_start, g.row_w, g.tile_h), ccut = cuts(g.col_cell_start, g.col_w, g.tile_w);
    for (size_t ri = 0; ri + 1 < rcut.size(); ++ri) {
        const size_t r0 = rcut[ri], r1 = rcut[ri + 1];
        if (r0 >= r1) continue;
        const RowWeight &rw = g.row_w[r0];
        for (size_t ci = 0; ci + 1 < ccut.size(); ++ci) {
            const size_t c0 = ccut[ci], c1 = ccut[ci + 1];
            if (c0 >= c1) continue;
            const RowWeight &cw = g.col_w[c0];
            const bool neg = rw.d < 0.0 || cw.d < 0.0; // constant sign inside the cut cell
            const size_t palign = P->piece_align; // (pieces: 64-px alignment measured 5 % SLOWER on the histogram pass, unlike the strips of the apply pass)
            const size_t cstart = c0 / palign * palign;
            constexpr size_t kCh = (size_t)kPieceChunk; // px per wave column
            size_t nch = (c1 - cstart + kCh - 1) / kCh, off = 0;
            while (nch > 0) {
                int lg = kPieceWavesLog2;
                while ((size_t(1) << lg) > nch) --lg;
                const size_t gw = size_t(1) << lg;
                Strip st{};
                st.it.r0 = (int32_t)r0; st.it.r1 = (int32_t)r1;
                st.it.cstart = (int32_t)(cstart + off * kCh);
                st.it.c0 = (int32_t)std::max(c0, cstart + off * kCh);
                st.it.c1 = (int32_t)std::min(c1, cstart + (off + gw) * kCh);
                st.it.gx_log2 = lg;
                st.it.flags = neg ? 1 : 0;
                st.it.id[0] = rw.t0 * kTiles + cw.t0; st.it.id[1] = rw.t0 * kTiles + cw.t1;
                st.it.id[2] = rw.t1 * kTiles + cw.t0; st.it.id[3] = rw.t1 * kTiles + cw.t1;
                st.it.tile = (int32_t)(std::min<size_t>(r0 / g.tile_h, kTiles - 1) * kTiles + std::min<size_t>(c0 / g.tile_w, kTiles - 1));
                st.cost_per_row = (double)gw;
                total += st.cost_per_row * (double)(r1 - r0);
                strips.push_back(st);
                nch -= gw; off += gw;
            }
        }
    }
    P->piece_grid = grid;
    P->piece_items.clear();
    P->piece_first.assign((size_t)grid + 1, 0);
    const double share = total / (double)grid;
    double acc = 0.0;
    int k = 0;
    for (const Strip &st : strips) {
        const int gy = kPieceWaves >> st.it.gx_log2;
        int r = st.it.r0;
        while (r < st.it.r1) {
            int take = st.it.r1 - r;
            if (k < grid - 1) {
                const double room = share * (double)(k + 1) - acc;
                int rows = (int)(room / st.cost_per_row);
                rows = std::max(gy, (rows + gy - 1) / gy * gy); // whole steps of the 16 waves
                take = std::min(take, rows);
            }
            PieceItem it = st.it;
            it.r0 = r; it.r1 = r + take;
            P->piece_items.push_back(it);
            P->piece_first[(size_t)k + 1] = (int32_t)P->piece_items.size();
            acc += st.cost_per_row * (double)take;
            r += take;
            if (k < grid - 1 && acc >= share * (double)(k + 1) - 1e-9) ++k;
        }
    }
    for (int i = 1; i <= grid; ++i) P->piece_first[(size_t)i] = std::max(P->piece_first[(size_t)i], P->piece_first[(size_t)i - 1]);
}

// The exact u16 kernel's work: the cell-major item list cut into `nwg` contiguous shares of equal rows (a share boundary falls
// inside an item: the item is cut there, at a multiple of the 16 waves' step).  A workgroup walks its share top to bottom, strip
// after strip: it rebuilds its tables only where the cell changes (three or four times per launch).
static void build_u16_shares(const StripePlan &P, int nwg, std::vector<Rect> *items, std::vector<int32_t> *first) {
    items->clear();
    first->assign((size_t)nwg + 1, 0);
    double total = 0.0;
    for (const Rect &r : P.u16_rects) total += (double)(r.r1 - r.r0);
    const double share = total / (double)nwg;
    double acc = 0.0;
    int k = 0;
    for (const Rect &src : P.u16_rects) {
        int r = src.r0;
        while (r < src.r1) {
            int take = src.r1 - r;
            if (k < nwg - 1) {
                const double room = share * (double)(k + 1) - acc;
                const int rows = std::max(16, ((int)std::ceil(room) + 15) / 16 * 16);
                take = std::min(take, rows);
            }
            Rect it = src;
            it.r0 = r; it.r1 = r + take;
            items->push_back(it);
            (*first)[(size_t)k + 1] = (int32_t)items->size();
            acc += (double)take;
            r += take;
            if (k < nwg - 1 && acc >= share * (double)(k + 1) - 1e-9) ++k;
        }
    }
    for (int i = 1; i <= nwg; ++i) (*first)[(size_t)i] = std::max((*first)[(size_t)i], (*first)[(size_t)i - 1]);
}

int get_plan(sarpro_hip_ctx *ctx, size_t rows_total, size_t cols, size_t row0, size_t rows_local, int vecw,
             StripePlan **out) {
    auto key = std::make_tuple(rows_total, cols, row0, rows_local, vecw);
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) { *out = it->second; return SARPRO_HIP_OK; }
    if (ctx->plans.size() > 16) { // bounded cache; plans held by an open stripe handle stay
        for (auto jt = ctx->plans.begin(); jt != ctx->plans.end();) {
            if (jt->second->refs > 0) { ++jt; continue; }
            jt->second->release_all();
            delete jt->second;
            jt = ctx->plans.erase(jt);
        }
    }
    StripePlan *P = new StripePlan();
    P->rows_total = rows_total; P->cols = cols; P->row0 = row0; P->rows_local = rows_local; P->vecw = vecw;
    if (ctx->attrs.is_set(A_STRIP_ALIGN)) P->strip_align_px = (size_t)std::max<long long>(1, ctx->attrs.val(A_STRIP_ALIGN, 64));
    P->piece_align = (size_t)kPieceVec;
    if (ctx->attrs.is_set(A_PIECE_ALIGN)) P->piece_align = (size_t)std::max<long long>(kPieceVec, ctx->attrs.val(A_PIECE_ALIGN, kPieceVec) / kPieceVec * kPieceVec);
    build_clahe_geometry(rows_total, cols, &P->geom);
    const size_t strips_across = (cols + 64 * vecw - 1) / (64 * vecw) + kTiles;
    const size_t target_items = 4096;
    size_t chunk_rows = std::min<size_t>(128, std::max<size_t>(16, (rows_local * strips_across + target_items - 1) / target_items)); // <= 128 rows: with line-aligned strips 112..160 rows measured 3-4 % faster than 234 and than 96, 64 rows 12 % slower (per-item table staging), 512 rows 10 % slower (the resident workgroups drift apart and lose the sweep's DRAM locality)
    const AttrSet &at = ctx->attrs; // planner tuning (experiments): read when a plan is built, the plan is cached per shape
    if (at.is_set(A_CHUNK_ROWS)) chunk_rows = (size_t)std::max<long long>(8, at.val(A_CHUNK_ROWS, 0));
    const ClaheGeometry &g = P->geom;
    const bool split = false; // edge lanes are masked inside the vector kernels; no separate sliver items
    for (size_t ty = 0; ty < (size_t)kTiles; ++ty) {
        const size_t r0 = std::min(ty * g.tile_h, rows_total), r1 = std::min((ty + 1) * g.tile_h, rows_total);
        for (size_t tx = 0; tx < (size_t)kTiles; ++tx) {
            const size_t c0 = std::min(tx * g.tile_w, cols), c1 = std::min((tx + 1) * g.tile_w, cols);
            const int ids[4] = {(int)(ty * kTiles + tx), 0, 0, 0};
            add_rects(P->hist_rects_tiled, split ? &P->hist_sliver_tiled : nullptr, *P, r0, r1, c0, c1, ids, chunk_rows, vecw);
        }
    }
    {
        const int ids[4] = {0, 0, 0, 0};
        add_rects(P->hist_rects_flat, split ? &P->hist_sliver_flat : nullptr, *P, 0, rows_total, 0, cols, ids, chunk_rows, vecw);
    }
    // interpolation cells = ranges of constant (t0, t1), cut again where the weight changes sign: the first half tile
    // extrapolates (d < 0, autoscale.rs:308-313) and takes the speculative kernel's wider margin, the half tile after it has the
    // same tiles but d >= 0 and takes the interior margin (uncut, a third of the scene ran with the wide one)
    auto sign_cuts = [](const std::vector<size_t> &starts, const std::vector<RowWeight> &w) {
        std::vector<size_t> out;
        for (size_t i = 0; i + 1 < starts.size(); ++i) {
            out.push_back(starts[i]);
            for (size_t k = starts[i] + 1; k < starts[i + 1]; ++k)
                if ((w[k - 1].d < 0.0) != (w[k].d < 0.0)) out.push_back(k);
        }
        out.push_back(starts.empty() ? 0 : starts.back());
        return out;
    };
    const std::vector<size_t> rcells = sign_cuts(g.row_cell_start, g.row_w), ccells = sign_cuts(g.col_cell_start, g.col_w);
    for (size_t ri = 0; ri + 1 < rcells.size(); ++ri) {
        const size_t r0 = rcells[ri], r1 = rcells[ri + 1];
        if (r0 >= r1) continue;
        const RowWeight &rw = g.row_w[r0];
        for (size_t ci = 0; ci + 1 < ccells.size(); ++ci) {
            const size_t c0 = ccells[ci], c1 = ccells[ci + 1];
            if (c0 >= c1) continue;
            const RowWeight &cw = g.col_w[c0];
            const int ids[4] = {rw.t0 * kTiles + cw.t0, rw.t0 * kTiles + cw.t1, rw.t1 * kTiles + cw.t0,
                                rw.t1 * kTiles + cw.t1};
            // bit 0: the cell holds negative blend weights (dy < 0 or dx < 0) -- the speculative apply kernel widens its f32
            // error margin there
            const bool neg = rw.d < 0.0 || cw.d < 0.0;
            const int cell_flags = (neg ? 1 : 0) | (rw.d < 0.0 ? 2 : 0) | (cw.d < 0.0 ? 4 : 0); // bit 1 / 2: which weight is negative (the margin depends on it)
            add_rects(P->apply_rects, split ? &P->apply_sliver : nullptr, *P, r0, r1, c0, c1, ids, chunk_rows, vecw, cell_flags);
            if (vecw == 8) { // the conflict-free exact kernel with u16 levels out (kernels.hip 4a)
                size_t urows = kU16ItemRows;
                if (at.is_set(A_U16_ITEM_ROWS)) urows = (size_t)std::min<long long>(1024, std::max<long long>(16, at.val(A_U16_ITEM_ROWS, 0)));
                P->u16_item_rows = urows;
                add_rects(P->u16_rects, nullptr, *P, r0, r1, c0, c1, ids, urows, vecw, cell_flags);
            }
            if (vecw == 8) { // the fused CLAHE -> RGB pass (whole scenes and row stripes): 16 waves walk an item, so items are taller
                // the pass hands its items out by a counter, in sweep order: the last rows of the stripe are cut into small items, so that
                // the workgroups that finish their last large item early find something left to do (a tail of at most one small item).
                // Large scenes (six 512-row items per workgroup or more) take the taller items: fewer prologues.  Scene A at 400 MP,
                // configurations interleaved on one box (tools/time_rgb_items.py, profiles/r5/rgb_items.txt): static stride 0.632 ms;
                // counter, 256-row items, no tail 0.607; + 1024 tail rows in 96-row items 0.601; 512 / 2500 / 128 0.597; 640 / 2500 / 128 0.614.
                const bool large = ((rows_local + 511) / 512) * ((cols + 511) / 512) >= (size_t)6 * (size_t)std::max(ctx->cu_count, 1);
                size_t frows = large ? kRgbItemRowsLarge : kRgbItemRows;
                if (at.is_set(A_RGB_ITEM_ROWS)) frows = (size_t)std::max<long long>(16, at.val(A_RGB_ITEM_ROWS, 0));
                size_t tail_rows = std::min<size_t>(large ? kRgbTailRowsLarge : kRgbTailRows, rows_local / 8), tail_item = large ? kRgbTailItemRowsLarge : kRgbTailItemRows;
                if (at.is_set(A_RGB_TAIL_ROWS)) tail_rows = (size_t)std::max<long long>(0, at.val(A_RGB_TAIL_ROWS, 0));
                if (at.is_set(A_RGB_TAIL_ITEM_ROWS)) tail_item = (size_t)std::max<long long>(16, at.val(A_RGB_TAIL_ITEM_ROWS, 0));
                const size_t stripe_end = row0 + rows_local, tail_start = stripe_end > tail_rows ? stripe_end - tail_rows : 0;
                if (r0 < tail_start) add_rects(P->rgb_rects, nullptr, *P, r0, std::min(r1, tail_start), c0, c1, ids, frows, vecw, cell_flags);
                if (r1 > tail_start) add_rects(P->rgb_rects, nullptr, *P, std::max(r0, tail_start), r1, c0, c1, ids, std::min(frows, tail_item), vecw, cell_flags);
                size_t srows = kSampleItemRows;
                if (at.is_set(A_SAMPLE_ITEM_ROWS)) srows = (size_t)std::max<long long>(16, at.val(A_SAMPLE_ITEM_ROWS, 0));
                add_rects(P->sample_rects, nullptr, *P, r0, r1, c0, c1, ids, srows, vecw, cell_flags);
            }
        }
    }
    // launch order = sweep order: consecutive work items cover adjacent column strips of the same row
    // chunk, so the workgroups resident at any moment read neighbouring 1-KiB segments of the same
    // image rows (DRAM page locality) instead of strips megabytes apart
    auto sweep_order = [](std::vector<Rect> &v) {
        std::stable_sort(v.begin(), v.end(), [](const Rect &x, const Rect &y) {
            return x.r0 != y.r0 ? x.r0 < y.r0 : x.cstart < y.cstart;
        });
    };
    if (!at.on(A_NO_SWEEP_ORDER)) {
        sweep_order(P->hist_rects_tiled);
        sweep_order(P->hist_rects_flat);
        sweep_order(P->apply_rects);
        sweep_order(P->rgb_rects);
        sweep_order(P->sample_rects);
    }
    if (vecw == 8 && row0 == 0 && rows_local == rows_total && ctx->cu_count > 0) {
        long long pg = ctx->cu_count; // PIECE_GRID: planner tuning (how many persistent workgroups share the histogram sweep)
        if (at.is_set(A_PIECE_GRID)) pg = std::max<long long>(1, at.val(A_PIECE_GRID, pg));
        build_pieces(P, (int)std::min<long long>(pg, kPieceMaxGrid));
    }
    int rc = upload_vec(ctx, P->d_hist_rects_tiled, P->hist_rects_tiled.data(), P->hist_rects_tiled.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_piece_items, P->piece_items.data(), P->piece_items.size() * sizeof(PieceItem));
    if (!rc) rc = upload_vec(ctx, P->d_piece_first, P->piece_first.data(), P->piece_first.size() * sizeof(int32_t));
    if (!rc) rc = upload_vec(ctx, P->d_hist_rects_flat, P->hist_rects_flat.data(), P->hist_rects_flat.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_apply_rects, P->apply_rects.data(), P->apply_rects.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_rgb_rects, P->rgb_rects.data(), P->rgb_rects.size() * sizeof(Rect));
    if (!rc && !P->rgb_rects.empty()) { // the fused pass's saturation tables (host_logic.h); the column table padded so that every lane's 8-byte load is in range
        std::vector<uint8_t> cc, rb;
        P->sat_ok = clahe_saturated_levels(g, &cc, &rb);
        if (P->sat_ok) {
            cc.resize(round_up(cols, 64) + 64, 0);
            rc = upload_vec(ctx, P->d_sat_col, cc.data(), cc.size());
            if (!rc) rc = upload_vec(ctx, P->d_sat_row, rb.data(), rb.size());
        }
    }
    if (!rc) rc = upload_vec(ctx, P->d_sample_rects, P->sample_rects.data(), P->sample_rects.size() * sizeof(Rect));
    for (int nb = 1; nb <= kMaxBands && !rc && !P->u16_rects.empty() && ctx->cu_count > 0; ++nb) { // the exact u16 kernel's shares, per band count of a launch
        build_u16_shares(*P, std::max(1, ctx->cu_count / nb), &P->u16_items[nb - 1], &P->u16_first[nb - 1]);
        P->u16_nwg[nb - 1] = (int)P->u16_first[nb - 1].size() - 1;
        rc = upload_vec(ctx, P->d_u16_items[nb - 1], P->u16_items[nb - 1].data(), P->u16_items[nb - 1].size() * sizeof(Rect));
        if (!rc) rc = upload_vec(ctx, P->d_u16_first[nb - 1], P->u16_first[nb - 1].data(), P->u16_first[nb - 1].size() * sizeof(int32_t));
    }
    if (!rc) rc = upload_vec(ctx, P->d_hist_sliver_tiled, P->hist_sliver_tiled.data(), P->hist_sliver_tiled.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_hist_sliver_flat, P->hist_sliver_flat.data(), P->hist_sliver_flat.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_apply_sliver, P->apply_sliver.data(), P->apply_sliver.size() * sizeof(Rect));
    if (!rc) rc = upload_vec(ctx, P->d_row_w, g.row_w.data(), g.row_w.size() * sizeof(RowWeight));
    if (!rc) rc = upload_vec(ctx, P->d_col_w, g.col_w.data(), g.col_w.size() * sizeof(RowWeight));
    if (rc) {
        P->release_all();
        delete P;
        return rc;
    }
    ctx->plans[key] = P;
    *out = P;
    return SARPRO_HIP_OK;
}

} // namespace sarpro

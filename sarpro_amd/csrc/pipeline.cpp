// pipeline.cpp -- the resident batch: K dual-pol scenes that are already in HBM through ONE context.
//
// Replaces the sequential batch loop of api/mod.rs:484-533 (one scene after the other: read, process, write) for callers that keep
// their rasters on the device; the product per scene is save.rs:317-367 at native resolution (sarpro_hip_dualpol_synrgb_u16_dev).
//
// Why lanes.  One scene's CLAHE chain is   H (DN histogram sweep, 4 B/px, HBM-bound)  ->  S (a dozen short dependent kernels:
// statistics, CLAHE bins, CDFs, sample, prediction; launch- and latency-bound, a few workgroups each)  ->  F (the fused CLAHE -> RGB
// pass, 7 B/px, bound by its LDS gathers).  On one stream the chip idles through S, through every launch gap and through the tails
// of H and F (persistent workgroups that finish at different times).  With L lanes -- internal contexts with their own stream,
// workspaces and plans -- scene i + 1's H and S are in the queue while scene i's F runs: its S kernels run beside F's tail and H,
// its H fills the compute units F's workgroups leave.  Both sweeps own whole compute units (F: 160 KiB of LDS and 128 VGPRs x 1024
// threads; H: 144 KiB), so they never share one: the lanes hide S, the launch gaps and the tails, not H behind F.  Measured on the
// nine-scene cycle at 400 MP (profiles/r5/pipe_sweep.txt): one stream 1.034 ms per scene, 2 lanes 0.962, 3 lanes 0.958-0.963, 4 lanes
// 1.006 (round 5's last build, another box: 0.980 / 0.964 / 0.988 for 2 / 3 / 4 lanes).  Leaving eight compute units free of F
// (RGB_GRID = 248) so that the other lanes' short kernels need not wait for F's tail measured 0.965-0.980 against 0.963-0.967: not kept.
// PIPE_ORDER = 1 chains the F passes by events (F of scene i + 1 waits for F of scene i): 0.969-0.974, no better than the free
// run (default 0).  RGB_GRID / PIECE_GRID (planner attributes) size the two sweeps' persistent grids: giving F and H disjoint sets
// of compute units (RGB_GRID + PIECE_GRID = 256) so that they run side by side was measured too and lost -- 1.12 ms at 192 + 64,
// 1.35 at 208 + 48, 1.54 at 224 + 32: H on a quarter of the chip cannot pull its 1.6 GB in the time F needs.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "chain_kernels.h"
#include "internal.h"

using namespace sarpro;

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
            return e__ == hipErrorOutOfMemory ? SARPRO_HIP_ERR_OOM : SARPRO_HIP_ERR_HIP;          \
        }                                                                                         \
    } while (0)

static int fail(sarpro_hip_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

constexpr int kPipeDefaultLanes = 3, kPipeMaxLanes = 8, kPipeDefaultOrder = 0;

static int ensure_lanes(sarpro_hip_ctx *ctx, int lanes) {
    while ((int)ctx->lanes.size() < lanes) {
        sarpro_hip_ctx *l = nullptr;
        const int rc = sarpro_hip_ctx_create(ctx->device, SARPRO_HIP_CTX_ASYNC_DEV | (ctx->timing ? SARPRO_HIP_CTX_TIMING : 0u), &l);
        if (rc != SARPRO_HIP_OK) { ctx->err = std::string("resident batch: lane context: ") + sarpro_hip_last_error(nullptr); return rc; }
        ctx->lanes.push_back(l);
    }
    for (sarpro_hip_ctx *l : ctx->lanes) { // the lanes follow their parent's switches (sarpro_hip_ctx_set_attr keeps them so between calls)
        l->attrs = ctx->attrs;
        l->time_only = ctx->time_only;
        l->pipe_wait_before_fused = nullptr;
        l->pipe_record_after_fused = nullptr;
        l->pipe_wait_before_hist = nullptr;
        l->pipe_record_before_fused = nullptr;
        l->pipe_record_after_hist = nullptr;
        l->pipe_defer = false;
        l->pipe_deferred = nullptr;
    }
    return SARPRO_HIP_OK;
}

static int route_of(const ChainSpecState &st) {
    if (!st.spec_ok) return SARPRO_HIP_ROUTE_UNPROVEN;
    if (st.pool_overflow) return SARPRO_HIP_ROUTE_POOL_OVERFLOW;
    return st.verdict != 0 ? SARPRO_HIP_ROUTE_REFUTED : st.retried ? SARPRO_HIP_ROUTE_RETRIED : SARPRO_HIP_ROUTE_ACCEPTED;
}

extern "C" int sarpro_hip_batch_dualpol_synrgb_u16_dev(sarpro_hip_ctx *ctx, sarpro_hip_resident_scene *scenes, size_t nscenes, size_t rows,
                                                       size_t cols, size_t in_pitch, int strategy, int mode, size_t rgb_pitch_px, int lanes,
                                                       int continue_on_error, sarpro_hip_batch_report *report) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!scenes && nscenes) || !report) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "resident batch: null scenes / report");
    if (lanes < 0 || lanes > kPipeMaxLanes) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "resident batch: lanes must be 0 (default) .. 8");
    std::memset(report, 0, sizeof(*report));
    for (size_t i = 0; i < nscenes; ++i) { scenes[i].status = SARPRO_HIP_OK; scenes[i].route = SARPRO_HIP_ROUTE_NONE; }
    if (!nscenes) return SARPRO_HIP_OK;
    if (lanes == 0) lanes = (int)std::min<long long>(kPipeMaxLanes, std::max<long long>(1, ctx->attrs.val(A_PIPE_LANES, kPipeDefaultLanes)));
    lanes = (int)std::min<size_t>((size_t)lanes, nscenes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure_lanes(ctx, lanes)) return rc;
    // PIPE_ORDER: 0 (default) = free run: every lane enqueues its chain, the hardware interleaves.
    // 3 = the two sweeps of all lanes in ONE order, H(i + 1), F(i), H(i + 2), F(i + 1), ... -- F(i) waits for the NEXT scene's histogram
    // sweep, H(i + 2) for F(i), so the fused pass has the chip to itself and its event bracket is its own time (an event has to be recorded
    // before a stream can wait for it: the chain of scene i stops in front of its fused pass -- pipe_defer -- and its rest is enqueued after
    // scene i + 1's sweep).  Round 6, nine-scene cycle (profiles/r6/pipe_sweep*.txt, trace_ord3.txt): 0.936-0.948 ms per scene against
    // 0.913-0.954 for the free run -- the sweeps never overlap, but two event hand-overs per scene (15-19 us each) stand in the critical
    // path and the short kernels S(i + 1) slow the sweep they run beside (H 330-347 us instead of 297); with bench.py's event pair around
    // the fused pass 0.991 against 0.961.  CU-time is the conserved quantity: H + S + F = 297 + ~45 + 575 us of a full chip per scene,
    // and the free run already packs it to within 1-3 %.  (One stream for both sweeps of all lanes, the lanes' own streams joining before
    // and after, measured 0.97: the statistics chain starved behind the fused pass.)
    // 1 = the fused passes chained, 2 = H(i + 1) paired with F(i) (round 5's co-residency experiment).
    const long long order = ctx->attrs.val(A_PIPE_ORDER, kPipeDefaultOrder);
    const bool serial = order == 3 && lanes > 1;
    const bool chain_f = (order == 1 || order == 2) && lanes > 1;
    const bool pair_fh = order == 2 && lanes > 1; // scene i + 1's histogram pass starts when scene i's fused pass starts (and the fused passes follow each other)
    while ((chain_f || serial) && ctx->pipe_events.size() < ((pair_fh || serial) ? 2 : 1) * nscenes) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_events.push_back(e);
    }
    // (pinned memory is allocated in large steps: a reallocation costs about a millisecond, a batch of twenty scenes takes twenty)
    HIPCHK(ctx, ctx->pipe_routes.reserve(sizeof(ChainSpecState) * ((nscenes + 255) / 256 * 256)));
    ChainSpecState *routes = ctx->pipe_routes.as<ChainSpecState>();
    std::vector<char> has_route(nscenes, 0);
    timing_reset(ctx); // (this context's own event pairs belong to an earlier call)
    ctx->lane_times.clear();

    int first_error = SARPRO_HIP_OK;
    auto scene_failed = [&](size_t i, int rc, sarpro_hip_ctx *l) {
        if (scenes[i].status != SARPRO_HIP_OK) return; // (counted already)
        scenes[i].status = rc;
        ctx->err = std::string("resident batch: scene ") + std::to_string(i) + ": " + l->err;
        ++report->errors;
        if (first_error == SARPRO_HIP_OK) first_error = rc;
    };
    auto copy_route = [&](size_t i, sarpro_hip_ctx *l) {
        // the scene's verdict, copied out in stream order (the next scene of this lane overwrites the state)
        if (scenes[i].status == SARPRO_HIP_OK && strategy == SARPRO_STRATEGY_CLAHE && l->spec_state.p && l->spec_ran &&
            hipMemcpyAsync(&routes[i], l->spec_state.p, sizeof(ChainSpecState), hipMemcpyDeviceToHost, l->stream) == hipSuccess) has_route[i] = 1;
    };
    // serial order: the rest of scene i's chain (its fused pass waits for scene i + 1's sweep when there is one)
    auto finish_scene = [&](size_t i, bool has_next) {
        sarpro_hip_ctx *l = ctx->lanes[i % (size_t)lanes];
        hipEvent_t e_f = ctx->pipe_events[i];
        if (l->pipe_deferred) {
            l->pipe_wait_before_fused = has_next ? ctx->pipe_events[nscenes + i + 1] : nullptr;
            l->pipe_record_after_fused = e_f;
            const int rc = l->pipe_deferred();
            l->pipe_deferred = nullptr;
            l->pipe_wait_before_fused = nullptr;
            if (rc != SARPRO_HIP_OK) scene_failed(i, rc, l);
            if (l->pipe_record_after_fused) { (void)hipEventRecord(e_f, l->stream); l->pipe_record_after_fused = nullptr; }
        } else {
            (void)hipEventRecord(e_f, l->stream); // a scene whose chain never reached the fused pass (another route, an error) still releases its successors
        }
        copy_route(i, l);
    };
    size_t enq = 0;
    for (; enq < nscenes; ++enq) {
        sarpro_hip_resident_scene &sc = scenes[enq];
        sarpro_hip_ctx *l = ctx->lanes[enq % (size_t)lanes];
        if (chain_f) {
            l->pipe_wait_before_fused = enq ? ctx->pipe_events[enq - 1] : nullptr;
            l->pipe_record_after_fused = ctx->pipe_events[enq];
        }
        if (pair_fh) {
            l->pipe_wait_before_hist = enq >= 2 ? ctx->pipe_events[nscenes + enq - 1] : nullptr; // (scene 1 starts at once: it has scene 0's whole chain to hide behind)
            l->pipe_record_before_fused = ctx->pipe_events[nscenes + enq];
        }
        if (serial) {
            l->pipe_defer = true;
            l->pipe_deferred = nullptr;
            l->pipe_wait_before_hist = enq >= 2 ? ctx->pipe_events[enq - 2] : nullptr; // H(i) behind F(i - 2)
            l->pipe_record_after_hist = ctx->pipe_events[nscenes + enq];
        }
        const int rc = sarpro_hip_dualpol_synrgb_u16_dev(l, sc.d_band1, sc.d_band2, rows, cols, in_pitch, strategy, mode, sc.d_rgb, rgb_pitch_px,
                                                         nullptr, nullptr, 0, nullptr);
        if (chain_f) {
            // a scene whose chain never reached the fused pass (another route, an error) still releases its successor
            if (l->pipe_record_after_fused) (void)hipEventRecord(l->pipe_record_after_fused, l->stream);
            if (l->pipe_record_before_fused) (void)hipEventRecord(l->pipe_record_before_fused, l->stream);
            l->pipe_wait_before_fused = nullptr;
            l->pipe_record_after_fused = nullptr;
            l->pipe_wait_before_hist = nullptr;
            l->pipe_record_before_fused = nullptr;
        }
        if (serial) {
            if (l->pipe_record_after_hist) (void)hipEventRecord(l->pipe_record_after_hist, l->stream); // (a chain without the piece sweep: recorded where it stands)
            l->pipe_record_after_hist = nullptr;
            l->pipe_wait_before_hist = nullptr;
            l->pipe_defer = false;
        }
        if (rc != SARPRO_HIP_OK) scene_failed(enq, rc, l);
        if (serial) {
            if (enq) finish_scene(enq - 1, true);
        } else {
            copy_route(enq, l);
        }
        if (report->errors && !continue_on_error) { ++enq; break; } // api/mod.rs:518-526
    }
    if (serial && enq) finish_scene(enq - 1, false);
    int sync_rc = SARPRO_HIP_OK;
    for (int k = 0; k < lanes; ++k) {
        sarpro_hip_ctx *l = ctx->lanes[(size_t)k];
        const hipError_t e = hipStreamSynchronize(l->stream);
        if (e != hipSuccess && sync_rc == SARPRO_HIP_OK) {
            ctx->err = std::string("resident batch: lane synchronisation: ") + hipGetErrorString(e);
            sync_rc = SARPRO_HIP_ERR_HIP;
        }
    }
    if (sync_rc != SARPRO_HIP_OK) { // the device failed under the batch: no raster of it can be trusted
        for (size_t i = 0; i < enq; ++i) if (scenes[i].status == SARPRO_HIP_OK) scenes[i].status = sync_rc;
        report->errors = enq; report->processed = 0; report->skipped = nscenes - enq;
        return sync_rc;
    }
    for (size_t i = 0; i < enq; ++i) {
        if (scenes[i].status == SARPRO_HIP_OK) ++report->processed;
        if (has_route[i]) scenes[i].route = route_of(routes[i]);
    }
    report->skipped = nscenes - report->processed - report->errors;
    if (ctx->timing) { // the lanes' event pairs, lane after lane (sarpro_hip_last_kernel_times of THIS context reports them)
        std::vector<const char *> names(8192);
        std::vector<float> ms(8192);
        for (int k = 0; k < lanes; ++k) {
            const int n = sarpro_hip_last_kernel_times(ctx->lanes[(size_t)k], names.data(), ms.data(), (int)names.size());
            for (int j = 0; j < n; ++j)
                if (std::strncmp(names[(size_t)j], "host:", 5) != 0) ctx->lane_times.emplace_back(names[(size_t)j], ms[(size_t)j]);
        }
    }
    if (report->errors && !continue_on_error) return first_error;
    return SARPRO_HIP_OK;
}


// The resident batch for f32 bands (round 6): the reference's DEFAULT flow hands the raster core bands of a few megapixels, resampled
// on read (api/mod.rs:374-449: ~4 MP), and loops over the scenes of a directory (api/mod.rs:474-536).  At that size the f32 chain is
// bound by its host turns, not by its kernels: thresholds are found on the host between its sweeps (bisection against glibc, section 3
// of DESIGN.md), three or four stream synchronisations per band, ~0.25 ms per 2048 x 2048 dual-pol scene of which the GPU works ~0.1.
// With L lanes -- the parent's internal contexts (stream, workspaces, plans: kept between calls) -- each driven by a host thread of its
// own for the duration of the call, the host turn of one scene runs while the other lanes' kernels do; scenes are dealt dynamically.
// Every scene is sarpro_hip_dualpol_synrgb_resized_f32_dev on its lane: the rasters are that call's, bit for bit.
extern "C" int sarpro_hip_batch_dualpol_synrgb_resized_f32_dev(sarpro_hip_ctx *ctx, sarpro_hip_resident_scene_f32 *scenes, size_t nscenes, size_t rows, size_t cols,
                                                               size_t in_pitch, int strategy, int mode, unsigned flags, size_t target_size, int pad, int lanes,
                                                               int continue_on_error, sarpro_hip_batch_report *report) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if ((!scenes && nscenes) || !report) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "resident f32 batch: null scenes / report");
    if (lanes < 0 || lanes > kPipeMaxLanes) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "resident f32 batch: lanes must be 0 (default) .. 8");
    std::memset(report, 0, sizeof(*report));
    for (size_t i = 0; i < nscenes; ++i) scenes[i].status = SARPRO_HIP_OK;
    if (!nscenes) return SARPRO_HIP_OK;
    if (lanes == 0) lanes = 8; // (64 scenes of 2048 x 2048, CLAHE: one call per scene 2490 scenes / s, 1 lane 2250, 2 lanes 3630, 4 lanes 4530, 8 lanes 5110: profiles/r6/batch_rate_f32.txt)
    lanes = (int)std::min<size_t>((size_t)lanes, nscenes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure_lanes(ctx, lanes)) return rc;
    std::atomic<size_t> next{0}, processed{0}, errors{0};
    std::atomic<bool> stop{false};
    std::atomic<int> first_error{SARPRO_HIP_OK};
    std::vector<std::string> lane_err((size_t)lanes);
    auto worker = [&](int k) {
        sarpro_hip_ctx *l = ctx->lanes[(size_t)k];
        for (;;) {
            if (stop.load()) break;
            const size_t i = next.fetch_add(1);
            if (i >= nscenes) break;
            sarpro_hip_resident_scene_f32 &sc = scenes[i];
            const int rc = sarpro_hip_dualpol_synrgb_resized_f32_dev(l, sc.d_band1, sc.d_band2, rows, cols, in_pitch, strategy, mode, flags, target_size, pad, sc.d_rgb, nullptr);
            sc.status = rc;
            if (rc == SARPRO_HIP_OK) {
                processed.fetch_add(1);
            } else {
                errors.fetch_add(1);
                int expected = SARPRO_HIP_OK;
                if (first_error.compare_exchange_strong(expected, rc)) lane_err[(size_t)k] = std::string("resident f32 batch: scene ") + std::to_string(i) + ": " + l->err;
                if (!continue_on_error) stop.store(true); // api/mod.rs:518-526
            }
        }
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < lanes; ++k) pool.emplace_back(worker, k);
    worker(0); // (the calling thread drives lane 0)
    for (auto &t : pool) t.join();
    report->processed = processed.load();
    report->errors = errors.load();
    report->skipped = nscenes - report->processed - report->errors;
    for (const std::string &e : lane_err) if (!e.empty()) ctx->err = e;
    if (report->errors && !continue_on_error) return first_error.load();
    return SARPRO_HIP_OK;
}

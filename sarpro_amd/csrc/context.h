// context.h -- sarpro_hip_ctx: device, streams, grow-only workspace, cached plans.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "kernels.h"
#include "piece_kernels.h"

namespace sarpro {

struct DevBuf { // grow-only device allocation, freed with its owner
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PinnedBuf { // grow-only pinned host allocation, freed with its owner
    void *p = nullptr;
    size_t cap = 0;
    unsigned flags = hipHostMallocDefault;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes, flags);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Work decomposition for one (scene shape, stripe) -- built once, cached on the context.
struct StripePlan {
    size_t rows_total = 0, cols = 0, row0 = 0, rows_local = 0;
    int vecw = 1;                       // pixels per lane of the kernels this plan feeds
    ClaheGeometry geom;                 // of the WHOLE scene (rows_total x cols)
    // For vecw == 8 the items are split: INTERIOR items have c0, c1 multiples of 8 (every lane owns
    // 8 valid pixels: the branch-free kernels), SLIVER items hold the < 8 leftover columns at each
    // tile / cell edge and run the scalar kernels.  Other widths keep everything in the first list.
    std::vector<Rect> hist_rects_tiled, hist_sliver_tiled; // per-tile DN histogram items
    std::vector<Rect> hist_rects_flat, hist_sliver_flat;   // single-histogram items (non-CLAHE)
    std::vector<Rect> apply_rects, apply_sliver;           // interpolation-cell items
    size_t strip_align_px = 64, piece_align = 4;           // planner tuning (attributes STRIP_ALIGN, PIECE_ALIGN), fixed when the plan is built
    std::vector<Rect> rgb_rects, sample_rects;             // whole scene, vecw == 8: taller cell items of the fused CLAHE -> RGB pass (256 rows) and of its sample-only pre-pass (1024 rows)
    DevBuf d_rgb_rects, d_sample_rects;
    // vecw == 8: the conflict-free exact u16 kernel (kernels.hip 4a).  u16_rects = items of <= u16_item_rows rows in cell-major order
    // (cell, 512-column strip, rows top to bottom: vertical neighbours share their tables); u16_items[n - 1] = the same list cut
    // into u16_nwg[n - 1] contiguous shares of equal rows, one per workgroup of an n-band launch (u16_first: share k = items
    // [first[k], first[k + 1]))
    std::vector<Rect> u16_rects, u16_items[kMaxBands];
    std::vector<int32_t> u16_first[kMaxBands];
    int u16_nwg[kMaxBands] = {0, 0};
    DevBuf d_u16_items[kMaxBands], d_u16_first[kMaxBands];
    size_t u16_item_rows = 0;
    DevBuf d_sat_col, d_sat_row;                           // fused pass: saturation classes of the columns / level bits of the rows (ClaheRgbArgs)
    bool sat_ok = false;
    DevBuf d_hist_rects_tiled, d_hist_rects_flat, d_apply_rects, d_row_w, d_col_w;
    DevBuf d_hist_sliver_tiled, d_hist_sliver_flat, d_apply_sliver;
    // whole scene, vecw == 8: every persistent workgroup's pieces, balanced by cost (piece_kernels.hip)
    std::vector<PieceItem> piece_items;
    std::vector<int32_t> piece_first; // [piece_grid + 1]
    int piece_grid = 0;
    DevBuf d_piece_items, d_piece_first;
    int refs = 0; // open stripe handles that hold this plan (the cache never evicts those)
    void release_all() {
        d_piece_items.release(); d_piece_first.release(); d_rgb_rects.release(); d_sample_rects.release(); for (int b = 0; b < kMaxBands; ++b) { d_u16_items[b].release(); d_u16_first[b].release(); } d_sat_col.release(); d_sat_row.release();
        d_hist_rects_tiled.release(); d_hist_rects_flat.release(); d_apply_rects.release();
        d_hist_sliver_tiled.release(); d_hist_sliver_flat.release(); d_apply_sliver.release();
        d_row_w.release(); d_col_w.release();
    }
};

struct KernelTime { const char *name; hipEvent_t start, stop; };

} // namespace sarpro

namespace sarpro {
// One persistent helper thread of a context: runs one job at a time (the OTHER band of a dual-pol f32 product, on the context's
// twin -- its own stream and workspaces -- while the caller's thread runs the first band).
struct BandWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    std::atomic<bool> pending{false}, finished{false};
    bool quit = false;
    int rc = 0;
    void start() { th = std::thread([this] { run(); }); }
    // Calls come back to back when scenes stream: both sides spin for a short while before they sleep on the condition variable
    // (a futex wake-up costs the second band 20-60 us of a 230-us call).
    static bool spin_until(const std::atomic<bool> &flag, int spins) {
        for (int i = 0; i < spins; ++i) {
            if (flag.load(std::memory_order_acquire)) return true;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
        return flag.load(std::memory_order_acquire);
    }
    void run() {
        for (;;) {
            if (!spin_until(pending, 4000)) { // (~0.1 ms: scenes that stream arrive within it; an idle helper then sleeps on the condition variable)
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return pending.load(std::memory_order_acquire) || quit; });
                if (quit) return;
            }
            const int r = job();
            {
                std::lock_guard<std::mutex> lk(m);
                rc = r;
                pending.store(false, std::memory_order_release);
                finished.store(true, std::memory_order_release);
            }
            cv.notify_all();
        }
    }
    void submit(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            finished.store(false, std::memory_order_release);
            pending.store(true, std::memory_order_release);
        }
        cv.notify_all();
    }
    int wait() {
        if (!spin_until(finished, 20000)) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return finished.load(std::memory_order_acquire); });
        }
        std::lock_guard<std::mutex> lk(m); // (the worker publishes rc under the lock)
        return rc;
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
} // namespace sarpro

// ---- context attributes: the route switches (cross-check twins of every fast route, test hooks, planner tuning).  A switch is
// a (set, value) pair on the CONTEXT: read from the environment (SARPRO_HIP_<NAME>) once, when the context is created, and from then
// on only through sarpro_hip_ctx_set_attr / _reset_attr -- no getenv on any call path (a library that re-reads the process
// environment per call races with setenv in other threads and cannot be driven from the Rust surface).
#define SARPRO_ATTR_LIST(X) \
    X(NO_CHAIN) X(NO_FUSED) X(NO_SPEC) X(NO_PIECE_HIST) X(NO_LINEAR_HIST) X(NO_SWEEP_ORDER) X(FULL_LEVEL_HIST) X(NO_SAMPLED_HIST) \
    X(SAMPLED_HIST_MIN_PX) X(SAMPLE_STRIDE) X(SEPARATE_CDFS) X(NO_FUSED_RGB) X(SPEC_FORCE) X(FORCE_UNCERTAIN) \
    X(STRIP_ALIGN) X(PIECE_ALIGN) X(CHUNK_ROWS) X(RGB_ITEM_ROWS) X(SAMPLE_ITEM_ROWS) \
    X(NO_MAILBOX) X(NO_STEP_ESTIMATE) X(F32_ZONES) X(F32_ZONES_DEBUG) X(F32_DIRECT) X(F32_DIRECT_QCAP) X(F32_LEVEL_GENERAL) \
    X(F32_LEVEL_TABLE) X(F32_LEVEL_QCAP) X(F32_HOST_CDFS) X(F32_NO_VEC8) X(NO_BAND_TWIN) X(RESIZE_GENERIC) X(NO_RESIZE_LUT) \
    X(NO_U16_CF) X(U16_ITEM_ROWS) X(PIPE_LANES) X(PIPE_ORDER) X(RGB_GRID) X(PIECE_GRID) X(COMM_RECORD) X(COMM_REPLAY) X(NO_SPEC_RESCALE) X(RGB_TAIL_ROWS) X(RGB_TAIL_ITEM_ROWS)
namespace sarpro {
enum Attr : int {
#define X(n) A_##n,
    SARPRO_ATTR_LIST(X)
#undef X
    A_COUNT
};
struct AttrSet {
    long long v[A_COUNT] = {};
    bool set[A_COUNT] = {};
    bool on(Attr a) const { return set[a] && v[a] != 0; }                     // a switch: set and non-zero
    long long val(Attr a, long long dflt) const { return set[a] ? v[a] : dflt; } // a value: the default when unset
    bool is_set(Attr a) const { return set[a]; }
};
const char *attr_name(int a);                         // "NO_CHAIN", ... (without the SARPRO_HIP_ prefix); nullptr past the end
int attr_index(const char *name);                     // with or without the prefix; -1: unknown
void attrs_from_environment(AttrSet *out);            // once per context, at creation
} // namespace sarpro

struct sarpro_hip_ctx {
    int device = 0;
    unsigned flags = 0;
    sarpro::AttrSet attrs;
    hipStream_t stream = nullptr;
    std::string err;

    // workspace (device)
    sarpro::DevBuf tile_hist[sarpro::kMaxBands]; // u32 [ntiles][65536]
    size_t tile_hist_clean_bytes = 0;            // leading bytes of tile_hist[0] that the last chain left zeroed (stream order)
    const void *tile_hist_clean_ptr = nullptr;   // ... of this allocation
    sarpro::DevBuf ghist;                        // u64 [2][65536]
    sarpro::DevBuf tile_bins;                    // u64 [2][64][256]
    sarpro::DevBuf cdfs;                         // f64 [2][64][256]
    sarpro::DevBuf luts;                         // 2 x 128 KiB: per-band DN table (u8 or u16 entries)
    sarpro::DevBuf level_hist;                   // u64 [2][256]
    sarpro::DevBuf spec_dump;                    // kSpecDumpBytes: write-only scratch of the speculative apply kernel
    sarpro::DevBuf hist_flags;                   // u32 [2]: band whose partial level histogram had to be recounted
    sarpro::DevBuf tables;                       // compose tables 66048 B
    sarpro::DevBuf spec_state;                   // ChainSpecState of the speculative CLAHE chain (chain_kernels.h)
    int cu_count = 0;                            // compute units of the device (grid of the persistent piece kernels)
    sarpro::DevBuf levels[sarpro::kMaxBands];    // u8 level rasters (intermediate)
    sarpro::DevBuf stage_in[sarpro::kMaxBands];  // host API staging
    sarpro::DevBuf stage_out[3];
    sarpro::DevBuf qtab;                         // synthetic scene tables
    sarpro::DevBuf f32ws;                        // f32-path workspace
    sarpro::DevBuf f32zone;                      // f32 zone route: samples kept by the min / max pass (a few per cent of the scene)
    uint32_t resize_key[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}}; // (in, out, elem, precision, window, block span) of the cached coefficient tables
    sarpro::DevBuf resize_tmp, resize_coef[2], resized[2]; // resize path: intermediate image, coefficient tables, resized bands
    sarpro::DevBuf resize_halo, resize_geom;               // striped resize: the boundary zones of the intermediate rasters, the ranks' stripe geometry
    sarpro::DevBuf chain_consts;                 // device-resident chain: dB table | suppressed lut_r/g per floor | blue pairs
    sarpro::DevBuf chain_scratch;               // statistics step: per-slice partials | 4096 bins per band
    sarpro::DevBuf chain_state;                  // ChainBandState[2] | resc[2][256] | identity[2] | floor
    bool chain_ready = false;
    bool blue_factors_ok = false;                // chain_consts holds the verified blue factor tables (host_logic: synrgb_blue_factors_supp)
    uint32_t chain_levels_cap = 4096;            // LDS bytes per band of the fused pass's DN tables (percentile chain)
    uint32_t chain_lut_cap = 4096;               // LDS capacity (entries) of the apply kernel's offset table
    // pinned host mirrors
    sarpro::PinnedBuf h_ghist, h_small, h_upload;
    sarpro::PinnedBuf mailbox;   // coherent pinned memory one-workgroup kernels post small results into (f32 flavour: no copy commands, no stream wait)
    uint32_t mail_seq = 0;       // the sequence word of the last post (word 0 of the mailbox)
    uint32_t mail_upload_slot = 0;
    // streaming ingest / egress: pinned ring on a side stream
    hipStream_t copy_stream = nullptr;
    hipEvent_t ring_evt[3] = {nullptr, nullptr, nullptr};
    sarpro::PinnedBuf h_ring;

    std::map<std::tuple<size_t, size_t, size_t, size_t, int>, sarpro::StripePlan *> plans;

    // per-kernel timing of the last call
    bool timing = false;
    bool async_dev = false;                      // SARPRO_HIP_CTX_ASYNC_DEV
    // resident batch (pipeline.cpp): the internal lanes -- contexts of their own (stream, workspaces, plans) that follow this
    // context's attributes --, the events that order the lanes' fused passes, and the lanes' kernel times of the last batch
    std::vector<sarpro_hip_ctx *> lanes;
    std::vector<hipEvent_t> pipe_events;
    hipEvent_t pipe_wait_before_fused = nullptr; // (on a lane, for ONE scene) the fused pass waits for this event ...
    hipEvent_t pipe_record_after_fused = nullptr; // ... and records this one behind itself
    hipEvent_t pipe_wait_before_hist = nullptr;   // PIPE_ORDER = 2: the scene's histogram pass waits for this event (the previous scene's fused pass is about to start) ...
    hipEvent_t pipe_record_before_fused = nullptr; // ... and the scene records this one in front of its own fused pass
    // PIPE_ORDER = 3 (the default of the resident batch): the two sweeps of all lanes in ONE order -- H(i + 1), F(i), H(i + 2), F(i + 1), ...
    hipEvent_t pipe_record_after_hist = nullptr;  // the scene records this one behind its histogram sweep
    bool pipe_defer = false;                      // the chain stops in front of its fused pass and parks the rest ...
    std::function<int()> pipe_deferred;           // ... here: the batch driver calls it once the NEXT scene's sweep is enqueued
    std::vector<std::pair<const char *, float>> lane_times;
    const void *last_final_hist = nullptr;       // the level histogram of the last CLAHE chain's exact kernels (device; sarpro_hip_ctx_chain_report)
    bool spec_ran = false;                       // the last u16 chain of this context took the speculative route (spec_state is that scene's)
    sarpro::PinnedBuf pipe_routes;               // ChainSpecState of every scene of the last batch (host copies)
    sarpro_hip_ctx *twin = nullptr;              // a second context on the same device (own stream, own workspaces): the other band of a dual-pol f32 product
    sarpro::BandWorker *band_worker = nullptr;   // ... and the thread that drives it
    bool f32_stripe_open = false;                // an open sarpro_hip_stripe_f32 owns the f32 workspace until its _end
    int timing_hold = 0;                         // > 0: timing_reset is a no-op (a composite call holds its steps' event pairs)
    bool async_pending = false;                  // event pairs of enqueued-but-unread calls are kept until read
    std::string time_only;                       // when set, only the kernel of this name is bracketed by events
    std::vector<sarpro::KernelTime> times;
    std::vector<std::pair<const char *, float>> host_times;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;

    // RCCL (lazy)
    void *rccl_lib = nullptr;
    void *comm = nullptr;
    int comm_nranks = 0, comm_rank = 0;
    struct sarpro_hip_local_group *local_group = nullptr; // the in-process communicator (comm.cpp), instead of RCCL: `comm` stays null
    sarpro::DevBuf local_tmp;                              // ... its rank-private sum buffer
    // measurement aid (bench.py `stripe_rank_model`, comm.cpp): the summed buffers of a stripe call's all-reduces, recorded on a rank of a
    // real N-rank run (COMM_RECORD) and played back to the same context running alone (COMM_REPLAY)
    std::vector<sarpro::DevBuf *> comm_saved;
    size_t comm_replay_pos = 0;
};

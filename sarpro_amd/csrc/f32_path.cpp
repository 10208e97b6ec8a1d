// f32_path.cpp -- orchestration of the f32-input flavour (arbitrary float samples: pol-op
// results, resampled reads, user arrays).  Same pass structure as the u16 flavour, with the
// per-DN tables replaced by f32 threshold tables (host_logic.cpp) resolved on the device by
// binary search.  No CPU fallback: all rasters come from the kernels in f32_kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "chain_kernels.h"
#include "f32_kernels.h"
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
#define RETCHK(expr)                                   \
    do {                                               \
        int rc__ = (expr);                             \
        if (rc__ != SARPRO_HIP_OK) return rc__;        \
    } while (0)

static int fail(sarpro_hip_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}
static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

namespace {

// device workspace layout (ctx->f32ws)
constexpr size_t kOffPartials = 0;                                   // 2048 x 32 B
constexpr size_t kOffThr4096 = 64 * 1024;                            // 16 KiB
constexpr size_t kOffHist4096 = kOffThr4096 + 16 * 1024;             // 32 KiB
constexpr size_t kOffThrLevel = kOffHist4096 + 32 * 1024;            // 256 KiB
constexpr size_t kOffTileBins = kOffThrLevel + 256 * 1024 + 4096;    // 128 KiB (the level table carries a sentinel entry)
constexpr size_t kOffLevelHist = kOffTileBins + 128 * 1024;          // 2 KiB
constexpr size_t kOffCdfs = kOffLevelHist + 4 * 1024;                // 128 KiB
constexpr size_t kOffMap = kOffCdfs + 128 * 1024;                    // 256 B
constexpr size_t kOffKeyHist = kOffMap + 4096;                       // zone route: 128 KiB sample histogram of the float's leading bits
constexpr size_t kOffZoneGe = kOffKeyHist + 128 * 1024;              // 2048 x 12 x 8 B
constexpr size_t kOffZoneN = kOffZoneGe + 2048 * 2 * kMaxZones * 8;  // 2048 x 4 B
constexpr size_t kOffZoneThr = kOffZoneN + 2048 * 4 * 4;             // (kZoneMaxThr + 2) x 4 B (zone_n: one count per wave)
constexpr size_t kOffZoneCounts = kOffZoneThr + 8192;                // (kZoneMaxThr + 1) x 8 B
constexpr size_t kOffSubHist = kOffZoneCounts + 16384;               // kMaxProbes x kSubKeys x 4 B
constexpr size_t kOffZoneWork = kOffSubHist + kMaxProbes * kSubKeys * 4;
constexpr size_t kOffZoneLut = kOffZoneWork + 4096;                  // kZoneLutKeys class bytes
constexpr size_t kWsBytes = kOffZoneLut + kZoneLutKeys;

struct F32Band {
    sarpro_hip_ctx *ctx = nullptr;
    const float *d_in = nullptr;
    size_t rows = 0, cols = 0, in_pitch = 0;
    int strategy = 0, bit_depth = 0, tamed = 0;
    void *d_out = nullptr;
    size_t out_pitch = 0;
    sarpro_hip_stats stats{};
    bool want_moments = true;   // mean / std of dB: reported statistics and the Adaptive strategy only
    F32Pol pol;                 // op >= 0: the samples are op(a, b) of two rasters, computed inside every kernel (d_in unused)
    size_t rows_total = 0, row0 = 0; // the scene this raster is a row stripe of (rows_total = 0: the raster is the scene)
    sarpro_hip_f32_partial local{}, global{};
    bool u8o = true, clahe = false, vec = false, empty = false;
    float t_valid = 0.f;
    double mean = 0.0, std_db = 0.0, min_db = 0.0, max_db = 0.0;
    StripePlan *plan = nullptr;
    bool stripe_handle = false, plan_held = false; // part of an open sarpro_hip_stripe_f32: the plan is reference-held between its phases
    // zone route (percentiles without the 4096-bin sweep): zones chosen from a row sample, resolved after the min / max pass
    bool allow_zones = false, use_zones = false, have_stats = false;
    int nz = 0, znp = 0, zgrid = 0;
    float zlo[kMaxZones], zhi[kMaxZones];
    uint32_t zcap = 0;
    uint64_t zgap[8];             // valid samples in the unkept buckets below run g of kept buckets
    int zrun[kMaxZones];          // the run zone j lies in
    uint64_t zone_kept = 0;       // samples the min / max pass kept in its side buffers
    bool zone_kept_known = false;
    const char *zone_note = "";
    uint64_t final_hist[256]{}; // histogram of the FINAL u8 raster (u8 output only)
    // small-scene direct route (percentile strategies): bins and levels evaluated in f64 on the device with a margin, the few samples
    // near a boundary settled by the host -- no threshold tables, no zone kernels, two synchronisations per band
    bool direct = false;
    uint32_t direct_queued = 0;     // samples the histogram pass queued (entries in ctx->f32zone)
    uint8_t *direct_host = nullptr; // host copy of the direct pass's results (mailbox or pinned buffer)
    uint64_t level_add[256]{};      // u8: levels of the samples the level pass queued (the kernel leaves them out of its histogram)
    bool level_hist_ready = false;  // direct route, u8: the level histogram came back with the level pass's queue (one synchronisation)
    uint64_t level_hist_host[256]{};
    bool idle = false;              // nothing was enqueued since the last synchronisation (the final one can be skipped)
};
constexpr uint32_t kQueueHead = 256; // queued samples fetched WITH their count (more than that: a second copy)
constexpr uint32_t kDirectQueueCap = 65536;
// ---- the mailbox: results posted into coherent pinned memory by a one-workgroup kernel, the host spins on a sequence word
constexpr size_t kMailBytes = 192 * 1024; // word 0: sequence; payload from byte 64 (<= 32 KiB - 64, or everything behind the upload slots)
static int mail_open(sarpro_hip_ctx *ctx) {
    if (!ctx->mailbox.p) {
        ctx->mailbox.flags = hipHostMallocCoherent | hipHostMallocMapped;
        HIPCHK(ctx, ctx->mailbox.reserve(kMailBytes));
        std::memset(ctx->mailbox.p, 0, kMailBytes);
        ctx->mail_seq = 0;
    }
    return SARPRO_HIP_OK;
}
static uint32_t *mail_flag(sarpro_hip_ctx *ctx) { return ctx->mailbox.as<uint32_t>(); }
template <typename T> static T *mail_payload(sarpro_hip_ctx *ctx, size_t off = 0) { return reinterpret_cast<T *>(ctx->mailbox.as<uint8_t>() + 64 + off); }
// waits for the post with this sequence number; a stream that went idle (or failed) without posting ends the wait
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
static int mail_wait(sarpro_hip_ctx *ctx, uint32_t seq) {
    volatile uint32_t *flag = mail_flag(ctx);
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SARPRO_HIP_OK;
        if ((spins & 0x3FFFu) == 0x3FFFu) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SARPRO_HIP_OK;
                return fail(ctx, SARPRO_HIP_ERR_HIP, "mailbox: the stream drained without the post");
            }
            if (q != hipErrorNotReady) HIPCHK(ctx, q);
        }
        cpu_relax();
        // a post that does not come within ~2 ms (a stalled device, a debugger, an oversubscribed host): stop burning the core
        if (spins > (1u << 18)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
// A batch of small host -> device words and zero fills: one kernel with the mailbox on, copy / fill commands without it.
// Upload sources come from upload_stage(): a slot of the mailbox (two, alternating: the previous upload may still be in flight;
// the one before it was consumed before the wait that preceded this call) or the pinned upload buffer.
struct Prep {
    PrepSegs g{};
    void upload(const void *h_src, void *d_dst, size_t bytes) { g.src[g.n] = h_src; g.dst[g.n] = d_dst; g.bytes[g.n] = (uint32_t)bytes; ++g.n; }
    void zero(void *d, size_t bytes) { g.zero[g.nz] = d; g.zbytes[g.nz] = (uint32_t)bytes; ++g.nz; }
};
static bool mail_enabled(const sarpro_hip_ctx *ctx);
constexpr size_t kUploadSlotBytes = 12 * 1024, kUploadOff = 32 * 1024, kMailBigOff = 64 * 1024; // big payloads: [64 KiB, 192 KiB)
static void *upload_stage(sarpro_hip_ctx *ctx, size_t bytes) {
    if (mail_enabled(ctx) && bytes <= kUploadSlotBytes && mail_open(ctx) == SARPRO_HIP_OK) {
        ctx->mail_upload_slot ^= 1u;
        return ctx->mailbox.as<uint8_t>() + kUploadOff + ctx->mail_upload_slot * kUploadSlotBytes;
    }
    return ctx->h_upload.p;
}
static int prep_run(sarpro_hip_ctx *ctx, const Prep &p) {
    bool by_kernel = mail_enabled(ctx);
    for (int k = 0; k < p.g.n; ++k) { // (a source outside the mailbox: the level tables, too large for a slot)
        const uint8_t *s = static_cast<const uint8_t *>(p.g.src[k]), *m = ctx->mailbox.as<uint8_t>();
        by_kernel = by_kernel && m && s >= m && s < m + kMailBytes;
    }
    if (by_kernel) { HIPCHK(ctx, launch_prep(p.g, ctx->stream)); return SARPRO_HIP_OK; }
    for (int k = 0; k < p.g.n; ++k) HIPCHK(ctx, hipMemcpyAsync(p.g.dst[k], p.g.src[k], p.g.bytes[k], hipMemcpyHostToDevice, ctx->stream));
    for (int k = 0; k < p.g.nz; ++k) HIPCHK(ctx, hipMemsetAsync(p.g.zero[k], 0, p.g.zbytes[k], ctx->stream));
    return SARPRO_HIP_OK;
}
static bool mail_enabled(const sarpro_hip_ctx *ctx) { return !ctx->attrs.on(A_NO_MAILBOX); } // (cross-check switch: copy / fill commands and stream waits, as round 2)

static uint32_t direct_queue_cap(const sarpro_hip_ctx *ctx) { // F32_DIRECT_QCAP: a tiny queue, so that the tests reach the overflow hand-back
    return (uint32_t)std::min<long long>(kDirectQueueCap, std::max<long long>(0, ctx->attrs.val(A_F32_DIRECT_QCAP, kDirectQueueCap)));
}
constexpr uint64_t kDirectMaxPx = 16ull << 20; // above this the zone route's fixed ~0.1 ms is small beside its sweeps

// estimate of a step table that is linear in dB: step = (dB(x) - low_db) / range_db * nsteps + bias, dB = 10 log10
F32StepEstimate step_estimate(const sarpro_hip_ctx *ctx, double low_db, double range_db, double nsteps, double bias, double gamma = 1.0) {
    F32StepEstimate e{};
    const double inv_x0 = std::pow(10.0, -low_db / 10.0), scale = 3.010299956639812 / range_db; // 10 log10(2)
    e.inv_x0 = (float)inv_x0; e.scale = (float)scale; e.bias = (float)bias; e.gamma = (float)gamma; e.nsteps = (float)nsteps;
    e.a_mul = (float)(scale * nsteps); e.b_add = (float)(std::log2(inv_x0) * scale * nsteps + bias);
    e.use = std::isfinite(e.inv_x0) && e.inv_x0 > 1e-30f && std::isfinite(e.scale) && e.scale * e.nsteps < 3.0e6f && std::isfinite(e.b_add) &&
            !ctx->attrs.on(A_NO_STEP_ESTIMATE);
    return e;
}

int rescale_in_place(F32Band &B, const uint64_t *level_hist) {
    sarpro_hip_ctx *ctx = B.ctx;
    uint8_t resc[256];
    bool identity = true;
    if (B.tamed) {
        for (int i = 0; i < 256; ++i) resc[i] = (uint8_t)i; // autoscale.rs:731-741 has no rescale
    } else {
        unsigned mn = 0, mx = 0;
        bool any = false;
        for (unsigned i = 0; i < 256; ++i)
            if (level_hist[i]) { if (!any) mn = i; mx = i; any = true; }
        u8_rescale_lut(mn, mx, resc); // autoscale.rs:348-364
        for (unsigned i = 0; i < 256; ++i)
            if (level_hist[i] && resc[i] != i) identity = false;
    }
    std::memset(B.final_hist, 0, sizeof(B.final_hist));
    for (int i = 0; i < 256; ++i) B.final_hist[resc[i]] += level_hist[i];
    if (!identity && B.rows && B.cols) {
        uint8_t *stage = ctx->h_upload.as<uint8_t>();
        std::memcpy(stage, resc, 256);
        uint8_t *d_map = ctx->f32ws.as<uint8_t>() + kOffMap;
        HIPCHK(ctx, hipMemcpyAsync(d_map, stage, 256, hipMemcpyHostToDevice, ctx->stream));
        B.idle = false;
        KernelTimer t(ctx, "remap_u8");
        HIPCHK(ctx, launch_remap_u8(reinterpret_cast<uint8_t *>(B.d_out), B.out_pitch, (uint32_t)B.rows, (uint32_t)B.cols, d_map, ctx->stream));
    }
    return SARPRO_HIP_OK;
}

// ---------------------------------------------------------------------------------------
// Zone route.  The 4096-bin histogram of autoscale.rs:102-117 exists to invert a handful of percentiles
// (autoscale.rs:120-140), and each inversion reads three numbers: the bin the target rank falls in, the count below that bin
// and the count in it.  A row sample (every ~32nd row, histogram of the float's leading 15 bits) says where each percentile
// lies to within a few of those buckets; the min / max pass counts the valid samples at or above each zone bound and keeps
// the samples inside the zones (a few per cent of the scene); once min / max are known the 4096-bin thresholds that fall
// inside a zone are counted against the kept samples.  Counts are exact, so the percentile is the reference's or -- when a
// zone missed its percentile, a workgroup's share of the side buffer overflowed, or the scene is small / degenerate -- the
// route is abandoned and the 4096-bin sweep runs as before.  Not used when the caller wants the full statistics
// (all eleven percentiles, dB moments) or for the Adaptive strategy, which reads most of them.
int needed_percentiles(int strategy, int tamed, double *p) {
    auto put = [&](std::initializer_list<double> v) { int n = 0; for (double x : v) p[n++] = x; return n; };
    if (tamed == kTamedCopol) return put({0.02, 0.05, 0.99});       // autoscale.rs:721-729
    if (tamed == kTamedCrosspol) return put({0.05, 0.99});
    switch (strategy) {
    case SARPRO_STRATEGY_STANDARD: return put({0.5, 0.25, 0.75, 0.02, 0.98});
    case SARPRO_STRATEGY_ROBUST: return put({0.25, 0.75, 0.01, 0.99});
    case SARPRO_STRATEGY_EQUALIZED:
    case SARPRO_STRATEGY_CLAHE: return put({0.01, 0.99});
    case SARPRO_STRATEGY_TAMED: return put({0.25, 0.99});
    case SARPRO_STRATEGY_DEFAULT: return put({0.05, 0.95});
    default: return 0; // Adaptive: skew, tail heaviness and five more
    }
}

void set_percentile(sarpro_hip_stats &s, double p, double v) {
    if (p == 0.5) s.median_db = v; else if (p == 0.01) s.p01 = v; else if (p == 0.02) s.p02 = v; else if (p == 0.05) s.p05 = v;
    else if (p == 0.10) s.p10 = v; else if (p == 0.25) s.p25 = v; else if (p == 0.75) s.p75 = v; else if (p == 0.90) s.p90 = v;
    else if (p == 0.95) s.p95 = v; else if (p == 0.98) s.p98 = v; else if (p == 0.99) s.p99 = v;
}

inline float f32_from_bits(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
inline uint32_t bits_of_f32(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }

// sample pass -> zones, all on the device (no host turn before the min / max pass): the sampled rows are stored (pol-op
// applied) and histogrammed by their leading 15 bits; one workgroup finds the bucket of every probe rank (each percentile
// -/+ delta); a second pass over the stored sample resolves those buckets to 2^-16; one wave builds the zones -- probe values
// widened by 2.5 bins of the estimated span, merged, given up when they would hold more than kZoneMaxMass of the samples.
constexpr float kZoneMaxMass = 0.10f;
int f32_zone_presample(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    B.use_zones = false;
    const long long zones_attr = ctx->attrs.val(A_F32_ZONES, 1); // 0: never (the 4096-bin sweep), 2 ("tiny"): no room in the side buffers
    if (zones_attr == 0) return SARPRO_HIP_OK;
    double pcts[8];
    const int np = needed_percentiles(B.strategy, B.tamed, pcts);
    const uint64_t px = (uint64_t)B.rows * B.cols;
    if (np == 0 || np > kMaxZones || B.want_moments || B.rows_total != B.rows || B.rows < 64 || B.cols < 64)
        return SARPRO_HIP_OK;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    uint32_t *d_keys = reinterpret_cast<uint32_t *>(ws + kOffKeyHist);
    uint32_t *d_sub = reinterpret_cast<uint32_t *>(ws + kOffSubHist);
    F32ZoneWork *d_work = reinterpret_cast<F32ZoneWork *>(ws + kOffZoneWork);
    const uint32_t stride = (uint32_t)std::max<uint64_t>(1, B.rows / 640);
    const uint32_t nsrows = ((uint32_t)B.rows + stride - 1) / stride;
    const uint32_t spitch = ((uint32_t)B.cols + 3) / 4 * 4;
    B.zgrid = f32_zone_grid((uint32_t)B.rows, (uint32_t)B.cols, B.vec);
    const double share = (double)px / (double)B.zgrid;
    B.zcap = (uint32_t)std::max(1024.0, share * (double)kZoneMaxMass) / 4 * 4; // a quarter per wave
    if (zones_attr == 2) B.zcap = 0; // test switch: no room at all, the first kept sample overflows
    // one buffer: the stored sample now, the kept samples of the min / max pass afterwards
    HIPCHK(ctx, ctx->f32zone.reserve(std::max((size_t)B.zgrid * B.zcap, (size_t)nsrows * spitch) * sizeof(float)));
    {
        Prep pr;
        pr.zero(d_keys, sizeof(uint32_t) * kSampleKeys);
        pr.zero(d_sub, sizeof(uint32_t) * kMaxProbes * kSubKeys);
        RETCHK(prep_run(ctx, pr));
    }
    F32ZoneSelectArgs sa{};
    sa.work = d_work; sa.key_hist = d_keys; sa.sub_hist = d_sub; sa.npcts = np; sa.t_valid = B.t_valid; sa.max_mass = kZoneMaxMass;
    for (int i = 0; i < np; ++i) sa.pcts[i] = pcts[i];
    sa.sample_fraction = stride == 1 ? 1.0f : (float)nsrows / (float)B.rows;
    sa.lut = ws + kOffZoneLut;
    {
        KernelTimer t(ctx, "f32_sample_keys");
        HIPCHK(ctx, launch_f32_sample_keys(B.d_in, B.in_pitch, (uint32_t)B.rows, (uint32_t)B.cols, B.t_valid, B.vec, stride, ctx->f32zone.as<float>(),
                                           spitch, d_keys, ctx->stream, B.pol));
    }
    {
        KernelTimer t(ctx, "f32_zone_select");
        HIPCHK(ctx, launch_f32_zone_pick(sa, ctx->stream));
        HIPCHK(ctx, launch_f32_sample_sub(ctx->f32zone.as<float>(), (uint64_t)nsrows * spitch, B.t_valid, d_work, d_sub, ctx->stream));
        HIPCHK(ctx, launch_f32_zone_finalize(sa, ctx->stream));
    }
    B.znp = np;
    B.use_zones = true;
    return SARPRO_HIP_OK;
}

// the min / max pass of the zone route: B.local, B.zgap, B.zone_kept, overflow check
int f32_zone_prepass(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    const int grid = B.zgrid;
    F32ZoneArgs a{};
    a.in = B.d_in; a.pitch = B.in_pitch; a.rows = (uint32_t)B.rows; a.cols = (uint32_t)B.cols; a.t_valid = B.t_valid; a.pol = B.pol;
    a.partials = reinterpret_cast<F32Partial *>(ws + kOffPartials);
    a.work = reinterpret_cast<const F32ZoneWork *>(ws + kOffZoneWork);
    a.gap_counts = reinterpret_cast<unsigned long long *>(ws + kOffZoneGe);
    a.lut = reinterpret_cast<const uint32_t *>(ws + kOffZoneLut);
    a.zone_buf = ctx->f32zone.as<float>();
    a.cap = B.zcap;
    a.zone_n = reinterpret_cast<uint32_t *>(ws + kOffZoneN);
    {
        KernelTimer t(ctx, "f32_prepass_zones");
        a.no_vec8 = ctx->attrs.on(A_F32_NO_VEC8) ? 1u : 0u;
        HIPCHK(ctx, launch_f32_prepass_zones(a, B.vec, grid, ctx->stream));
    }
    F32ZoneWork work_copy;
    const F32ZoneWork *h_work = &work_copy;
    bool overflow = false;
    std::memset(B.zgap, 0, sizeof(B.zgap));
    if (mail_enabled(ctx)) {
        RETCHK(mail_open(ctx));
        F32ZoneMail *mail = mail_payload<F32ZoneMail>(ctx);
        const uint32_t seq = ++ctx->mail_seq;
        HIPCHK(ctx, launch_f32_zone_post(a, grid, mail, mail_flag(ctx), seq, ctx->stream));
        RETCHK(mail_wait(ctx, seq));
        work_copy = mail->work;
        B.local.min_v = std::fmin(B.local.min_v, mail->min_v); B.local.max_v = std::fmax(B.local.max_v, mail->max_v);
        for (int k = 0; k < 7; ++k) B.zgap[k] = mail->gap[k];
        overflow = mail->overflow != 0;
        B.zone_kept = mail->kept; B.zone_kept_known = true;
    } else {
        uint8_t *h = ctx->h_small.as<uint8_t>();
        F32Partial *h_part = reinterpret_cast<F32Partial *>(h);
        uint64_t *h_ge = reinterpret_cast<uint64_t *>(h + sizeof(F32Partial) * 2048);
        uint32_t *h_n = reinterpret_cast<uint32_t *>(h + sizeof(F32Partial) * 2048 + 2048 * 2 * kMaxZones * 8);
        HIPCHK(ctx, hipMemcpyAsync(h_part, a.partials, sizeof(F32Partial) * (size_t)grid, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(h_ge, a.gap_counts, sizeof(uint64_t) * 8 * (size_t)grid, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(h_n, a.zone_n, sizeof(uint32_t) * 4 * (size_t)grid, hipMemcpyDeviceToHost, ctx->stream));
        F32ZoneWork *h_w = reinterpret_cast<F32ZoneWork *>(h + sizeof(F32Partial) * 2048 + 2048 * 2 * kMaxZones * 8 + 2048 * 4 * 4);
        HIPCHK(ctx, hipMemcpyAsync(h_w, a.work, sizeof(F32ZoneWork), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        work_copy = *h_w;
        B.zone_kept = 0; B.zone_kept_known = true;
        for (int i = 0; i < grid; ++i) {
            B.local.min_v = std::fmin(B.local.min_v, h_part[i].minv); B.local.max_v = std::fmax(B.local.max_v, h_part[i].maxv);
            for (int k = 0; k < 7; ++k) B.zgap[k] += h_ge[(size_t)i * 8 + k];
            for (int w = 0; w < 4; ++w) { overflow |= h_n[4 * i + w] > B.zcap / 4; B.zone_kept += h_n[4 * i + w]; }
        }
    }
    // every valid sample was either counted in a gap or kept
    for (int k = 0; k < 7; ++k) B.local.count += B.zgap[k];
    B.local.count += B.zone_kept;
    B.nz = std::min(std::max(h_work->nz, 0), kMaxZones);
    for (int i = 0; i < B.nz; ++i) { B.zlo[i] = h_work->bounds[2 * i]; B.zhi[i] = h_work->bounds[2 * i + 1]; B.zrun[i] = std::min(std::max(h_work->zone_run[i], 0), 6); }
    if (B.nz == 0) { B.use_zones = false; B.zone_note = "no zones (sample too small, or zones too heavy)"; }
    if (ctx->attrs.on(A_F32_ZONES_DEBUG)) {
        std::fprintf(stderr, "[zones] ns=%u kmin=%#x kmax=%#x nprobe=%u nz=%d mass_est=%.4f cap=%u grid=%d\n", h_work->ns, h_work->kmin, h_work->kmax,
                     h_work->nprobe, h_work->nz, (double)h_work->mass_est, B.zcap, B.zgrid);
        for (unsigned i = 0; i < h_work->nprobe && i < (unsigned)kMaxProbes; ++i)
            std::fprintf(stderr, "[zones]   probe %u: rank %u key %#x base %u\n", i, h_work->probe_rank[i], h_work->probe_key[i], h_work->probe_base[i]);
        for (int i = 0; i < 2 * kMaxZones; i += 2) std::fprintf(stderr, "[zones]   zone %d: [%g, %g)\n", i / 2, (double)h_work->bounds[i], (double)h_work->bounds[i + 1]);
    }
    if (overflow) { B.use_zones = false; B.zone_note = "side buffer overflow"; }
    return SARPRO_HIP_OK;
}

// min / max known: count the kept samples against the 4096-bin thresholds inside the zones -> the strategy's percentiles
int f32_zone_resolve(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    const sarpro_hip_f32_partial &G = B.global;
    B.have_stats = false;
    if (G.count == 0 || std::isinf(G.max_v)) return SARPRO_HIP_OK; // the ordinary route reports these
    const double min_db = db_of_f32(G.min_v), max_db = db_of_f32(G.max_v);
    if (std::fabs(max_db - min_db) < 2.220446049250313e-16) return SARPRO_HIP_OK;
    struct Zr { int b_lo, b_hi, s, e; } zr[kMaxZones];
    float tt[kZoneMaxThr + 2];
    int kk[kZoneMaxThr + 2]; // bin index k of tt[i] when it is a threshold, -1 for a zone bound
    int n = 0;
    tt[0] = -INFINITY; kk[0] = -1;
    for (int j = 0; j < B.nz; ++j) {
        const float hi_in = std::isinf(B.zhi[j]) ? G.max_v : std::nextafterf(B.zhi[j], 0.0f);
        zr[j].b_lo = bin4096_of_f32(std::fmax(B.zlo[j], G.min_v), min_db, max_db);
        zr[j].b_hi = bin4096_of_f32(std::fmin(hi_in, G.max_v), min_db, max_db);
        if (B.zlo[j] > G.max_v || hi_in < G.min_v) zr[j].b_hi = zr[j].b_lo; // zone outside the data: no thresholds
        const int nt = zr[j].b_hi - zr[j].b_lo;
        if (n + nt + 2 > kZoneMaxThr) { B.zone_note = "too many thresholds"; return SARPRO_HIP_OK; }
        zr[j].s = ++n; tt[n] = B.zlo[j]; kk[n] = -1;
        if (nt > 0) {
            build_bin4096_thresholds_range(min_db, max_db, zr[j].b_lo + 1, zr[j].b_hi, tt + n + 1);
            for (int t = 0; t < nt; ++t) kk[n + 1 + t] = zr[j].b_lo + 1 + t;
            n += nt;
        }
        zr[j].e = ++n; tt[n] = B.zhi[j]; kk[n] = -1;
    }
    for (int i = 2; i <= n; ++i)
        if (!(tt[i] >= tt[i - 1])) { B.zone_note = "threshold order"; return SARPRO_HIP_OK; }
    float *h_thr = static_cast<float *>(upload_stage(ctx, sizeof(float) * (size_t)(n + 1)));
    std::memcpy(h_thr, tt, sizeof(float) * (size_t)(n + 1));
    float *d_thr = reinterpret_cast<float *>(ws + kOffZoneThr);
    unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(ws + kOffZoneCounts);
    {
        Prep pr;
        pr.upload(h_thr, d_thr, sizeof(float) * (size_t)(n + 1));
        pr.zero(d_counts, sizeof(uint64_t) * (kZoneMaxThr + 1));
        RETCHK(prep_run(ctx, pr));
    }
    {
        KernelTimer t(ctx, "f32_zone_count");
        HIPCHK(ctx, launch_f32_zone_count(ctx->f32zone.as<float>(), reinterpret_cast<uint32_t *>(ws + kOffZoneN), B.zcap / 4, B.zgrid * 4, d_thr, n, d_counts,
                                          ctx->stream));
    }
    uint64_t *h_counts = ctx->h_small.as<uint64_t>();
    if (mail_enabled(ctx)) {
        RETCHK(mail_open(ctx));
        h_counts = mail_payload<uint64_t>(ctx);
        PostSegs g{};
        g.n = 1; g.src[0] = d_counts; g.dst[0] = h_counts; g.bytes[0] = (uint32_t)(sizeof(uint64_t) * (size_t)(n + 1));
        const uint32_t seq = ++ctx->mail_seq;
        HIPCHK(ctx, launch_post(g, mail_flag(ctx), seq, ctx->stream));
        RETCHK(mail_wait(ctx, seq));
    } else {
        HIPCHK(ctx, hipMemcpyAsync(h_counts, d_counts, sizeof(uint64_t) * (kZoneMaxThr + 1), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    // cum[i] = valid samples below tt[i], for every i inside a zone
    uint64_t cum[kZoneMaxThr + 2];
    const uint64_t N = G.count;
    // below a threshold inside run r: the gaps 0 .. r and the kept samples the count kernel found below it (kept samples between
    // the zones proper -- the rest of their buckets -- fall into the intervals between zone bounds and count here)
    uint64_t kept_below[kZoneMaxThr + 2];
    kept_below[0] = 0;
    for (int i = 0; i <= n; ++i) kept_below[i + 1] = kept_below[i] + h_counts[i]; // kept_below[i] = kept samples below tt[i], i >= 1
    for (int j = 0; j < B.nz; ++j) {
        uint64_t c = kept_below[zr[j].s];
        for (int g = 0; g <= B.zrun[j]; ++g) c += B.zgap[g];
        for (int i = zr[j].s; i <= zr[j].e; ++i) { cum[i] = c; c += h_counts[i]; }
        if (cum[zr[j].e] > N) { B.zone_note = "count mismatch"; return SARPRO_HIP_OK; }
    }
    { // every kept sample lies in exactly one interval of one zone
        uint64_t kept = 0, counted = kept_below[n + 1];
        kept = B.zone_kept;
        if (B.zone_kept_known && kept != counted) { B.zone_note = "count mismatch"; return SARPRO_HIP_OK; }
    }
    sarpro_hip_stats st{};
    st.valid_count = N; st.min_db = min_db; st.max_db = max_db;
    double pcts[8];
    const int np = needed_percentiles(B.strategy, B.tamed, pcts);
    for (int q = 0; q < np; ++q) {
        const uint64_t target = percentile_target(N, pcts[q]);
        bool done = false;
        for (int j = 0; j < B.nz && !done; ++j) {
            if (!(cum[zr[j].s] <= target && target < cum[zr[j].e])) continue;
            // the bin b with cum(thr_b) <= target < cum(thr_{b+1}): both thresholds must lie inside the zone
            for (int i = zr[j].s + 1; i < zr[j].e; ++i) {
                if (cum[i] > target) { // tt[i] = thr_{b+1}
                    const int b = kk[i] - 1;
                    uint64_t below;
                    if (i - 1 > zr[j].s) below = cum[i - 1];                      // tt[i-1] = thr_b
                    else if (cum[zr[j].s] == 0) below = 0;                          // nothing below the zone at all
                    else break;                                                     // thr_b lies below the zone
                    st.valid_count = N;
                    set_percentile(st, pcts[q], percentile_from_bin(min_db, max_db, b, target, below, cum[i] - below));
                    done = true;
                    break;
                }
            }
            if (!done && zr[j].e - 1 > zr[j].s && cum[zr[j].e] == N && kk[zr[j].e - 1] == 4095) { // last bin, open above
                const uint64_t below = cum[zr[j].e - 1];
                set_percentile(st, pcts[q], percentile_from_bin(min_db, max_db, 4095, target, below, N - below));
                done = true;
            }
        }
        if (!done) { B.zone_note = "percentile outside its zone"; return SARPRO_HIP_OK; }
    }
    B.min_db = min_db; B.max_db = max_db; B.mean = 0.0; B.std_db = 0.0;
    B.stats = st;
    RETCHK(select_window(&B.stats, B.strategy, B.tamed));
    B.have_stats = true;
    return SARPRO_HIP_OK;
}

// The band runs in five phases; between them sit the three reductions of a row-striped scene (SURVEY 8e row 1:
// autoscale.rs:35-117 needs count / min / max of the WHOLE scene before it can bin, the bins before it can select the
// window; autoscale.rs:572-608 needs whole-tile histograms).  One rank: f32_band_run runs them back to back.
//   a  prepass                 -> B.local   (count, min, max, dB moments)        merge: sum / min / max
//   b  4096-bin histogram      -> d_hist    (u64[4096], device)                  merge: sum
//   c  window; CLAHE tile bins -> d_tile_bins (u64[64*256], device; CLAHE only)  merge: sum
//   d  level / CLAHE apply     -> d_level_hist (u64[256], device; u8 only)       merge: sum
//   e  u8 rescale in place (autoscale.rs:348-364), final synchronisation
int f32_phase_a(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    B.u8o = B.tamed || B.bit_depth == SARPRO_BITDEPTH_U8;
    B.clahe = B.strategy == SARPRO_STRATEGY_CLAHE && !B.tamed;
    if (B.rows_total == 0) { B.rows_total = B.rows; B.row0 = 0; }
    if ((B.pol.op < 0 ? B.in_pitch : B.pol.pitch) < B.cols || B.out_pitch < B.cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    if (B.rows_total > 0x7FFFFFFFull || B.cols > 0x7FFFFFFFull) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "raster too large");
    if (B.row0 + B.rows > B.rows_total) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "stripe outside the scene");
    if (B.clahe && !clahe_shape_ok(B.rows_total, B.cols))
        return fail(ctx, SARPRO_HIP_ERR_UNSUPPORTED_SHAPE,
                    "CLAHE tile arithmetic underflows for this shape (reference panics: autoscale.rs:250,254)");
    std::memset(&B.stats, 0, sizeof(B.stats));
    std::memset(B.final_hist, 0, sizeof(B.final_hist));
    std::memset(&B.local, 0, sizeof(B.local));
    B.local.min_v = INFINITY; B.local.max_v = -INFINITY;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->f32ws.reserve(kWsBytes));
    HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands + 64 * 1024 + 65536 * 16));
    HIPCHK(ctx, ctx->h_upload.reserve(2 * 131072 + 2 * 64 * 256 * 8 + 66048 + 1024));
    B.vec = B.pol.op < 0 ? (B.in_pitch % 4 == 0 && aligned16(B.d_in))
                         : (B.pol.pitch % 4 == 0 && aligned16(B.pol.a) && aligned16(B.pol.b));
    B.t_valid = valid_threshold_f32();
    if (B.rows == 0 || B.cols == 0) return SARPRO_HIP_OK;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    const uint32_t rows = (uint32_t)B.rows, cols = (uint32_t)B.cols;
    B.use_zones = B.have_stats = false;
    // (CLAHE takes the direct route for its statistics only: tile histograms and the blend keep the host-built bin table)
    if (B.allow_zones && !B.direct) RETCHK(f32_zone_presample(B));
    if (B.use_zones) return f32_zone_prepass(B);
    const int pgrid = f32_prepass_grid(rows, cols, B.vec);
    F32Partial *d_part = reinterpret_cast<F32Partial *>(ws + kOffPartials);
    const bool moments = B.want_moments || (B.strategy == SARPRO_STRATEGY_ADAPTIVE && !B.tamed);
    {
        KernelTimer t(ctx, "f32_prepass");
        HIPCHK(ctx, launch_f32_prepass(B.d_in, B.in_pitch, rows, cols, B.t_valid, B.vec, moments, d_part, pgrid, ctx->stream, B.pol));
    }
    F32Partial *h_part = ctx->h_small.as<F32Partial>();
    const bool mail = mail_enabled(ctx);
    if (mail) RETCHK(mail_open(ctx));
    PostSegs pg{};
    if (mail) { // (big payload area: partials <= 64 KiB | 4096 bins 32 KiB | queue count | queue head 4 KiB)
        h_part = reinterpret_cast<F32Partial *>(ctx->mailbox.as<uint8_t>() + kMailBigOff);
        pg.src[0] = d_part; pg.dst[0] = h_part; pg.bytes[0] = (uint32_t)(sizeof(F32Partial) * (size_t)pgrid); pg.n = 1;
    } else {
        HIPCHK(ctx, hipMemcpyAsync(h_part, d_part, sizeof(F32Partial) * (size_t)pgrid, hipMemcpyDeviceToHost, ctx->stream));
    }
    B.direct_host = nullptr;
    if (B.direct) { // the 4096 bins in the same stream turn: the kernel merges the partials itself, nothing waits for the host
        unsigned long long *d_hist = reinterpret_cast<unsigned long long *>(ws + kOffHist4096);
        uint32_t *d_qn = reinterpret_cast<uint32_t *>(ws + kOffZoneCounts);
        HIPCHK(ctx, ctx->f32zone.reserve((size_t)kDirectQueueCap * sizeof(uint4)));
        {
            Prep pr;
            pr.zero(d_hist, sizeof(uint64_t) * 4096);
            pr.zero(d_qn, 4);
            RETCHK(prep_run(ctx, pr));
        }
        {
            KernelTimer t(ctx, "f32_hist4096_direct");
            HIPCHK(ctx, launch_f32_hist4096_direct(B.d_in, B.in_pitch, rows, cols, B.t_valid, B.vec, d_part, pgrid, d_hist, d_qn, ctx->f32zone.as<uint4>(),
                                                   direct_queue_cap(ctx), ctx->stream, B.pol));
        }
        uint8_t *h = mail ? ctx->mailbox.as<uint8_t>() + kMailBigOff + 64 * 1024 : ctx->h_small.as<uint8_t>() + 64 * 1024; // behind the partials (<= 2048 x 32 B)
        B.direct_host = h;
        if (mail) {
            pg.src[1] = d_hist; pg.dst[1] = h; pg.bytes[1] = sizeof(uint64_t) * 4096;
            pg.src[2] = d_qn; pg.dst[2] = h + 32768; pg.bytes[2] = 4;
            pg.src[3] = ctx->f32zone.p; pg.dst[3] = h + 32768 + 64; pg.bytes[3] = kQueueHead * 16; // usually all of them
            pg.n = 4;
        } else {
            HIPCHK(ctx, hipMemcpyAsync(h, d_hist, sizeof(uint64_t) * 4096, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(h + 32768, d_qn, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(h + 32768 + 64, ctx->f32zone.p, (size_t)kQueueHead * 16, hipMemcpyDeviceToHost, ctx->stream)); // usually all of them
        }
    }
    if (mail) {
        const uint32_t seq = ++ctx->mail_seq;
        HIPCHK(ctx, launch_post(pg, mail_flag(ctx), seq, ctx->stream));
        RETCHK(mail_wait(ctx, seq));
    } else {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    for (int i = 0; i < pgrid; ++i) {
        B.local.count += h_part[i].count; B.local.sum_db += h_part[i].sum; B.local.sumsq_db += h_part[i].sumsq;
        B.local.min_v = std::fmin(B.local.min_v, h_part[i].minv); B.local.max_v = std::fmax(B.local.max_v, h_part[i].maxv);
    }
    return SARPRO_HIP_OK;
}

// B.global holds the merged partial.  Leaves the local 4096-bin histogram in d_hist (zeros where the scene has none).
int f32_phase_b(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    const sarpro_hip_f32_partial &G = B.global;
    unsigned long long *d_hist = reinterpret_cast<unsigned long long *>(ws + kOffHist4096);
    HIPCHK(ctx, hipMemsetAsync(d_hist, 0, sizeof(uint64_t) * 4096, ctx->stream));
    B.empty = G.count == 0;
    if (B.empty) return SARPRO_HIP_OK; // autoscale.rs:376-378 / 466-468 / 716-718: zero raster
    if (std::isinf(G.max_v)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "non-finite (+inf) sample: the reference's statistics are undefined for it");
    B.mean = G.sum_db / (double)G.count;
    const double var = G.sumsq_db / (double)G.count - B.mean * B.mean;
    B.std_db = G.count > 1 ? std::sqrt(std::fmax(var, 0.0)) : 0.0;
    B.min_db = db_of_f32(G.min_v); B.max_db = db_of_f32(G.max_v);
    if (!(std::fabs(B.max_db - B.min_db) < 2.220446049250313e-16) && B.rows && B.cols) {
        float *thr = ctx->h_upload.as<float>();
        build_bin4096_thresholds(B.min_db, B.max_db, thr);
        float *d_thr = reinterpret_cast<float *>(ws + kOffThr4096);
        HIPCHK(ctx, hipMemcpyAsync(d_thr, thr, sizeof(float) * 4096, hipMemcpyHostToDevice, ctx->stream));
        KernelTimer t(ctx, "f32_hist4096");
        HIPCHK(ctx, launch_f32_hist4096(B.d_in, B.in_pitch, (uint32_t)B.rows, (uint32_t)B.cols, B.t_valid, B.vec, d_thr, d_hist,
                                        step_estimate(ctx, B.min_db, B.max_db - B.min_db, 4096.0, 0.0), ctx->stream, B.pol));
    }
    return SARPRO_HIP_OK;
}

// d_hist holds the merged bins.  CLAHE: leaves the local tile histograms in d_tile_bins.
int f32_phase_c(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    unsigned long long *d_tile_bins = reinterpret_cast<unsigned long long *>(ws + kOffTileBins);
    unsigned long long *d_level_hist = reinterpret_cast<unsigned long long *>(ws + kOffLevelHist);
    Prep pr;
    if (B.clahe) pr.zero(d_tile_bins, sizeof(uint64_t) * 64 * 256);
    pr.zero(d_level_hist, sizeof(uint64_t) * 256);
    pr.zero(ws + kOffZoneCounts, 4); // the level pass's queue counter (phase d; the zone counts were read before this phase)
    if (B.empty || !B.have_stats || !B.clahe || B.rows == 0 || B.cols == 0) { RETCHK(prep_run(ctx, pr)); pr = Prep(); }
    if (B.empty) return SARPRO_HIP_OK;
    if (!B.have_stats) {
        uint64_t *h_hist = ctx->h_small.as<uint64_t>();
        HIPCHK(ctx, hipMemcpyAsync(h_hist, ws + kOffHist4096, sizeof(uint64_t) * 4096, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        RETCHK(stats_from_bins4096(B.global.count, B.min_db, B.max_db, B.mean, B.std_db, h_hist, &B.stats));
        RETCHK(select_window(&B.stats, B.strategy, B.tamed));
    }
    if (!B.clahe || B.rows == 0 || B.cols == 0) return SARPRO_HIP_OK;
    RETCHK(get_plan(ctx, B.rows_total, B.cols, B.row0, B.rows, B.vec ? 4 : 1, &B.plan));
    if (B.stripe_handle && !B.plan_held) { ++B.plan->refs; B.plan_held = true; } // the plan cache never evicts a held plan (dropped in sarpro_hip_stripe_f32_end)
    float *thr = static_cast<float *>(upload_stage(ctx, sizeof(float) * 256));
    build_clahe_bin_thresholds(B.stats, thr);
    float *d_thr = reinterpret_cast<float *>(ws + kOffThrLevel);
    pr.upload(thr, d_thr, sizeof(float) * 256); // (with the zero fills above when the statistics were already known: one launch)
    RETCHK(prep_run(ctx, pr));
    if (B.plan->hist_rects_tiled.empty()) return SARPRO_HIP_OK;
    F32TileHistArgs ta{};
    ta.in = B.d_in; ta.pitch = B.in_pitch; ta.rects = B.plan->d_hist_rects_tiled.as<Rect>();
    ta.t_valid = B.t_valid; ta.thr = d_thr; ta.tile_bins = d_tile_bins;
    ta.pol = B.pol;
    ta.est = step_estimate(ctx, B.stats.low_clip, std::fmax(B.stats.high_clip - B.stats.low_clip, 1.0), 255.0, 0.5);
    KernelTimer t(ctx, "f32_tile_hist");
    HIPCHK(ctx, launch_f32_tile_hist(ta, (int)B.plan->hist_rects_tiled.size(), B.vec, ctx->stream));
    return SARPRO_HIP_OK;
}

// CLAHE: d_tile_bins holds the merged tile histograms.  Writes the stripe's levels; u8: local level histogram in d_level_hist.
int f32_phase_d(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    const bool u8o = B.u8o;
    const size_t esz = u8o ? 1 : 2;
    const uint32_t rows = (uint32_t)B.rows, cols = (uint32_t)B.cols;
    unsigned long long *d_level_hist = reinterpret_cast<unsigned long long *>(ws + kOffLevelHist);
    if (B.empty) {
        if (B.rows && B.cols) HIPCHK(ctx, hipMemset2DAsync(B.d_out, B.out_pitch * esz, 0, B.cols * esz, B.rows, ctx->stream));
        return SARPRO_HIP_OK;
    }
    if (B.rows == 0 || B.cols == 0) return SARPRO_HIP_OK;
    if (!B.clahe) {
        const int nlevels = u8o ? 255 : 65535;
        float *thr = ctx->h_upload.as<float>();
        float *d_thr = reinterpret_cast<float *>(ws + kOffThrLevel);
        F32LevelArgs a{};
        a.est = step_estimate(ctx, B.stats.low_clip, std::fmax(B.stats.high_clip - B.stats.low_clip, 1.0), (double)nlevels, 0.0, B.stats.gamma);
        a.in = B.d_in; a.out = B.d_out; a.in_pitch = B.in_pitch; a.out_pitch = B.out_pitch;
        a.rows = rows; a.cols = cols; a.t_valid = B.t_valid; a.thr = d_thr; a.level_hist = d_level_hist;
        a.pol = B.pol;
        const bool f64_levels = (!u8o || B.direct) && !ctx->attrs.on(A_NO_STEP_ESTIMATE);
        if (f64_levels) { // 65535 levels: f64 evaluation on the device, the reference's own arithmetic only near a level boundary
            a.low = B.stats.low_clip; a.high = B.stats.high_clip; a.range = std::fmax(B.stats.high_clip - B.stats.low_clip, 1.0);
            a.gamma = B.stats.gamma; a.max_val = (double)nlevels;
            if (a.gamma == 1.0 && !ctx->attrs.on(A_F32_LEVEL_GENERAL)) { // (cross-check switch: the general form at gamma == 1)
                const double inv_range = 1.0 / a.range;
                a.lin = 1;
                a.lin_a = 3.0102999566398120 * inv_range * a.max_val; // 10 log10(2)
                a.lin_b = -a.low * inv_range * a.max_val;
                a.lin_ymax = (a.high - a.low) * inv_range * a.max_val;
            }
        }
        // Without the 65535-entry table (1-2 ms of glibc per call): the samples within 1e-6 of a boundary -- a few hundred of
        // 4e8 -- are queued and settled here, with level_of_db.  SARPRO_HIP_F32_LEVEL_TABLE=1 (or a queue that overflows) builds
        // the table and lets the kernel search it instead.
        constexpr uint32_t kUqCapMax = 65536;
        // F32_LEVEL_QCAP: a tiny queue, so that the tests reach the overflow route below
        const uint32_t kUqCap = (uint32_t)std::min<long long>(kUqCapMax, std::max<long long>(0, ctx->attrs.val(A_F32_LEVEL_QCAP, kUqCapMax)));
        bool queued = f64_levels && !ctx->attrs.on(A_F32_LEVEL_TABLE);
        if (queued) {
            HIPCHK(ctx, ctx->f32zone.reserve((size_t)kUqCapMax * sizeof(uint4)));
            a.uq_count = reinterpret_cast<uint32_t *>(ws + kOffZoneCounts);
            a.uq_entries = ctx->f32zone.as<uint4>();
            a.uq_cap = kUqCap;
            a.t_first = level_threshold_one(B.stats, nlevels, 1); a.t_last = level_threshold_one(B.stats, nlevels, nlevels);
            a.f64_levels = 2;
            { // (a.uq_count was zeroed with phase c's fills)
                KernelTimer t(ctx, "f32_level");
                HIPCHK(ctx, launch_f32_level(a, B.vec, !u8o, ctx->stream));
            }
            // count, the first kQueueHead entries and (u8) the level histogram come back together: one turn
            uint32_t *h_n = ctx->h_small.as<uint32_t>();
            uint32_t *h_e = ctx->h_small.as<uint32_t>() + 4; // 16-byte aligned; reserved in phase a
            uint64_t *h_lh = reinterpret_cast<uint64_t *>(ctx->h_small.as<uint8_t>() + 16 + (size_t)kUqCapMax * 16);
            if (mail_enabled(ctx)) { // (the level raster is complete when the post arrives: it follows the level kernel in the stream)
                RETCHK(mail_open(ctx));
                uint8_t *m = mail_payload<uint8_t>(ctx);
                PostSegs g{};
                g.src[0] = a.uq_count; g.dst[0] = m; g.bytes[0] = 4;
                g.src[1] = a.uq_entries; g.dst[1] = m + 16; g.bytes[1] = kQueueHead * 16;
                g.n = 2;
                if (u8o) { g.src[2] = d_level_hist; g.dst[2] = m + 16 + kQueueHead * 16; g.bytes[2] = sizeof(uint64_t) * 256; g.n = 3; }
                const uint32_t seq = ++ctx->mail_seq;
                HIPCHK(ctx, launch_post(g, mail_flag(ctx), seq, ctx->stream));
                RETCHK(mail_wait(ctx, seq));
                std::memcpy(h_n, m, 4);
                std::memcpy(h_e, m + 16, (size_t)kQueueHead * 16);
                if (u8o) std::memcpy(h_lh, m + 16 + kQueueHead * 16, sizeof(uint64_t) * 256);
            } else {
                HIPCHK(ctx, hipMemcpyAsync(h_n, a.uq_count, 4, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(h_e, a.uq_entries, (size_t)kQueueHead * 16, hipMemcpyDeviceToHost, ctx->stream));
                if (u8o) HIPCHK(ctx, hipMemcpyAsync(h_lh, d_level_hist, sizeof(uint64_t) * 256, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
            B.idle = true;
            const uint32_t n = *h_n;
            if (n <= kUqCap) {
                if (u8o) { std::memcpy(B.level_hist_host, h_lh, sizeof(B.level_hist_host)); B.level_hist_ready = true; }
                if (n) {
                    if (n > kQueueHead) {
                        HIPCHK(ctx, hipMemcpyAsync(h_e, a.uq_entries, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
                        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                    }
                    uint32_t npatch = 0; // only the samples whose provisional level (entry.w) is not the reference's are rewritten
                    for (uint32_t i = 0; i < n; ++i) {
                        float x;
                        std::memcpy(&x, &h_e[4 * i + 2], 4);
                        const uint32_t lv = level_of_db(db_of_f32(x), B.stats.low_clip, B.stats.high_clip, B.stats.gamma, (double)nlevels);
                        if (u8o) B.level_add[lv & 255u] += 1; // the kernel left the queued samples out of its level histogram
                        if (lv != h_e[4 * i + 3]) {
                            h_e[4 * npatch + 0] = h_e[4 * i + 0]; h_e[4 * npatch + 1] = h_e[4 * i + 1]; h_e[4 * npatch + 2] = lv;
                            ++npatch;
                        }
                    }
                    if (npatch) {
                        HIPCHK(ctx, hipMemcpyAsync(a.uq_entries, h_e, (size_t)npatch * 16, hipMemcpyHostToDevice, ctx->stream));
                        KernelTimer t(ctx, "f32_level_patch");
                        if (u8o) HIPCHK(ctx, launch_patch_u8(reinterpret_cast<uint8_t *>(B.d_out), B.out_pitch, a.uq_entries, npatch, ctx->stream));
                        else HIPCHK(ctx, launch_patch_u16(reinterpret_cast<uint16_t *>(B.d_out), B.out_pitch, a.uq_entries, npatch, ctx->stream));
                        B.idle = false;
                    }
                }
                return SARPRO_HIP_OK;
            }
            queued = false; // overflow: the table route redoes the raster
            B.idle = false; // (a copy and a kernel follow: phase e must synchronise again before the call returns)
            if (u8o) { // (the table route of the u8 form counts every level itself: start from a clean histogram)
                HIPCHK(ctx, hipMemsetAsync(d_level_hist, 0, sizeof(uint64_t) * 256, ctx->stream));
                a.f64_levels = 0;
            }
        }
        build_level_thresholds(B.stats, nlevels, thr);
        thr[nlevels + 1] = INFINITY; // sentinel read by the estimate's verification
        HIPCHK(ctx, hipMemcpyAsync(d_thr, thr, sizeof(float) * (size_t)(nlevels + 2), hipMemcpyHostToDevice, ctx->stream));
        if (f64_levels && !u8o) { a.t_first = thr[1]; a.t_last = thr[nlevels]; a.f64_levels = 1; }
        KernelTimer t(ctx, "f32_level");
        HIPCHK(ctx, launch_f32_level(a, B.vec, !u8o, ctx->stream));
        return SARPRO_HIP_OK;
    }
    if (B.plan->apply_rects.empty()) return SARPRO_HIP_OK;
    double *d_cdfs = reinterpret_cast<double *>(ws + kOffCdfs);
    if (ctx->attrs.on(A_F32_HOST_CDFS)) { // the host twin of the kernel below (cross-check switch)
        uint64_t *h_tb = ctx->h_small.as<uint64_t>();
        HIPCHK(ctx, hipMemcpyAsync(h_tb, ws + kOffTileBins, sizeof(uint64_t) * 64 * 256, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        double *h_cdfs = reinterpret_cast<double *>(ctx->h_upload.as<uint8_t>() + 4096);
        RETCHK(clahe_cdfs(h_tb, B.rows_total, B.cols, h_cdfs));
        HIPCHK(ctx, hipMemcpyAsync(d_cdfs, h_cdfs, sizeof(double) * 64 * 256, hipMemcpyHostToDevice, ctx->stream));
    } else { // clip / redistribute / CDF per tile on the device (chain_kernels.hip, the u16 chain's kernel): no host turn
        KernelTimer t(ctx, "chain_cdfs");
        HIPCHK(ctx, launch_chain_cdfs(reinterpret_cast<const unsigned long long *>(ws + kOffTileBins), d_cdfs, (uint32_t)B.rows_total, (uint32_t)B.cols, 1,
                                      ctx->stream));
    }
    F32ClaheApplyArgs a{};
    a.in = B.d_in; a.out = B.d_out; a.in_pitch = B.in_pitch; a.out_pitch = B.out_pitch;
    a.rects = B.plan->d_apply_rects.as<Rect>(); a.cdfs = d_cdfs; a.t_valid = B.t_valid; a.thr = reinterpret_cast<float *>(ws + kOffThrLevel);
    a.row_w = B.plan->d_row_w.as<RowWeight>() + B.row0; // the table is indexed by the scene's row, the kernel by the stripe's
    a.col_w = B.plan->d_col_w.as<RowWeight>();
    a.level_hist = d_level_hist; a.max_val = u8o ? 255.0 : 65535.0;
    a.est = step_estimate(ctx, B.stats.low_clip, std::fmax(B.stats.high_clip - B.stats.low_clip, 1.0), 255.0, 0.5);
    a.pol = B.pol;
    KernelTimer t(ctx, "f32_clahe_apply");
    a.no_spec = ctx->attrs.on(A_NO_SPEC) ? 1u : 0u;
    HIPCHK(ctx, launch_f32_clahe_apply(a, (int)B.plan->apply_rects.size(), B.vec, !u8o, ctx->stream));
    return SARPRO_HIP_OK;
}

// u8: d_level_hist holds the merged histogram of the pre-rescale levels.
int f32_phase_e(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    if (B.empty) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        B.final_hist[0] = (uint64_t)B.rows_total * B.cols;
        return SARPRO_HIP_OK;
    }
    if (B.u8o) {
        uint64_t lh[256];
        if (B.level_hist_ready) {
            std::memcpy(lh, B.level_hist_host, sizeof(lh));
        } else {
            uint64_t *h_lh = ctx->h_small.as<uint64_t>();
            if (mail_enabled(ctx)) { // (everything before the post in the stream, the level raster included, is complete when it arrives)
                RETCHK(mail_open(ctx));
                h_lh = mail_payload<uint64_t>(ctx);
                PostSegs g{};
                g.n = 1; g.src[0] = ctx->f32ws.as<uint8_t>() + kOffLevelHist; g.dst[0] = h_lh; g.bytes[0] = sizeof(uint64_t) * 256;
                const uint32_t seq = ++ctx->mail_seq;
                HIPCHK(ctx, launch_post(g, mail_flag(ctx), seq, ctx->stream));
                RETCHK(mail_wait(ctx, seq));
            } else {
                HIPCHK(ctx, hipMemcpyAsync(h_lh, ctx->f32ws.as<uint8_t>() + kOffLevelHist, sizeof(uint64_t) * 256, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
            std::memcpy(lh, h_lh, sizeof(lh));
            B.idle = true;
        }
        for (int i = 0; i < 256; ++i) lh[i] += B.level_add[i];
        RETCHK(rescale_in_place(B, lh)); // (enqueues a remap kernel unless the rescale is the identity on the occupied levels: idle = false)
    }
    if (!B.idle) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// Direct route, after phase a: the bins came back with the partials; settle the queued samples with glibc, then the statistics
// and the window exactly as phase b / c compute them.  A queue that overflowed hands the band back to the threshold route.
int f32_direct_stats(F32Band &B) {
    sarpro_hip_ctx *ctx = B.ctx;
    const sarpro_hip_f32_partial &G = B.global;
    B.empty = G.count == 0;
    if (B.empty) return SARPRO_HIP_OK;
    if (std::isinf(G.max_v)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "non-finite (+inf) sample: the reference's statistics are undefined for it");
    uint8_t *h = B.direct_host; // where phase a had the bins, the queue count and the queue's head delivered
    uint64_t *h_hist = reinterpret_cast<uint64_t *>(h);
    uint32_t nq = 0;
    std::memcpy(&nq, h + 32768, 4);
    if (nq > direct_queue_cap(ctx)) { B.direct = false; return SARPRO_HIP_OK; }
    B.mean = G.sum_db / (double)G.count;
    const double var = G.sumsq_db / (double)G.count - B.mean * B.mean;
    B.std_db = G.count > 1 ? std::sqrt(std::fmax(var, 0.0)) : 0.0;
    B.min_db = db_of_f32(G.min_v); B.max_db = db_of_f32(G.max_v);
    if (nq && !(std::fabs(B.max_db - B.min_db) < 2.220446049250313e-16)) {
        uint32_t *h_e = reinterpret_cast<uint32_t *>(h + 32768 + 64);
        if (nq > kQueueHead) { // (the first kQueueHead entries came with the count; all of them: into the pinned buffer, the mailbox has no room)
            h_e = reinterpret_cast<uint32_t *>(ctx->h_small.as<uint8_t>() + 64 * 1024 + 32768 + 64);
            HIPCHK(ctx, hipMemcpyAsync(h_e, ctx->f32zone.p, (size_t)nq * 16, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        const double span = B.max_db - B.min_db, inv_span = 1.0 / span; // autoscale.rs:106-117
        for (uint32_t i = 0; i < nq; ++i) {
            float x;
            std::memcpy(&x, &h_e[4 * i + 2], 4);
            const double t = std::fmin(std::fmax((db_of_f32(x) - B.min_db) * inv_span, 0.0), 1.0);
            uint64_t idx = (uint64_t)(t * 4096.0);
            if (idx >= 4096) idx = 4095;
            h_hist[idx] += 1;
        }
    }
    B.direct_queued = nq;
    RETCHK(stats_from_bins4096(G.count, B.min_db, B.max_db, B.mean, B.std_db, h_hist, &B.stats));
    RETCHK(select_window(&B.stats, B.strategy, B.tamed));
    B.have_stats = true;
    return SARPRO_HIP_OK;
}

static bool f32_direct_wanted(const F32Band &B) {
    if (B.ctx->attrs.is_set(A_F32_DIRECT)) return B.ctx->attrs.val(A_F32_DIRECT, 0) != 0; // 0: never, 1: at every size (tests)
    return (uint64_t)B.rows * B.cols <= kDirectMaxPx && !B.ctx->attrs.on(A_NO_STEP_ESTIMATE);
}

int f32_band_run(F32Band &B) {
    if (B.ctx->f32_stripe_open) return fail(B.ctx, SARPRO_HIP_ERR_INVALID_ARG, "an f32 row stripe is open on this context: its phases share the context's f32 workspace");
    B.rows_total = B.rows; B.row0 = 0;
    B.allow_zones = true;
    B.direct = f32_direct_wanted(B);
    RETCHK(f32_phase_a(B));
    if (B.rows == 0 || B.cols == 0) return SARPRO_HIP_OK;
    B.global = B.local;
    if (B.direct && !B.use_zones) RETCHK(f32_direct_stats(B)); // (may hand the band back: direct = false, no statistics yet)
    if (B.use_zones) RETCHK(f32_zone_resolve(B));
    if (B.ctx->attrs.on(A_F32_ZONES_DEBUG)) std::fprintf(stderr, "[zones] answered=%d note='%s'\n", (int)B.have_stats, B.zone_note);
    if (B.have_stats) { B.empty = false; }
    else RETCHK(f32_phase_b(B));
    RETCHK(f32_phase_c(B));
    RETCHK(f32_phase_d(B));
    return f32_phase_e(B);
}

} // namespace

// ---------------------------------------------------------------------------------------
extern "C" int sarpro_hip_autoscale_band_f32_dev(sarpro_hip_ctx *ctx, const float *d_in, size_t rows, size_t cols,
                                                 size_t in_pitch, int strategy, int bit_depth, void *d_out,
                                                 size_t out_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if ((!d_in || !d_out) && rows * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    timing_reset(ctx);
    F32Band B;
    B.ctx = ctx; B.d_in = d_in; B.rows = rows; B.cols = cols; B.in_pitch = in_pitch;
    B.strategy = strategy; B.bit_depth = bit_depth; B.d_out = d_out; B.out_pitch = out_pitch;
    B.want_moments = stats_out != nullptr;
    int rc = f32_band_run(B);
    if (rc == SARPRO_HIP_OK && stats_out) *stats_out = B.stats;
    return rc;
}

// Fused pol-op -> autoscale (io/sentinel1.rs:1501-1578: ops.rs:4-44, then pipeline.rs:42-67 on the result): the op is
// computed inside every pass of the f32 flavour, the f32 pol-op raster is never materialised.
static int polop_band_dev(sarpro_hip_ctx *ctx, int op, const void *d_a, const void *d_b, int elem_u16, size_t rows, size_t cols, size_t in_pitch,
                          int strategy, int bit_depth, void *d_out, size_t out_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (op < SARPRO_OP_SUM || op > SARPRO_OP_LOGRATIO) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarisation operation");
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if ((!d_a || !d_b || !d_out) && rows * cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    F32Band B;
    B.ctx = ctx; B.rows = rows; B.cols = cols; B.in_pitch = in_pitch;
    B.pol.a = d_a; B.pol.b = d_b; B.pol.pitch = in_pitch; B.pol.op = op; B.pol.u16 = elem_u16;
    B.strategy = strategy; B.bit_depth = bit_depth; B.d_out = d_out; B.out_pitch = out_pitch;
    B.want_moments = stats_out != nullptr;
    int rc = f32_band_run(B);
    if (rc == SARPRO_HIP_OK && stats_out) *stats_out = B.stats;
    return rc;
}

extern "C" int sarpro_hip_polop_autoscale_band_f32_dev(sarpro_hip_ctx *ctx, int op, const float *d_a, const float *d_b, size_t rows, size_t cols,
                                                       size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                                       sarpro_hip_stats *stats_out) {
    if (ctx) timing_reset(ctx);
    return polop_band_dev(ctx, op, d_a, d_b, 0, rows, cols, in_pitch, strategy, bit_depth, d_out, out_pitch, stats_out);
}

extern "C" int sarpro_hip_polop_autoscale_band_u16_dev(sarpro_hip_ctx *ctx, int op, const uint16_t *d_a, const uint16_t *d_b, size_t rows,
                                                       size_t cols, size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                                       sarpro_hip_stats *stats_out) {
    if (ctx) timing_reset(ctx);
    return polop_band_dev(ctx, op, d_a, d_b, 1, rows, cols, in_pitch, strategy, bit_depth, d_out, out_pitch, stats_out);
}

// ---------------------------------------------------------------------------------------
// Row stripes of the f32 flavour (autoscale.rs:35-117 needs the scene's count / min / max before it can
// bin and the scene's bins before it can select the window).  One open f32 stripe per context: the phases keep their
// intermediate buffers in the context's workspace.
struct sarpro_hip_stripe_f32 {
    F32Band B;
    int phase = 0;
};

static int stripe_f32_begin(sarpro_hip_ctx *ctx, const float *d_in, int op, const void *d_a, const void *d_b, int elem_u16, size_t rows_total,
                            size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int bit_depth, void *d_out,
                            size_t out_pitch, sarpro_hip_stripe_f32 **out) {
    if (!ctx || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if (op >= 0 && (op < SARPRO_OP_SUM || op > SARPRO_OP_LOGRATIO)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarisation operation");
    if (rows_local * cols && (!d_out || (op < 0 ? !d_in : (!d_a || !d_b)))) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (row0 + rows_local > rows_total) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "stripe outside the scene");
    if (ctx->f32_stripe_open) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "one open f32 stripe per context (its phases share the context's f32 workspace)");
    sarpro_hip_stripe_f32 *s = new sarpro_hip_stripe_f32();
    ctx->f32_stripe_open = true;
    F32Band &B = s->B;
    B.stripe_handle = true;
    B.ctx = ctx; B.d_in = d_in; B.rows = rows_local; B.cols = cols; B.in_pitch = in_pitch;
    B.rows_total = rows_total; B.row0 = row0;
    if (op >= 0) { B.pol.a = d_a; B.pol.b = d_b; B.pol.pitch = in_pitch; B.pol.op = op; B.pol.u16 = elem_u16; }
    B.strategy = strategy; B.bit_depth = bit_depth; B.d_out = d_out; B.out_pitch = out_pitch;
    B.want_moments = true;
    timing_reset(ctx);
    *out = s;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_begin_f32(sarpro_hip_ctx *ctx, const float *d_in, size_t rows_total, size_t cols, size_t row0,
                                           size_t rows_local, size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                           sarpro_hip_stripe_f32 **out) {
    return stripe_f32_begin(ctx, d_in, -1, nullptr, nullptr, 0, rows_total, cols, row0, rows_local, in_pitch, strategy, bit_depth, d_out, out_pitch, out);
}

extern "C" int sarpro_hip_stripe_begin_polop(sarpro_hip_ctx *ctx, int op, const void *d_a, const void *d_b, int elem_u16, size_t rows_total,
                                             size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int bit_depth,
                                             void *d_out, size_t out_pitch, sarpro_hip_stripe_f32 **out) {
    if (op < 0) return ctx ? fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad polarisation operation") : SARPRO_HIP_ERR_INVALID_ARG;
    return stripe_f32_begin(ctx, nullptr, op, d_a, d_b, elem_u16 ? 1 : 0, rows_total, cols, row0, rows_local, in_pitch, strategy, bit_depth, d_out,
                            out_pitch, out);
}

extern "C" int sarpro_hip_stripe_f32_phase1(sarpro_hip_stripe_f32 *s, sarpro_hip_f32_partial *local_out) {
    if (!s || !local_out || s->phase != 0) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(f32_phase_a(s->B));
    *local_out = s->B.local;
    s->phase = 1;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_host_f32_merge_partials(const sarpro_hip_f32_partial *parts, size_t n, sarpro_hip_f32_partial *out) {
    if ((!parts && n) || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    sarpro_hip_f32_partial g{};
    g.min_v = INFINITY; g.max_v = -INFINITY;
    for (size_t i = 0; i < n; ++i) { // rank order: the f64 sums are reproducible for a given stripe plan
        g.count += parts[i].count; g.sum_db += parts[i].sum_db; g.sumsq_db += parts[i].sumsq_db;
        g.min_v = std::fmin(g.min_v, parts[i].min_v); g.max_v = std::fmax(g.max_v, parts[i].max_v);
    }
    *out = g;
    return SARPRO_HIP_OK;
}

static int stripe_f32_buf(sarpro_hip_stripe_f32 *s, size_t off, size_t n, bool present, uint64_t **d_buf, size_t *count) {
    HIPCHK(s->B.ctx, hipStreamSynchronize(s->B.ctx->stream)); // the buffer is complete when the phase returns
    *d_buf = present ? reinterpret_cast<uint64_t *>(s->B.ctx->f32ws.as<uint8_t>() + off) : nullptr;
    *count = present ? n : 0;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_f32_phase2(sarpro_hip_stripe_f32 *s, const sarpro_hip_f32_partial *global, uint64_t **d_buf, size_t *count) {
    if (!s || !global || !d_buf || !count || s->phase != 1) return SARPRO_HIP_ERR_INVALID_ARG;
    s->B.global = *global;
    RETCHK(f32_phase_b(s->B));
    s->phase = 2;
    return stripe_f32_buf(s, kOffHist4096, 4096, true, d_buf, count);
}

extern "C" int sarpro_hip_stripe_f32_phase3(sarpro_hip_stripe_f32 *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 2) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(f32_phase_c(s->B));
    s->phase = 3;
    return stripe_f32_buf(s, kOffTileBins, 64 * 256, s->B.clahe, d_buf, count);
}

extern "C" int sarpro_hip_stripe_f32_phase4(sarpro_hip_stripe_f32 *s, uint64_t **d_buf, size_t *count) {
    if (!s || !d_buf || !count || s->phase != 3) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(f32_phase_d(s->B));
    s->phase = 4;
    return stripe_f32_buf(s, kOffLevelHist, 256, s->B.u8o, d_buf, count);
}

extern "C" int sarpro_hip_stripe_f32_phase5(sarpro_hip_stripe_f32 *s, sarpro_hip_stats *stats_out) {
    if (!s || s->phase != 4) return SARPRO_HIP_ERR_INVALID_ARG;
    RETCHK(f32_phase_e(s->B));
    if (stats_out) *stats_out = s->B.stats;
    s->phase = 5;
    return SARPRO_HIP_OK;
}

extern "C" void sarpro_hip_stripe_f32_end(sarpro_hip_stripe_f32 *s) {
    if (!s) return;
    if (s->B.plan && s->B.plan_held) --s->B.plan->refs;
    if (s->B.ctx) s->B.ctx->f32_stripe_open = false;
    delete s;
}

// the stripe in one call per rank, reductions over the library's communicator
static int stripe_f32_run(sarpro_hip_stripe_f32 *s, sarpro_hip_stats *stats_out) {
    F32Band &B = s->B;
    sarpro_hip_ctx *ctx = B.ctx;
    if (!ctx->comm && !ctx->local_group) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "communicator not initialised");
    const int n = ctx->comm_nranks, me = ctx->comm_rank;
    if (n > 1024) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "more than 1024 ranks");
    // A rank that fails locally (bad pitch, a HIP error in its first pass, no plan) must not leave its peers waiting in a
    // collective: its status travels WITH the first reduction (one more word behind the gathered partials) and in a one-word
    // reduction after the phase that builds plans and tables; every rank then returns an error together.  A failure between later
    // collectives (device faults) still needs the communicator torn down by the caller.
    const int rc_a = f32_phase_a(B);
    // all-gather of the 32-byte partials as an all-reduce(sum) of a buffer that is zero outside the rank's own slot: RCCL adds
    // the words as u64, x + 0 + ... + 0 is exact whatever bit pattern x is
    static_assert(sizeof(sarpro_hip_f32_partial) == 32, "partial = 4 words");
    HIPCHK(ctx, ctx->f32ws.reserve(kWsBytes)); // (phase a may have failed before it did)
    uint64_t *d_g = reinterpret_cast<uint64_t *>(ctx->f32ws.as<uint8_t>() + kOffPartials);
    HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
    uint64_t *h_g = ctx->h_small.as<uint64_t>();
    HIPCHK(ctx, hipMemsetAsync(d_g, 0, 32 * (size_t)n + 8, ctx->stream));
    if (rc_a == SARPRO_HIP_OK) std::memcpy(h_g, &B.local, 32); else std::memset(h_g, 0, 32);
    h_g[4] = rc_a == SARPRO_HIP_OK ? 0u : 1u;
    HIPCHK(ctx, hipMemcpyAsync(d_g + 4 * me, h_g, 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_g + 4 * (size_t)n, h_g + 4, 8, hipMemcpyHostToDevice, ctx->stream));
    RETCHK(comm_allreduce_sum_u64_async(ctx, d_g, 4 * (size_t)n + 1));
    HIPCHK(ctx, hipMemcpyAsync(h_g, d_g, 32 * (size_t)n + 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (rc_a != SARPRO_HIP_OK) return rc_a;
    if (h_g[4 * (size_t)n] != 0) return fail(ctx, SARPRO_HIP_ERR_RCCL, "another rank failed in the first pass of the stripe; every rank returns");
    std::vector<sarpro_hip_f32_partial> parts((size_t)n);
    std::memcpy(parts.data(), h_g, 32 * (size_t)n);
    RETCHK(sarpro_hip_host_f32_merge_partials(parts.data(), parts.size(), &B.global));
    uint8_t *ws = ctx->f32ws.as<uint8_t>();
    RETCHK(f32_phase_b(B));
    RETCHK(comm_allreduce_sum_u64_async(ctx, reinterpret_cast<uint64_t *>(ws + kOffHist4096), 4096));
    {
        const int rc_c = f32_phase_c(B);
        h_g[0] = rc_c == SARPRO_HIP_OK ? 0u : 1u;
        uint64_t *d_s = d_g + 4 * (size_t)n; // (the gathered partials are consumed)
        HIPCHK(ctx, hipMemcpyAsync(d_s, h_g, 8, hipMemcpyHostToDevice, ctx->stream));
        RETCHK(comm_allreduce_sum_u64_async(ctx, d_s, 1));
        HIPCHK(ctx, hipMemcpyAsync(h_g, d_s, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (rc_c != SARPRO_HIP_OK) return rc_c;
        if (h_g[0] != 0) return fail(ctx, SARPRO_HIP_ERR_RCCL, "another rank failed while building its plan and tables; every rank returns");
    }
    if (B.clahe) RETCHK(comm_allreduce_sum_u64_async(ctx, reinterpret_cast<uint64_t *>(ws + kOffTileBins), 64 * 256));
    RETCHK(f32_phase_d(B));
    if (B.u8o) RETCHK(comm_allreduce_sum_u64_async(ctx, reinterpret_cast<uint64_t *>(ws + kOffLevelHist), 256));
    RETCHK(f32_phase_e(B));
    if (stats_out) *stats_out = B.stats;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_stripe_run_f32(sarpro_hip_ctx *ctx, const float *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                         size_t in_pitch, int strategy, int bit_depth, void *d_out, size_t out_pitch,
                                         sarpro_hip_stats *stats_out) {
    sarpro_hip_stripe_f32 *s = nullptr;
    int rc = sarpro_hip_stripe_begin_f32(ctx, d_in, rows_total, cols, row0, rows_local, in_pitch, strategy, bit_depth, d_out, out_pitch, &s);
    if (rc == SARPRO_HIP_OK) {
        rc = stripe_f32_run(s, stats_out);
        sarpro_hip_stripe_f32_end(s);
    }
    if (rc != SARPRO_HIP_OK) comm_abort_local_group(ctx); // (an in-process group: the peers must not wait for this rank)
    return rc;
}

namespace sarpro {
// One f32 band's row stripe through autoscale_db_image_tamed_synrgb_u8 (autoscale.rs:710-742; tamed = 1 co-pol, 2 cross-pol): the
// band-specific re-autoscale of save.rs:324-351 over stripes.  The same collectives as sarpro_hip_stripe_run_f32 at U8 -- the
// Tamed windows (p02 / p05 .. p99) come from the same all-reduced 4096 bins, the level histogram joins its all-reduce although
// a9 has no u8 rescale (every rank takes the same sequence).
int stripe_run_f32_tamed(sarpro_hip_ctx *ctx, const float *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int tamed,
                         uint8_t *d_out, size_t out_pitch) {
    sarpro_hip_stripe_f32 *s = nullptr;
    int rc = stripe_f32_begin(ctx, d_in, -1, nullptr, nullptr, 0, rows_total, cols, row0, rows_local, in_pitch, SARPRO_STRATEGY_TAMED, SARPRO_BITDEPTH_U8, d_out, out_pitch, &s);
    if (rc == SARPRO_HIP_OK) {
        s->B.tamed = tamed;
        rc = stripe_f32_run(s, nullptr);
        sarpro_hip_stripe_f32_end(s);
    }
    if (rc != SARPRO_HIP_OK) comm_abort_local_group(ctx);
    return rc;
}
} // namespace sarpro

extern "C" int sarpro_hip_stripe_run_polop(sarpro_hip_ctx *ctx, int op, const void *d_a, const void *d_b, int elem_u16, size_t rows_total,
                                           size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int strategy, int bit_depth,
                                           void *d_out, size_t out_pitch, sarpro_hip_stats *stats_out) {
    sarpro_hip_stripe_f32 *s = nullptr;
    int rc = sarpro_hip_stripe_begin_polop(ctx, op, d_a, d_b, elem_u16, rows_total, cols, row0, rows_local, in_pitch, strategy, bit_depth, d_out,
                                           out_pitch, &s);
    if (rc == SARPRO_HIP_OK) {
        rc = stripe_f32_run(s, stats_out);
        sarpro_hip_stripe_f32_end(s);
    }
    if (rc != SARPRO_HIP_OK) comm_abort_local_group(ctx);
    return rc;
}

// Self-test of the division the u16 pol-op kernels use (f32_kernels.hip div_small_ints): all 2^32 pairs of u16 values against the
// compiler's IEEE division.  *mismatches_out = 0 is the proof the fused pol-op rests on.
extern "C" int sarpro_hip_selftest_polop_division(sarpro_hip_ctx *ctx, uint64_t *mismatches_out) {
    if (!ctx || !mismatches_out) return SARPRO_HIP_ERR_INVALID_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->f32ws.reserve(kWsBytes));
    unsigned long long *d = reinterpret_cast<unsigned long long *>(ctx->f32ws.as<uint8_t>() + kOffZoneCounts);
    HIPCHK(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    HIPCHK(ctx, launch_selftest_div_small_ints(d, ctx->stream));
    unsigned long long h = 0;
    HIPCHK(ctx, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches_out = h;
    return SARPRO_HIP_OK;
}

static int host_polop_band(sarpro_hip_ctx *ctx, int op, const void *a, const void *b, int elem_u16, size_t rows, size_t cols, int strategy,
                           int bit_depth, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    const bool u8o = bit_depth == SARPRO_BITDEPTH_U8;
    if (rows * cols && (!a || !b || (u8o ? (void *)out_u8 : (void *)out_u16) == nullptr)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    timing_reset(ctx);
    const size_t esz = elem_u16 ? 2 : 4;
    size_t pitch = 0, pitch2 = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], a, rows, cols, esz, &pitch));
    RETCHK(stage_in_2d(ctx, ctx->stage_in[1], b, rows, cols, esz, &pitch2));
    const size_t osz = u8o ? 1 : 2;
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(rows, 1) * pitch * osz));
    RETCHK(polop_band_dev(ctx, op, ctx->stage_in[0].p, ctx->stage_in[1].p, elem_u16, rows, cols, pitch, strategy, bit_depth, ctx->stage_out[0].p,
                          pitch, stats_out));
    return fetch_out_2d(ctx, u8o ? (void *)out_u8 : (void *)out_u16, ctx->stage_out[0].p, pitch * osz, cols * osz, rows);
}

extern "C" int sarpro_hip_polop_autoscale_band_f32(sarpro_hip_ctx *ctx, int op, const float *a, const float *b, size_t rows, size_t cols,
                                                   int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    return host_polop_band(ctx, op, a, b, 0, rows, cols, strategy, bit_depth, out_u8, out_u16, stats_out);
}

extern "C" int sarpro_hip_polop_autoscale_band_u16(sarpro_hip_ctx *ctx, int op, const uint16_t *a, const uint16_t *b, size_t rows, size_t cols,
                                                   int strategy, int bit_depth, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    return host_polop_band(ctx, op, a, b, 1, rows, cols, strategy, bit_depth, out_u8, out_u16, stats_out);
}

static int host_band_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols, int strategy, int bit_depth,
                         int tamed, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!tamed && (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    const bool u8o = tamed || bit_depth == SARPRO_BITDEPTH_U8;
    if (rows * cols && (!in || (u8o ? (void *)out_u8 : (void *)out_u16) == nullptr)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    timing_reset(ctx);
    size_t pitch = 0;
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], in, rows, cols, 4, &pitch));
    const size_t osz = u8o ? 1 : 2;
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(rows, 1) * pitch * osz));
    F32Band B;
    B.ctx = ctx; B.d_in = ctx->stage_in[0].as<float>(); B.rows = rows; B.cols = cols; B.in_pitch = pitch;
    B.strategy = tamed ? SARPRO_STRATEGY_TAMED : strategy; B.bit_depth = u8o ? SARPRO_BITDEPTH_U8 : SARPRO_BITDEPTH_U16;
    B.tamed = tamed; B.d_out = ctx->stage_out[0].p; B.out_pitch = pitch;
    B.want_moments = stats_out != nullptr;
    RETCHK(f32_band_run(B));
    if (stats_out) *stats_out = B.stats;
    return fetch_out_2d(ctx, u8o ? (void *)out_u8 : (void *)out_u16, ctx->stage_out[0].p, pitch * osz, cols * osz, rows);
}

extern "C" int sarpro_hip_autoscale_band_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols, int strategy,
                                             int bit_depth, uint8_t *out_u8, uint16_t *out_u16, sarpro_hip_stats *stats_out) {
    return host_band_f32(ctx, in, rows, cols, strategy, bit_depth, 0, out_u8, out_u16, stats_out);
}

extern "C" int sarpro_hip_tamed_synrgb_u8_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols, int is_copol,
                                              uint8_t *out_u8) {
    return host_band_f32(ctx, in, rows, cols, SARPRO_STRATEGY_TAMED, SARPRO_BITDEPTH_U8, is_copol ? kTamedCopol : kTamedCrosspol,
                         out_u8, nullptr, nullptr);
}

extern "C" int sarpro_hip_db_mask_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols, double *db_out,
                                      uint8_t *mask_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    const size_t n = rows * cols;
    if (n && !in) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (!n || (!db_out && !mask_out)) return SARPRO_HIP_OK;
    timing_reset(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, ctx->stage_in[0].reserve(n * 4));
    if (db_out) HIPCHK(ctx, ctx->stage_out[0].reserve(n * 8));
    if (mask_out) HIPCHK(ctx, ctx->stage_out[1].reserve(n));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_in[0].p, in, n * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        KernelTimer t(ctx, "db_mask_f32");
        HIPCHK(ctx, launch_db_mask_f32(ctx->stage_in[0].as<float>(), n, valid_threshold_f32(),
                                       db_out ? ctx->stage_out[0].as<double>() : nullptr,
                                       mask_out ? ctx->stage_out[1].as<uint8_t>() : nullptr, ctx->stream));
    }
    if (db_out) HIPCHK(ctx, hipMemcpyAsync(db_out, ctx->stage_out[0].p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out) HIPCHK(ctx, hipMemcpyAsync(mask_out, ctx->stage_out[1].p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SARPRO_HIP_OK;
}

// The dual-pol product of f32 bands -- what the reference's DEFAULT flow feeds the raster core: `--size N` resamples on read
// (sentinel1.rs:1074-1108), so both bands arrive as non-integer f32.  One implementation behind every entry point:
//   src   host arrays (staged through stage_in[0]) or device rasters with a pitch;
//   flow  save.rs:317-367 (pipeline(U8) per band; under Tamed the bands are re-autoscaled band-specifically, autoscale.rs:710-742)
//         or, with SARPRO_HIP_DUALPOL_PLAIN_PIPELINE, api/mod.rs:404-437 (process_safe_to_buffer_with_mode: BOTH bands through
//         process_scalar_data_pipeline with the caller's strategy, no Tamed re-autoscale);
//   tail  native resolution (levels -> tables from the combined level histogram -> compose pass), or the resized flow of both
//         files: per band resize -> pad, then the composition on the final rasters (the suppressed floor sees the padding).
struct DualF32Src { const float *host[2]; const float *dev[2]; size_t dev_pitch; };
struct DualF32Out { uint8_t *rgb_host, *rgb_dev; size_t rgb_dev_pitch_px; uint8_t *u8_host[2], *u8_dev[2]; size_t u8_dev_pitch; };

static int dualpol_f32_impl(sarpro_hip_ctx *ctx, const DualF32Src &src, size_t rows, size_t cols, int strategy, int mode, unsigned flags,
                            bool resized, size_t target_size, int pad, const DualF32Out &out, sarpro_hip_stats *stats_out,
                            sarpro_hip_resize_meta *meta) {
    if (strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad strategy");
    if (mode < 0 || mode > SARPRO_SYNRGB_ENHANCED) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad synrgb mode");
    if (flags & ~SARPRO_HIP_DUALPOL_PLAIN_PIPELINE) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "unknown dual-pol flag");
    TimingHold hold(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t fc = cols, fr = rows;
    if (resized) RETCHK(sarpro_hip_resize_output_dims(cols, rows, target_size, pad, &fc, &fr));
    sarpro_hip_resize_meta m{};
    if (!resized) { m.final_cols = cols; m.final_rows = rows; m.scale_x = m.scale_y = 1.0; }
    if (rows * cols == 0) { if (meta) *meta = m; return SARPRO_HIP_OK; } // (the resized flow of an empty raster has nothing to compose either)
    const bool plain = (flags & SARPRO_HIP_DUALPOL_PLAIN_PIPELINE) != 0;
    const size_t r1 = std::max<size_t>(rows, 1);
    const size_t opitch = round_up(std::max<size_t>(fc, 1), 64);
    uint64_t combined[256];
    std::memset(combined, 0, sizeof(combined));
    const size_t lpitch_b = round_up(std::max<size_t>(cols, 1), 64);
    // One band: autoscale to u8 levels (levels[b], or a scratch raster of the WORKING context when a resize follows), the
    // resize + pad into resized[b], the optional copy of the levels.  `w` is the context whose stream and f32 workspaces do the
    // work: the caller's, or its twin for the second band (buffers named through `ctx` were reserved by the caller beforehand).
    struct BandResult { uint64_t hist[256]; sarpro_hip_stats stats; sarpro_hip_resize_meta m; };
    BandResult res[2];
    for (int b = 0; b < 2; ++b) { std::memset(&res[b], 0, sizeof(res[b])); res[b].m = m; }
    auto one_band = [&](sarpro_hip_ctx *w, int b, const float *d_in, size_t pitch) -> int {
        uint8_t *lv = resized ? w->stage_out[0].as<uint8_t>() : ctx->levels[b].as<uint8_t>();
        F32Band B;
        B.ctx = w; B.d_in = d_in; B.rows = rows; B.cols = cols; B.in_pitch = pitch;
        B.strategy = strategy; B.bit_depth = SARPRO_BITDEPTH_U8;
        B.tamed = (!plain && strategy == SARPRO_STRATEGY_TAMED) ? (b == 0 ? kTamedCopol : kTamedCrosspol) : 0;
        B.d_out = lv; B.out_pitch = lpitch_b;
        B.want_moments = stats_out != nullptr;
        RETCHK(f32_band_run(B));
        res[b].stats = B.stats;
        std::memcpy(res[b].hist, B.final_hist, sizeof(res[b].hist));
        if (resized) RETCHK(resize_pad_dev(w, lv, cols, rows, lpitch_b, target_size, 1, pad, ctx->resized[b].p, opitch, &res[b].m));
        if (!resized && out.u8_dev[b])
            HIPCHK(w, hipMemcpy2DAsync(out.u8_dev[b], out.u8_dev_pitch, lv, lpitch_b, cols, rows, hipMemcpyDeviceToDevice, w->stream));
        return SARPRO_HIP_OK;
    };
    // Device-resident bands: the second band runs on the context's twin (own stream, workspaces and mailbox) from a helper thread
    // while this thread runs the first -- at the reference's usual 2048^2 a band is a chain of short kernels and two host turns,
    // and two of them side by side take little longer than one.  (A timing context keeps one stream: its kernel table is per
    // context.  SARPRO_HIP_NO_BAND_TWIN=1: one band after the other, as host-resident bands go.)
    bool twin = src.dev[0] && src.dev[1] && !ctx->timing && !ctx->f32_stripe_open && !ctx->attrs.on(A_NO_BAND_TWIN);
    if (twin && !ctx->twin) { // first use.  No twin (or no thread) is not an error: one band after the other gives the same raster
        sarpro_hip_ctx *t = nullptr;
        BandWorker *w = nullptr;
        if (sarpro_hip_ctx_create(ctx->device, ctx->flags & ~(unsigned)SARPRO_HIP_CTX_TIMING, &t) == SARPRO_HIP_OK) {
            t->attrs = ctx->attrs; // the twin follows its parent's switches (sarpro_hip_ctx_set_attr keeps it so)
            try {
                w = new BandWorker();
                w->start();
            } catch (...) { // std::bad_alloc, std::system_error (no thread to be had): nothing may unwind across the C ABI
                delete w;
                w = nullptr;
            }
        }
        if (t && w) { ctx->twin = t; ctx->band_worker = w; }
        else { if (t) sarpro_hip_ctx_destroy(t); twin = false; }
    }
    if (twin) {
        sarpro_hip_ctx *t = ctx->twin;
        for (int b = 0; b < 2; ++b) {
            if (!resized) HIPCHK(ctx, ctx->levels[b].reserve(r1 * lpitch_b));
            else HIPCHK(ctx, ctx->resized[b].reserve(std::max<size_t>(fr, 1) * opitch));
        }
        if (resized) { HIPCHK(ctx, ctx->stage_out[0].reserve(r1 * lpitch_b)); HIPCHK(ctx, t->stage_out[0].reserve(r1 * lpitch_b)); }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (the bands may come from work still queued on this context's stream)
        ctx->band_worker->submit([&, t]() -> int {
            if (hipSetDevice(t->device) != hipSuccess) return SARPRO_HIP_ERR_HIP;
            int rc = one_band(t, 1, src.dev[1], src.dev_pitch);
            if (rc == SARPRO_HIP_OK && hipStreamSynchronize(t->stream) != hipSuccess) rc = SARPRO_HIP_ERR_HIP;
            return rc;
        });
        const int rc0 = one_band(ctx, 0, src.dev[0], src.dev_pitch);
        const int rc1 = ctx->band_worker->wait(); // (always: the job holds references to this frame)
        if (rc0 != SARPRO_HIP_OK) return rc0;
        if (rc1 != SARPRO_HIP_OK) return fail(ctx, rc1, t->err.empty() ? "dual-pol f32: the second band failed" : t->err.c_str());
    } else {
        for (int b = 0; b < 2; ++b) {
            const float *d_in = nullptr;
            size_t pitch = 0;
            if (src.dev[b]) { d_in = src.dev[b]; pitch = src.dev_pitch; }
            else { RETCHK(stage_in_2d(ctx, ctx->stage_in[0], src.host[b], rows, cols, 4, &pitch)); d_in = ctx->stage_in[0].as<float>(); }
            if (!resized) HIPCHK(ctx, ctx->levels[b].reserve(r1 * lpitch_b));
            else { HIPCHK(ctx, ctx->stage_out[0].reserve(r1 * lpitch_b)); HIPCHK(ctx, ctx->resized[b].reserve(std::max<size_t>(fr, 1) * opitch)); }
            RETCHK(one_band(ctx, b, d_in, pitch));
            if (!resized && out.u8_host[b]) RETCHK(fetch_out_2d(ctx, out.u8_host[b], ctx->levels[b].p, lpitch_b, cols, rows));
        }
    }
    for (int b = 0; b < 2; ++b) {
        if (stats_out) stats_out[b] = res[b].stats;
        for (int i = 0; i < 256; ++i) combined[i] += res[b].hist[i];
    }
    if (resized) m = res[0].m;
    if (meta) *meta = m;
    if (resized) { // composition on the resized, padded bands (compacted: the flat entry point applies), as the u16 flow does
        if (!fc || !fr) return SARPRO_HIP_OK;
        HIPCHK(ctx, ctx->stage_out[1].reserve(fc * fr));
        HIPCHK(ctx, ctx->stage_out[2].reserve(fc * fr));
        HIPCHK(ctx, ctx->stage_out[0].reserve(std::max(fc * fr * 3, r1 * round_up(std::max<size_t>(cols, 1), 64))));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_out[1].p, fc, ctx->resized[0].p, opitch, fc, fr, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpy2DAsync(ctx->stage_out[2].p, fc, ctx->resized[1].p, opitch, fc, fr, hipMemcpyDeviceToDevice, ctx->stream));
        RETCHK(sarpro_hip_synrgb_u8_dev(ctx, mode, strategy, ctx->stage_out[1].as<uint8_t>(), ctx->stage_out[2].as<uint8_t>(), fc * fr,
                                        ctx->stage_out[0].as<uint8_t>()));
        if (out.rgb_dev) HIPCHK(ctx, hipMemcpyAsync(out.rgb_dev, ctx->stage_out[0].p, fc * fr * 3, hipMemcpyDeviceToDevice, ctx->stream));
        else HIPCHK(ctx, hipMemcpyAsync(out.rgb_host, ctx->stage_out[0].p, fc * fr * 3, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return SARPRO_HIP_OK;
    }
    const size_t lpitch = round_up(std::max<size_t>(cols, 1), 64);
    std::vector<uint8_t> luts(66048), tables(66048);
    int fwc = -1;
    if (strategy == SARPRO_STRATEGY_TAMED || strategy == SARPRO_STRATEGY_CLAHE) { // synthetic_rgb.rs:188-194
        fwc = synrgb_floor_from_hist(combined, (uint64_t)rows * cols);
        synrgb_luts_suppressed(fwc, luts.data());
    } else {
        synrgb_luts_default(luts.data());
    }
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    fold_compose_tables(luts.data(), fwc, ident, ident, tables.data());
    HIPCHK(ctx, ctx->tables.reserve(66048 + 512));
    if (mail_enabled(ctx) && mail_open(ctx) == SARPRO_HIP_OK) { // the tables through the mailbox's big-payload area (both bands' turns are over): a kernel, not a copy command
        uint8_t *tstage = ctx->mailbox.as<uint8_t>() + kMailBigOff;
        std::memcpy(tstage, tables.data(), 66048);
        Prep pr;
        pr.upload(tstage, ctx->tables.p, 66048);
        RETCHK(prep_run(ctx, pr));
    } else {
        uint8_t *tstage = ctx->h_upload.as<uint8_t>();
        std::memcpy(tstage, tables.data(), 66048);
        HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p, tstage, 66048, hipMemcpyHostToDevice, ctx->stream));
    }
    ComposeArgs c{};
    c.b1 = ctx->levels[0].as<uint8_t>(); c.b2 = ctx->levels[1].as<uint8_t>(); c.in_pitch = lpitch;
    c.rows = (uint32_t)rows; c.cols = (uint32_t)cols;
    c.tables = ctx->tables.as<uint8_t>();
    if (out.rgb_dev) { c.rgb = out.rgb_dev; c.rgb_pitch_px = out.rgb_dev_pitch_px; }
    else { HIPCHK(ctx, ctx->stage_out[0].reserve(r1 * lpitch * 3)); c.rgb = ctx->stage_out[0].as<uint8_t>(); c.rgb_pitch_px = lpitch; }
    const int cvec = (c.rgb_pitch_px % 16 == 0 && (reinterpret_cast<uintptr_t>(c.rgb) & 15) == 0) ? 16 : 1;
    {
        KernelTimer t(ctx, "compose_u8");
        HIPCHK(ctx, launch_compose_u8(c, cvec, ctx->stream));
    }
    if (out.rgb_dev) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return SARPRO_HIP_OK; }
    return fetch_out_2d(ctx, out.rgb_host, ctx->stage_out[0].p, lpitch * 3, cols * 3, rows);
}

extern "C" int sarpro_hip_dualpol_synrgb_f32(sarpro_hip_ctx *ctx, const float *band1, const float *band2, size_t rows,
                                             size_t cols, int strategy, int mode, uint8_t *rgb_out, uint8_t *u8_band1,
                                             uint8_t *u8_band2, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!band1 || !band2 || !rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    const DualF32Src src{{band1, band2}, {nullptr, nullptr}, 0};
    const DualF32Out out{rgb_out, nullptr, 0, {u8_band1, u8_band2}, {nullptr, nullptr}, 0};
    return dualpol_f32_impl(ctx, src, rows, cols, strategy, mode, 0u, false, 0, 0, out, stats_out, nullptr);
}

extern "C" int sarpro_hip_dualpol_synrgb_f32_dev(sarpro_hip_ctx *ctx, const float *d_band1, const float *d_band2, size_t rows, size_t cols,
                                                 size_t in_pitch, int strategy, int mode, unsigned flags, uint8_t *d_rgb, size_t rgb_pitch_px,
                                                 uint8_t *d_u8_band1, uint8_t *d_u8_band2, size_t u8_pitch, sarpro_hip_stats *stats_out) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!d_band1 || !d_band2 || !d_rgb)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (in_pitch < cols || rgb_pitch_px < cols || ((d_u8_band1 || d_u8_band2) && u8_pitch < cols)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    const DualF32Src src{{nullptr, nullptr}, {d_band1, d_band2}, in_pitch};
    const DualF32Out out{nullptr, d_rgb, rgb_pitch_px, {nullptr, nullptr}, {d_u8_band1, d_u8_band2}, u8_pitch};
    return dualpol_f32_impl(ctx, src, rows, cols, strategy, mode, flags, false, 0, 0, out, stats_out, nullptr);
}

extern "C" int sarpro_hip_dualpol_synrgb_resized_f32(sarpro_hip_ctx *ctx, const float *band1, const float *band2, size_t rows, size_t cols,
                                                     int strategy, int mode, unsigned flags, size_t target_size, int pad, uint8_t *rgb_out,
                                                     sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!band1 || !band2 || !rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    const DualF32Src src{{band1, band2}, {nullptr, nullptr}, 0};
    const DualF32Out out{rgb_out, nullptr, 0, {nullptr, nullptr}, {nullptr, nullptr}, 0};
    return dualpol_f32_impl(ctx, src, rows, cols, strategy, mode, flags, true, target_size, pad, out, nullptr, meta);
}

extern "C" int sarpro_hip_dualpol_synrgb_resized_f32_dev(sarpro_hip_ctx *ctx, const float *d_band1, const float *d_band2, size_t rows,
                                                         size_t cols, size_t in_pitch, int strategy, int mode, unsigned flags,
                                                         size_t target_size, int pad, uint8_t *d_rgb_out, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (rows * cols && (!d_band1 || !d_band2 || !d_rgb_out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    if (in_pitch < cols) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "pitch < cols");
    const DualF32Src src{{nullptr, nullptr}, {d_band1, d_band2}, in_pitch};
    const DualF32Out out{nullptr, d_rgb_out, 0, {nullptr, nullptr}, {nullptr, nullptr}, 0};
    return dualpol_f32_impl(ctx, src, rows, cols, strategy, mode, flags, true, target_size, pad, out, nullptr, meta);
}

// ---------------------------------------------------------------------------------------
// host half of the f32 flavour (no GPU): exported for the CPU test-suite
// ---------------------------------------------------------------------------------------
extern "C" int sarpro_hip_host_stats_from_bins4096(uint64_t valid_count, double min_db, double max_db, double mean_db,
                                                   double std_db, const uint64_t hist4096[4096], sarpro_hip_stats *out) {
    if (!hist4096 || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    return stats_from_bins4096(valid_count, min_db, max_db, mean_db, std_db, hist4096, out);
}

extern "C" float sarpro_hip_host_f32_valid_threshold(void) { return valid_threshold_f32(); }

extern "C" int sarpro_hip_host_f32_bin4096_thresholds(double min_db, double max_db, float thr_out[4096]) {
    if (!thr_out || !(max_db > min_db)) return SARPRO_HIP_ERR_INVALID_ARG;
    build_bin4096_thresholds(min_db, max_db, thr_out);
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_host_f32_level_thresholds(const sarpro_hip_stats *stats, int bit_depth, float *thr_out) {
    if (!stats || !thr_out || (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16)) return SARPRO_HIP_ERR_INVALID_ARG;
    build_level_thresholds(*stats, bit_depth == SARPRO_BITDEPTH_U8 ? 255 : 65535, thr_out);
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_host_f32_clahe_bin_thresholds(const sarpro_hip_stats *stats, float thr_out[256]) {
    if (!stats || !thr_out) return SARPRO_HIP_ERR_INVALID_ARG;
    build_clahe_bin_thresholds(*stats, thr_out);
    return SARPRO_HIP_OK;
}

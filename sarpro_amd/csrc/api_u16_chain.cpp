// api_u16_chain.cpp -- the device-resident chains of the u16 flavour: the host only enqueues (CLAHE with its speculative fused route,
// the six percentile strategies), and job_run_all, which picks a scene's route.
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
// Device-resident chain: CLAHE, u8 output, vector layout, whole scene on this GPU.  The host only
// enqueues: statistics, CLAHE bins, CDFs, rescale and compose tables are computed by small kernels
// (chain_kernels.hip), so there is ONE stream synchronisation per scene, at the end.
// ---------------------------------------------------------------------------------------
constexpr size_t kChainOffDb = 0, kChainOffSupp = 65536 * 8, kChainOffBlue = kChainOffSupp + 21504;
constexpr size_t kChainOffBlueDef = kChainOffBlue + 65536, kChainOffDefRg = kChainOffBlueDef + 65536;
constexpr size_t kChainOffGamma = kChainOffDefRg + 512, kChainOffBluePQ = kChainOffGamma + 3 * 256 * 8, kChainConstBytes = kChainOffBluePQ + 512 * 4;
constexpr size_t kTablesOffPQ = 66048 + 512, kTablesBytes = kTablesOffPQ + 2 * 256 * 4; // compose tables | per-band maps | Pv[256] f32 | Qv[256] f32 (the blue factors by LEVEL)
constexpr size_t kStateOffResc = 2 * sizeof(ChainBandState), kStateOffIdent = kStateOffResc + 512,
                 kStateOffFloor = kStateOffIdent + 16, kStateBytes = kStateOffFloor + 16;

extern "C" int sarpro_hip_ctx_chain_report(sarpro_hip_ctx *ctx, sarpro_hip_chain_report *out) {
    if (!ctx || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    out->floor_with_cushion = -1;
    if (!ctx->chain_state.p || !ctx->last_final_hist) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "no CLAHE chain has run on this context");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint8_t *state = ctx->chain_state.as<uint8_t>();
    HIPCHK(ctx, hipMemcpy(out->rescale, state + kStateOffResc, 512, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out->identity, state + kStateOffIdent, 2, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(&out->floor_with_cushion, state + kStateOffFloor, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(out->level_hist, ctx->last_final_hist, sizeof(out->level_hist), hipMemcpyDeviceToHost));
    return SARPRO_HIP_OK;
}


static int chain_prepare(sarpro_hip_ctx *ctx) {
    if (ctx->chain_ready) return SARPRO_HIP_OK;
    HIPCHK(ctx, ctx->chain_consts.reserve(kChainConstBytes));
    HIPCHK(ctx, ctx->chain_state.reserve(kStateBytes));
    uint8_t *d = ctx->chain_consts.as<uint8_t>();
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffDb, db_table_u16(), 65536 * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffSupp, synrgb_supp_rg_tables(), 41 * 512, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffBlue, synrgb_blue_pair_supp(), 65536, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffBlueDef, synrgb_blue_pair_default(), 65536, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint8_t> dflt(66048);
    synrgb_luts_default(dflt.data());
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffDefRg, dflt.data(), 512, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d + kChainOffGamma, gamma_level_thresholds_u8(), 3 * 256 * 8, hipMemcpyHostToDevice, ctx->stream));
    if (const float *pq = synrgb_blue_factors_supp()) { // the suppressed blue as a product of two factors (verified against the pair table for all pairs)
        HIPCHK(ctx, hipMemcpyAsync(d + kChainOffBluePQ, pq, 512 * 4, hipMemcpyHostToDevice, ctx->stream));
        ctx->blue_factors_ok = true;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->chain_ready = true;
    return SARPRO_HIP_OK;
}

bool chain_eligible(const U16Job &J) {
    if (J.ctx->attrs.on(A_NO_CHAIN)) return false;
    // u16 output (no rescale, no composition) takes the same chain up to the blend, with the exact f64 kernel
    return J.clahe() && (J.u8_out() || !J.synrgb) && J.vec && !J.tamed_force && (J.reduce || (J.row0 == 0 && J.rows_local == J.rows_total));
}

static int chain_stats_scratch(sarpro_hip_ctx *ctx, ChainStatsArgs *sa) {
    const size_t part_bytes = (sizeof(ChainStatsPartial) * kChainStatsParts * kMaxBands + 255) & ~(size_t)255;
    HIPCHK(ctx, ctx->chain_scratch.reserve(part_bytes + sizeof(uint64_t) * 4096 * kMaxBands));
    sa->partials = ctx->chain_scratch.as<ChainStatsPartial>();
    sa->bins4096 = reinterpret_cast<unsigned long long *>(ctx->chain_scratch.as<uint8_t>() + part_bytes);
    return SARPRO_HIP_OK;
}

// row-stripe mode: merge a small integer buffer across ranks without leaving the stream
static int chain_reduce(U16Job &J, void *d_buf, size_t count_u64, const char *what) {
    if (!J.reduce) return SARPRO_HIP_OK;
    KernelTimer t(J.ctx, what);
    return comm_allreduce_sum_u64_async(J.ctx, reinterpret_cast<uint64_t *>(d_buf), count_u64);
}

// End of a device-resident chain: return at once on a stream-ordered context, else read the statistics back.
static int chain_tail(U16Job &J, sarpro_hip_stats *stats_out, ChainBandState *d_state) {
    sarpro_hip_ctx *ctx = J.ctx;
    if (ctx->async_dev && J.allow_async && !stats_out && !J.reduce) { // stream-ordered: nothing is read back, the LDS capacity keeps its value
        ctx->async_pending = ctx->timing;
        return SARPRO_HIP_OK;
    }
    ChainBandState *h_state = ctx->h_small.as<ChainBandState>();
    HIPCHK(ctx, hipMemcpyAsync(h_state, d_state, sizeof(ChainBandState) * (size_t)J.nbands, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the only synchronisation of the chain
    uint32_t hi = 0;
    for (int b = 0; b < J.nbands; ++b) {
        J.stats[b] = h_state[b].stats;
        if (stats_out) stats_out[b] = J.stats[b];
        hi = std::max(hi, h_state[b].win_hi);
    }
    // size the LDS offset table of the NEXT scene from this scene's window (speed only: a window larger
    // than the capacity is gathered from global memory, with identical results)
    ctx->chain_lut_cap = std::min<uint32_t>(16384, std::max<uint32_t>(1024, (hi + 1 + 255) / 256 * 256));
    return SARPRO_HIP_OK;
}

// Scenes below this size keep the exact partial histogram: their chain is launch-bound, the gated kernels would cost more than
// the sampled histogram saves.  SARPRO_HIP_SAMPLED_HIST_MIN_PX overrides (the tests run the speculative chain on small rasters).
constexpr size_t kSampledHistMinPx = 32u << 20;
static uint32_t spec_force_flags(const sarpro_hip_ctx *ctx) { // SPEC_FORCE = mispredict (1) | nospec (2) | predicted lowest level + 1 (4): every rare branch of the speculative chain is testable
    return (uint32_t)ctx->attrs.val(A_SPEC_FORCE, 0) & (kSpecForceMispredict | kSpecForceNoSpec | kSpecForceMinMispredict | kSpecForceNoRetry | kSpecForceMispredict2);
}

static int chain_tail(U16Job &J, sarpro_hip_stats *stats_out, ChainBandState *d_state);

struct FusedTail {
    sarpro_hip_ctx *ctx; U16Job *J; ClaheRgbArgs fa; ClaheApplyArgs a; ChainSpecState *d_spec; ChainBandState *d_state;
    unsigned long long *exact_hist; StripePlan *plan; uint8_t *d_levels[2]; size_t lvl_pitch; uint32_t rows, cols;
    unsigned long long total_px; uint8_t *d_rgb; size_t rgb_pitch_px;
    ChainPredictArgs pa; // (the prediction's arguments: its second launch, behind an undercut lowest level, takes the same)
};
// The fused pass and what is gated on its verdict (job_run_fused_rgb's second half).  T.J is null when the call is deferred (never a row stripe).
static int fused_rgb_tail(const FusedTail &T) {
    sarpro_hip_ctx *ctx = T.ctx;
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainSpecState *d_spec = T.d_spec;
    ClaheApplyArgs a = T.a;
    const bool reduce = T.J && T.J->reduce;
    {
        // resident batch (pipeline.cpp): this scene's pass behind the previous scene's pass (another lane's stream), its own completion
        // published for the next one -- the passes own whole compute units (160 KiB of LDS each), two of them at once only split the chip
        if (ctx->pipe_wait_before_fused) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_wait_before_fused, 0));
        if (ctx->pipe_record_before_fused) { HIPCHK(ctx, hipEventRecord(ctx->pipe_record_before_fused, ctx->stream)); ctx->pipe_record_before_fused = nullptr; }
        int grid = std::max(ctx->cu_count, 1);
        if (ctx->attrs.is_set(A_RGB_GRID)) grid = (int)std::min<long long>(1024, std::max<long long>(1, ctx->attrs.val(A_RGB_GRID, grid)));
        {
            KernelTimer t(ctx, "clahe_rgb_fused");
            HIPCHK(ctx, launch_clahe_rgb_fused(T.fa, grid, ctx->stream));
        }
        if (ctx->pipe_record_after_fused) {
            HIPCHK(ctx, hipEventRecord(ctx->pipe_record_after_fused, ctx->stream));
            ctx->pipe_record_after_fused = nullptr; // recorded (the batch records it itself behind a chain that never got here)
        }
    }
    if (reduce) { // the verification counts of all stripes, then the verdict every rank shares
        static_assert(offsetof(ChainSpecState, n_below_min) == offsetof(ChainSpecState, n_lt) + 16 && offsetof(ChainSpecState, below_hist) == offsetof(ChainSpecState, n_lt) + 24,
                      "the verification counts are one buffer");
        RETCHK(chain_reduce(*T.J, &d_spec->n_lt[0], kSpecCountWords, "allreduce_spec_counts"));
        HIPCHK(ctx, launch_spec_verdict(d_spec, T.d_state, ctx->stream));
    }
    {   // a refuted floor gets one second pass with the floor the first pass's counts point to (both launches return at once otherwise)
        ChainRepredictArgs ra{};
        ra.spec = d_spec; ra.resc_in = state + kStateOffResc; ra.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        ra.tables = ctx->tables.as<uint8_t>(); ra.supp_rg = consts + kChainOffSupp; ra.blue_pair_supp = consts + kChainOffBlue;
        ra.blue_pq = ctx->blue_factors_ok ? reinterpret_cast<const float *>(consts + kChainOffBluePQ) : nullptr;
        ra.blue_by_level = reinterpret_cast<float *>(ctx->tables.as<uint8_t>() + kTablesOffPQ);
        ra.stripes = reduce ? 1u : 0u;
        {
            KernelTimer t(ctx, "chain_repredict");
            HIPCHK(ctx, launch_chain_repredict(ra, ctx->stream));
            {   // an undercut lowest level whose true value the pass recorded (row stripes: the ranks' summed presence counts): the prediction again, on that level
                ChainPredictArgs p2 = T.pa;
                p2.second = 1u;
                HIPCHK(ctx, launch_chain_predict(p2, ctx->stream));
            }
        }
        ClaheRgbArgs fr = T.fa;
        fr.retry = 1u;
        int grid = std::max(ctx->cu_count, 1);
        if (ctx->attrs.is_set(A_RGB_GRID)) grid = (int)std::min<long long>(1024, std::max<long long>(1, ctx->attrs.val(A_RGB_GRID, grid)));
        {
            KernelTimer t(ctx, "clahe_rgb_fused_retry");
            HIPCHK(ctx, launch_clahe_rgb_fused_retry(fr, grid, ctx->stream));
        }
        if (reduce) {
            RETCHK(chain_reduce(*T.J, &d_spec->n_lt[0], kSpecCountWords, "allreduce_spec_counts_retry"));
            HIPCHK(ctx, launch_spec_verdict(d_spec, T.d_state, ctx->stream, 1));
        }
    }
    {   // gated on the verdict: levels of every pixel with the full histogram -> exact tables -> composition
        KernelTimer t(ctx, "spec_fallback_apply");
        a.hist_mode = 0u; a.gate = d_spec;
        a.rects = T.plan->d_apply_rects.as<Rect>();
        for (int b = 0; b < 2; ++b) a.level_hist[b] = T.exact_hist + (size_t)b * 256;
        HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)T.plan->apply_rects.size(), 2, ctx->stream));
    }
    if (reduce) RETCHK(chain_reduce(*T.J, T.exact_hist, (size_t)256 * kMaxBands, "allreduce_level_hist")); // (all zero when the fused RGB stood: the gated recount did not run)
    {
        ChainFinishArgs f{};
        f.level_hist = T.exact_hist; f.gate = d_spec;
        f.total_px = T.total_px; f.nbands = 2;
        f.resc_out = state + kStateOffResc; f.identity_out = state + kStateOffIdent;
        f.tables = ctx->tables.as<uint8_t>();
        f.supp_rg = consts + kChainOffSupp; f.blue_pair_supp = consts + kChainOffBlue;
        f.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        f.suppressed = 1;
        KernelTimer t(ctx, "chain_finish");
        HIPCHK(ctx, launch_chain_finish(f, ctx->stream));
    }
    {
        ComposeArgs c{};
        c.b1 = T.d_levels[0]; c.b2 = T.d_levels[1]; c.in_pitch = T.lvl_pitch;
        c.rgb = T.d_rgb; c.rgb_pitch_px = T.rgb_pitch_px; c.rows = T.rows; c.cols = T.cols;
        c.tables = ctx->tables.as<uint8_t>();
        c.spec = d_spec; c.speculative = 0;
        KernelTimer t(ctx, "spec_fallback_compose");
        HIPCHK(ctx, launch_compose_u8(c, 16, ctx->stream));
    }
    return SARPRO_HIP_OK;
}

// The fused CLAHE -> RGB route of job_run_chain, from the CDFs on (everything before it is shared with the other routes).
static int job_run_fused_rgb(U16Job &J, const ClaheRgbArgs &fa, uint8_t *d_rgb, size_t rgb_pitch_px, uint32_t sample_stride,
                             sarpro_hip_stats *stats_out) {
    sarpro_hip_ctx *ctx = J.ctx;
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainBandState *d_state = reinterpret_cast<ChainBandState *>(state);
    ChainSpecState *d_spec = ctx->spec_state.as<ChainSpecState>();
    const uint32_t rows = (uint32_t)J.rows_local, cols = (uint32_t)J.cols;
    RETCHK(ensure_levels(J)); // the fallback's level rasters (allocated once per shape; untouched when the fused RGB stands)
    HIPCHK(ctx, ctx->spec_dump.reserve(kSpecDumpBytes));
    unsigned long long *sample_hist = ctx->level_hist.as<unsigned long long>(), *exact_hist = sample_hist + 256 * kMaxBands * kSampleReplicas;
    ctx->last_final_hist = exact_hist;
    ClaheApplyArgs a{};
    for (int b = 0; b < 2; ++b) {
        a.in[b] = J.d_in[b];
        a.out[b] = J.d_levels[b];
        a.cdfs[b] = fa.cdfs[b];
        a.binlut[b] = fa.binlut[b];
        a.level_hist[b] = sample_hist + (size_t)b * 256;
    }
    a.in_pitch = J.in_pitch; a.out_pitch = J.lvl_pitch;
    a.row_w = fa.row_w; a.col_w = fa.col_w; a.row_off = fa.row_off;
    a.max_val = 255.0; a.dev_state = d_state; a.lut_cap = ctx->chain_lut_cap;
    a.dump = ctx->spec_dump.as<uint8_t>();
    a.sample_stride = sample_stride; a.sample_phase = sample_stride / 2; a.sample_valid = d_spec->sample_valid_rep;
    {   // the sampled rows of both bands through the blend: level histogram + valid counts, nothing stored
        a.hist_mode = 3u;
        a.rects = J.plan->d_sample_rects.as<Rect>();
        KernelTimer t(ctx, "clahe_sample");
        HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)J.plan->sample_rects.size(), 2, ctx->stream));
    }
    // row stripes: the sample of the SCENE (every replica of the sampled histogram, the valid counts beside them)
    RETCHK(chain_reduce(J, sample_hist, (size_t)256 * kMaxBands * kSampleReplicas, "allreduce_sample_hist"));
    RETCHK(chain_reduce(J, d_spec->sample_valid_rep, (size_t)kSampleReplicas * 2, "allreduce_sample_valid"));
    ChainPredictArgs pa_first{};
    {
        ChainPredictArgs pa{};
        pa.sample_hist = sample_hist; pa.exact_hist = exact_hist; pa.spec = d_spec; pa.state = d_state;
        pa.total_px = (unsigned long long)J.rows_total * J.cols;
        pa.resc_out = state + kStateOffResc; pa.identity_out = state + kStateOffIdent;
        pa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        pa.tables = ctx->tables.as<uint8_t>();
        pa.supp_rg = consts + kChainOffSupp; pa.blue_pair_supp = consts + kChainOffBlue;
        pa.force = spec_force_flags(ctx);
        pa.allow_rescaled = ctx->attrs.on(A_NO_SPEC_RESCALE) ? 0u : 1u; // the fused pass verifies a predicted lowest level
        pa.blue_pq = ctx->blue_factors_ok ? reinterpret_cast<const float *>(consts + kChainOffBluePQ) : nullptr;
        pa.blue_by_level = reinterpret_cast<float *>(ctx->tables.as<uint8_t>() + kTablesOffPQ);
        pa_first = pa;
        KernelTimer t(ctx, "chain_predict");
        HIPCHK(ctx, launch_chain_predict(pa, ctx->stream));
    }
    // everything from the fused pass on: at once, or -- resident batch with PIPE_ORDER = 3 -- when the batch driver says so (the next
    // scene's histogram sweep is enqueued on another lane FIRST, so that this pass can wait for it: an event must be recorded
    // before a stream can be made to wait for it)
    FusedTail T{};
    T.ctx = ctx; T.J = &J; T.fa = fa; T.a = a; T.d_spec = d_spec; T.d_state = d_state; T.exact_hist = exact_hist;
    T.plan = J.plan; T.d_levels[0] = J.d_levels[0]; T.d_levels[1] = J.d_levels[1]; T.lvl_pitch = J.lvl_pitch;
    T.rows = rows; T.cols = cols; T.total_px = (unsigned long long)J.rows_total * J.cols; T.d_rgb = d_rgb; T.rgb_pitch_px = rgb_pitch_px;
    T.pa = pa_first;
    if (ctx->pipe_defer && !J.reduce && ctx->async_dev && J.allow_async && !stats_out) {
        T.J = nullptr; // (the job object is the caller's: gone when the tail runs)
        ctx->pipe_deferred = [T]() { return fused_rgb_tail(T); };
        ctx->async_pending = ctx->timing; // what chain_tail does on a stream-ordered context
        return SARPRO_HIP_OK;
    }
    RETCHK(fused_rgb_tail(T));
    return chain_tail(J, stats_out, d_state);
}

int job_run_chain(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                         sarpro_hip_stats *stats_out) {
    sarpro_hip_ctx *ctx = J.ctx;
    RETCHK(chain_prepare(ctx));
    const uint32_t rows = (uint32_t)J.rows_local, cols = (uint32_t)J.cols;
    HIPCHK(ctx, ctx->luts.reserve(2 * 131072));
    HIPCHK(ctx, ctx->tile_bins.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands));
    HIPCHK(ctx, ctx->cdfs.reserve(sizeof(double) * 64 * 256 * kMaxBands));
    // [kSampleReplicas][2][256]: the apply / sampling pass's histogram (replica 0 alone unless it is sampled); then [2][256]: the gated recount
    HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands * (kSampleReplicas + 1)));
    HIPCHK(ctx, ctx->tables.reserve(kTablesBytes));
    HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainBandState *d_state = reinterpret_cast<ChainBandState *>(state);
    // Dual-pol u8 scene on one device: the level histogram is counted on sampled rows only and the composition is speculative
    // (k_chain_predict); SARPRO_HIP_NO_SAMPLED_HIST=1 keeps the partial histogram of every row and the unconditional tail.
    const bool exact_only = ctx->attrs.on(A_NO_SPEC); // cross-check: every pixel through the exact f64 blend
    const size_t rgb_pitch_ok = rgb_pitch_px % 16 == 0 && ptr_aligned16(d_rgb);
    // (a row stripe takes the speculative route in its fused form only -- the sampled histogram, the valid counts and the pass's
    // verification counts are summed over the ranks, so every rank proves, predicts and decides the same; decided below from
    // what all ranks share: an empty stripe still joins every reduction)
    const bool whole = J.row0 == 0 && J.rows_local == J.rows_total;
    const bool fused_wanted = !d_out[0] && !d_out[1] && !ctx->attrs.on(A_NO_FUSED_RGB) && J.in_pitch % 8 == 0;
    bool sampled = J.synrgb && J.nbands == 2 && J.u8_out() && !exact_only && !ctx->attrs.on(A_FULL_LEVEL_HIST) &&
                   !ctx->attrs.on(A_NO_SAMPLED_HIST) && rgb_pitch_px % 16 == 0 && (J.reduce ? fused_wanted : (d_rgb && rgb_pitch_ok && whole));
    // every 17th row; every 33rd on scenes of 12000 rows and more (606 sampled rows at 20000).  The sample pass is 5 % of the fused pass's work
    // and, in a resident batch, runs beside another lane's histogram sweep: nine-scene cycle on three lanes 0.967 -> 0.949 ms per scene with
    // 33, 0.944 with 65 (all nine accepted either way); the estimate's error on these scenes is a bias of the row phase against the scene's
    // structure (1-2.5 % of a level's population at 9, 17, 33 and 65 alike: profiles/r3/spec_accuracy.txt), its random part grows with sqrt(stride)
    uint32_t sample_stride = J.rows_total >= 12000 ? 33 : 17;
    if (sampled) {
        size_t min_px = kSampledHistMinPx;
        if (ctx->attrs.is_set(A_SAMPLED_HIST_MIN_PX)) min_px = (size_t)std::max<long long>(0, ctx->attrs.val(A_SAMPLED_HIST_MIN_PX, 0));
        if (ctx->attrs.is_set(A_SAMPLE_STRIDE)) sample_stride = (uint32_t)std::max<long long>(5, ctx->attrs.val(A_SAMPLE_STRIDE, 0));
        if ((size_t)J.rows_total * J.cols < min_px) sampled = false;
    }
    ChainSpecState *d_spec = nullptr;
    ctx->spec_ran = sampled;
    if (sampled) {
        HIPCHK(ctx, ctx->spec_state.reserve(sizeof(ChainSpecState)));
        d_spec = ctx->spec_state.as<ChainSpecState>();
    }

    RETCHK(job_phase1(J)); // per-tile DN histograms -> ctx->ghist
    RETCHK(chain_reduce(J, ctx->ghist.p, 65536 * (size_t)J.nbands, "allreduce_dn_hist"));
    {
        ChainStatsArgs sa{};
        sa.ghist = ctx->ghist.as<unsigned long long>();
        sa.db = reinterpret_cast<const double *>(consts + kChainOffDb);
        sa.state = d_state;
        sa.binlut = ctx->luts.as<uint8_t>();
        sa.binlut_stride = 131072;
        sa.level_hist = ctx->level_hist.as<unsigned long long>(); // cleared here for the apply kernel (one fill kernel less)
        sa.sample_valid = d_spec ? d_spec->sample_valid_rep : nullptr;
        KernelTimer t(ctx, "chain_stats");
        RETCHK(chain_stats_scratch(ctx, &sa));
        HIPCHK(ctx, launch_chain_stats(sa, J.nbands, ctx->stream));
    }
    {
        TileBinHistArgs ta{};
        for (int b = 0; b < J.nbands; ++b) {
            ta.tile_hist[b] = tile_hist_of(ctx, b, kTiles * kTiles);
            ta.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
            ta.out[b] = ctx->tile_bins.as<unsigned long long>() + (size_t)b * 64 * 256;
        }
        ta.clear = 1u; // the last reader of the tile histograms
        if (!J.reduce && !ctx->attrs.on(A_SEPARATE_CDFS)) { // one device: the CDFs in the same launch
            for (int b = 0; b < J.nbands; ++b) ta.cdfs_out[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
            ta.rows = (uint32_t)J.rows_total; ta.cols = cols;
        }
        KernelTimer t(ctx, "tile_bin_hist");
        HIPCHK(ctx, launch_tile_bin_hist(ta, kTiles * kTiles, J.nbands, ctx->stream));
        mark_tile_hist_clean(J);
    }
    RETCHK(chain_reduce(J, ctx->tile_bins.p, 64 * 256 * (size_t)J.nbands, "allreduce_tile_hists"));
    if (J.reduce || ctx->attrs.on(A_SEPARATE_CDFS)) {
        KernelTimer t(ctx, "chain_cdfs");
        HIPCHK(ctx, launch_chain_cdfs(ctx->tile_bins.as<unsigned long long>(), ctx->cdfs.as<double>(), (uint32_t)J.rows_total, cols,
                                      J.nbands, ctx->stream));
    }
    // Whole dual-pol u8 scene, RGB only: the fused pass (kernels.hip 6a) -- sample-only pass -> identity proof + predicted floor +
    // tables -> ONE sweep DN, DN -> RGB that verifies the floor; refuted (or unproven, or windows beyond the pass's LDS pool), the
    // gated apply -> finish -> compose kernels below produce the raster.  SARPRO_HIP_NO_FUSED_RGB=1: the apply + compose route.
    if (sampled && fused_wanted && (J.reduce || !J.plan->rgb_rects.empty())) {
        ClaheRgbArgs fa{};
        for (int b = 0; b < 2; ++b) {
            fa.in[b] = J.d_in[b];
            fa.cdfs[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
            fa.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        }
        fa.in_pitch = J.in_pitch; fa.rgb = d_rgb; fa.rgb_pitch_px = rgb_pitch_px;
        fa.rects = J.plan->d_rgb_rects.as<Rect>(); fa.nrects = (int)J.plan->rgb_rects.size();
        fa.row_w = J.plan->d_row_w.as<RowWeight>(); fa.col_w = J.plan->d_col_w.as<RowWeight>(); fa.row_off = (int32_t)J.row0;
        fa.dev_state = d_state; fa.spec = d_spec; fa.tables = ctx->tables.as<uint8_t>();
        fa.blue_by_level = ctx->blue_factors_ok ? reinterpret_cast<const float *>(ctx->tables.as<uint8_t>() + kTablesOffPQ) : nullptr;
        fa.sat_ok = J.plan->sat_ok ? 1u : 0u; fa.sat_col = J.plan->d_sat_col.as<uint8_t>(); fa.sat_row = J.plan->d_sat_row.as<uint8_t>() + J.row0; // (the table is indexed by the scene's row, the kernel by the stripe's)
        fa.sat_cols = (uint32_t)(round_up(J.cols, 64) + 64);
        // the sample-only pass costs ~0.025 ms + (apply pass) / stride: 0.056 ms at 17, 0.033 at 33; the wider stride's larger sigma (x 1.4: ~2.4 %
        // of scenes refuted instead of ~1.7 %, 1 ms each) costs 0.007 ms in expectation
        const uint32_t fused_stride = ctx->attrs.is_set(A_SAMPLE_STRIDE) ? sample_stride : 33u;
        fa.no_verdict = J.reduce ? 1u : 0u;
        if (clahe_rgb_fused_supported(fa)) return job_run_fused_rgb(J, fa, d_rgb, rgb_pitch_px, fused_stride, stats_out);
        if (J.reduce) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "row stripe: the rasters of every rank must be 16-byte aligned (pitches % 8 / % 16)");
    }
    // apply: levels into the internal rasters (dual-pol) or straight into the caller's raster (single band)
    const bool direct = !J.synrgb;
    if (!direct) RETCHK(ensure_levels(J));
    ClaheApplyArgs a{};
    for (int b = 0; b < J.nbands; ++b) {
        a.in[b] = J.d_in[b];
        a.out[b] = direct ? d_out[b] : (void *)J.d_levels[b];
        a.cdfs[b] = ctx->cdfs.as<double>() + (size_t)b * 64 * 256;
        a.binlut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072;
        a.level_hist[b] = ctx->level_hist.as<unsigned long long>() + (size_t)b * 256;
    }
    a.in_pitch = J.in_pitch;
    a.out_pitch = direct ? out_pitch : J.lvl_pitch;
    a.rects = J.plan->d_apply_rects.as<Rect>();
    a.row_w = J.plan->d_row_w.as<RowWeight>();
    a.col_w = J.plan->d_col_w.as<RowWeight>();
    a.row_off = (int32_t)J.row0;
    const bool u16o = !J.u8_out();
    a.max_val = u16o ? 65535.0 : 255.0;
    a.dev_state = d_state;
    a.lut_cap = ctx->chain_lut_cap;
    if (a.out_pitch % 8 != 0 || !ptr_aligned16(a.out[0]) || (J.nbands > 1 && !ptr_aligned16(a.out[1])))
        return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "output raster must be 16-byte aligned with pitch % 8 == 0 when the input is");
    if (u16o) { // autoscale.rs:600-606 at max_val 65535: no u8 rescale, nothing downstream needs a level histogram
        for (int b = 0; b < J.nbands; ++b) a.level_hist[b] = nullptr;
        {
            KernelTimer t(ctx, "clahe_apply_u16");
            if (!ctx->attrs.on(A_NO_U16_CF) && J.nbands <= kMaxBands && J.plan->u16_nwg[J.nbands - 1] > 0 &&
                clahe_apply_u16_cf_supported(a, J.nbands, J.plan->u16_item_rows)) { // the conflict-free form: one persistent workgroup per share
                a.rects = J.plan->d_u16_items[J.nbands - 1].as<Rect>();
                HIPCHK(ctx, launch_clahe_apply_u16_cf(a, J.plan->d_u16_first[J.nbands - 1].as<int32_t>(), J.plan->u16_nwg[J.nbands - 1], J.nbands, ctx->stream));
            } else {
                HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_rects.size(), J.nbands, true, true, ctx->stream));
            }
        }
        ChainBandState *h_state = ctx->h_small.as<ChainBandState>();
        HIPCHK(ctx, hipMemcpyAsync(h_state, d_state, sizeof(ChainBandState) * (size_t)J.nbands, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t hi = 0;
        for (int b = 0; b < J.nbands; ++b) {
            J.stats[b] = h_state[b].stats;
            if (stats_out) stats_out[b] = J.stats[b];
            hi = std::max(hi, h_state[b].win_hi);
        }
        ctx->chain_lut_cap = std::min<uint32_t>(16384, std::max<uint32_t>(1024, (hi + 1 + 255) / 256 * 256));
        return SARPRO_HIP_OK;
    }
    // (the level histogram was cleared by the statistics kernels)
    // whole scene on this device: levels >= 64 are only counted in bulk (chain_kernels.hip k_level_hist_guard); a row
    // stripe keeps the full histogram, which is what the ranks sum
    a.hist_mode = sampled ? 2u : (!J.reduce && !exact_only && !ctx->attrs.on(A_FULL_LEVEL_HIST)) ? 1u : 0u;
    if (sampled) {
        a.sample_stride = sample_stride;
        a.sample_phase = sample_stride / 2; // mid-phase: the row weights of the sampled rows average to those of all rows
        a.sample_valid = d_spec->sample_valid_rep;
    }
    if (exact_only) {
        KernelTimer t(ctx, "clahe_apply_u16");
        HIPCHK(ctx, launch_clahe_apply_u16(a, (int)J.plan->apply_rects.size(), J.nbands, true, false, ctx->stream));
    } else {
        HIPCHK(ctx, ctx->spec_dump.reserve(kSpecDumpBytes));
        a.dump = ctx->spec_dump.as<uint8_t>();
        KernelTimer t(ctx, "clahe_apply_u8_spec");
        HIPCHK(ctx, launch_clahe_apply_u8_spec(a, (int)J.plan->apply_rects.size(), J.nbands, ctx->stream));
    }
    if (a.hist_mode == 1) {
        KernelTimer t(ctx, "level_hist_guard");
        HIPCHK(ctx, ctx->hist_flags.reserve(sizeof(uint32_t) * kMaxBands));
        HIPCHK(ctx, launch_level_hist_guard(ctx->level_hist.as<unsigned long long>(), (unsigned long long)J.rows_total * J.cols, J.nbands,
                                            ctx->hist_flags.as<uint32_t>(), ctx->stream));
        LevelRecountArgs ra{};
        for (int b = 0; b < J.nbands; ++b) ra.levels[b] = reinterpret_cast<const uint8_t *>(a.out[b]);
        ra.pitch = a.out_pitch; ra.rows = (uint32_t)J.rows_local; ra.cols = cols;
        ra.level_hist = ctx->level_hist.as<unsigned long long>(); ra.flags = ctx->hist_flags.as<uint32_t>();
        HIPCHK(ctx, launch_level_hist_if_flagged(ra, J.nbands, ctx->stream));
    }
    RETCHK(chain_reduce(J, ctx->level_hist.p, 256 * kMaxBands, "allreduce_level_hist"));
    ComposeArgs c{};
    int cvec = 1;
    if (J.synrgb) {
        c.b1 = J.d_levels[0]; c.b2 = J.d_levels[1]; c.in_pitch = J.lvl_pitch;
        c.rgb = d_rgb; c.rgb_pitch_px = rgb_pitch_px; c.rows = rows; c.cols = cols;
        c.tables = ctx->tables.as<uint8_t>();
        cvec = (c.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(c.b1) && ptr_aligned16(c.b2) && ptr_aligned16(d_rgb)) ? 16 : 1;
    }
    unsigned long long *final_hist = ctx->level_hist.as<unsigned long long>();
    if (sampled) { // identity proof + predicted floor + tables -> speculative composition (counts and verdict) -> gated exact recount
        if (cvec != 16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "sampled level histogram without the vector compose pass");
        ChainPredictArgs pa{};
        pa.sample_hist = ctx->level_hist.as<unsigned long long>();
        pa.exact_hist = final_hist = ctx->level_hist.as<unsigned long long>() + 256 * kMaxBands * kSampleReplicas;
        pa.spec = d_spec;
        pa.state = d_state;
        pa.total_px = (unsigned long long)J.rows_total * J.cols;
        pa.resc_out = state + kStateOffResc;
        pa.identity_out = state + kStateOffIdent;
        pa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        pa.tables = ctx->tables.as<uint8_t>();
        pa.supp_rg = consts + kChainOffSupp;
        pa.blue_pair_supp = consts + kChainOffBlue;
        pa.force = spec_force_flags(ctx);
        {
            KernelTimer t(ctx, "chain_predict");
            HIPCHK(ctx, launch_chain_predict(pa, ctx->stream));
        }
        c.spec = d_spec;
        c.speculative = 1;
        {
            KernelTimer t(ctx, "compose_u8");
            HIPCHK(ctx, launch_compose_u8(c, 16, ctx->stream));
        }
        c.speculative = 0; // the composition below is the gated fallback
        KernelTimer t(ctx, "spec_fallback_recount");
        LevelRecountArgs ra{};
        for (int b = 0; b < J.nbands; ++b) ra.levels[b] = reinterpret_cast<const uint8_t *>(a.out[b]);
        ra.pitch = a.out_pitch; ra.rows = (uint32_t)J.rows_local; ra.cols = cols;
        ra.level_hist = final_hist; ra.gate = d_spec;
        HIPCHK(ctx, launch_level_hist_if_flagged(ra, J.nbands, ctx->stream));
    }
    {
        ChainFinishArgs fa{};
        fa.level_hist = final_hist;
        ctx->last_final_hist = final_hist;
        fa.gate = d_spec;
        fa.total_px = (unsigned long long)J.rows_total * J.cols;
        fa.nbands = J.nbands;
        fa.resc_out = state + kStateOffResc;
        fa.identity_out = state + kStateOffIdent;
        fa.tables = J.synrgb ? ctx->tables.as<uint8_t>() : nullptr;
        fa.supp_rg = consts + kChainOffSupp;
        fa.blue_pair_supp = consts + kChainOffBlue;
        fa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        fa.suppressed = 1; // CLAHE always composes with the suppressed variant (synthetic_rgb.rs:188-194)
        KernelTimer t(ctx, "chain_finish");
        HIPCHK(ctx, launch_chain_finish(fa, ctx->stream));
    }
    if (J.synrgb) {
        {
            KernelTimer t(ctx, sampled ? "spec_fallback_compose" : "compose_u8");
            HIPCHK(ctx, launch_compose_u8(c, cvec, ctx->stream));
        }
        for (int b = 0; b < 2; ++b) // optional per-band u8 rasters: levels through the band's rescale
            if (d_out[b])
                HIPCHK(ctx, launch_chain_remap(J.d_levels[b], J.lvl_pitch, reinterpret_cast<uint8_t *>(d_out[b]), out_pitch, rows, cols,
                                               state + kStateOffResc + (size_t)b * 256, nullptr, ctx->stream));
    } else {
        // single band: the apply pass wrote levels into the caller's raster; rescale in place unless it is the identity
        HIPCHK(ctx, launch_chain_remap(reinterpret_cast<uint8_t *>(d_out[0]), out_pitch, reinterpret_cast<uint8_t *>(d_out[0]), out_pitch, rows,
                                       cols, state + kStateOffResc, state + kStateOffIdent, ctx->stream));
    }
    return chain_tail(J, stats_out, d_state);
}

// Device-resident chain for the percentile strategies, dual-pol, RGB only: histogram -> statistics + window +
// u8 level of every DN + level histogram (k_chain_stats, levels mode) -> rescale / floor / tables and DN -> final
// u8 tables (k_chain_finish) -> ONE fused pass DN,DN -> RGB (k_lut_compose_u16).  No host synchronisation in
// between; gamma != 1 is resolved against host-built thresholds, so no pow runs on the device.
// dual-pol -> RGB (fused pass) and / or per-band u8 rasters (table pass), all with device-built tables
bool chain_levels_eligible(const U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, const uint8_t *d_rgb,
                                  size_t rgb_pitch_px) {
    if (J.ctx->attrs.on(A_NO_CHAIN)) return false;
    if (J.clahe() || !J.vec || !(J.reduce || (J.row0 == 0 && J.rows_local == J.rows_total))) return false;
    if (!J.u8_out() && J.synrgb) return false; // u16 levels: per-band rasters only
    if (J.synrgb && !(J.nbands == 2 && J.in_pitch % 16 == 0 && rgb_pitch_px % 16 == 0 && ptr_aligned16(d_rgb))) return false;
    bool any_out = false;
    for (int b = 0; b < J.nbands; ++b)
        if (d_out[b]) { any_out = true; if (out_pitch % 8 != 0 || !ptr_aligned16(d_out[b])) return false; }
    return J.synrgb || any_out || J.tables_only;
}


int job_run_chain_levels(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                                sarpro_hip_stats *stats_out) {
    sarpro_hip_ctx *ctx = J.ctx;
    const int nb = J.nbands;
    const bool u16o = !J.u8_out();
    RETCHK(chain_prepare(ctx));
    HIPCHK(ctx, ctx->luts.reserve(2 * 131072));
    HIPCHK(ctx, ctx->level_hist.reserve(sizeof(uint64_t) * 256 * kMaxBands));
    HIPCHK(ctx, ctx->tables.reserve(kTablesBytes));
    HIPCHK(ctx, ctx->h_small.reserve(sizeof(uint64_t) * 64 * 256 * kMaxBands + sizeof(uint64_t) * 256 * kMaxBands));
    uint8_t *consts = ctx->chain_consts.as<uint8_t>(), *state = ctx->chain_state.as<uint8_t>();
    ChainBandState *d_state = reinterpret_cast<ChainBandState *>(state);
    const bool suppressed = J.strategy == SARPRO_STRATEGY_TAMED; // CLAHE is not handled here

    J.clear_after_sum = true; // untiled: k_sum_tile_hists is the only reader of the histogram
    RETCHK(job_phase1(J)); // DN histograms -> ctx->ghist
    RETCHK(chain_reduce(J, ctx->ghist.p, 65536 * (size_t)nb, "allreduce_dn_hist"));
    {
        ChainStatsArgs sa{};
        sa.ghist = ctx->ghist.as<unsigned long long>();
        sa.db = reinterpret_cast<const double *>(consts + kChainOffDb);
        sa.state = d_state;
        sa.binlut = ctx->luts.as<uint8_t>();
        sa.binlut_stride = 131072;
        sa.levels_mode = u16o ? 2 : 1;
        sa.lut16 = ctx->luts.as<uint16_t>();
        sa.strategy = J.strategy;
        for (int b = 0; b < nb; ++b) sa.tamed_kind[b] = J.tamed_kind(b);
        sa.total_px = (unsigned long long)J.rows_total * J.cols;
        sa.level_hist = ctx->level_hist.as<unsigned long long>();
        sa.gamma_thr = reinterpret_cast<const double *>(consts + kChainOffGamma);
        KernelTimer t(ctx, "chain_stats");
        RETCHK(chain_stats_scratch(ctx, &sa));
        HIPCHK(ctx, launch_chain_stats(sa, nb, ctx->stream));
    }
    if (!u16o) { // u16 levels have no rescale (autoscale.rs:689-703) and are never composed
        ChainFinishArgs fa{};
        fa.level_hist = ctx->level_hist.as<unsigned long long>();
        fa.total_px = (unsigned long long)J.rows_total * J.cols;
        fa.nbands = nb;
        fa.resc_out = state + kStateOffResc;
        fa.identity_out = state + kStateOffIdent;
        fa.tables = J.synrgb ? ctx->tables.as<uint8_t>() : nullptr; // no composition: only the DN -> final u8 tables
        fa.supp_rg = consts + kChainOffSupp;
        fa.blue_pair_supp = consts + kChainOffBlue;
        fa.floor_out = reinterpret_cast<int *>(state + kStateOffFloor);
        fa.levels_mode = 1;
        for (int b = 0; b < nb; ++b) fa.no_rescale[b] = J.tamed_kind(b) != kNotTamedSynrgb;
        fa.suppressed = suppressed ? 1 : 0;
        fa.dn_tables = ctx->luts.as<uint8_t>();
        fa.dn_table_stride = 131072;
        fa.default_rg = consts + kChainOffDefRg;
        fa.blue_pair_default = consts + kChainOffBlueDef;
        KernelTimer t(ctx, "chain_finish");
        HIPCHK(ctx, launch_chain_finish(fa, ctx->stream));
    }
    for (int b = 0; b < nb; ++b) { // per-band u8 rasters: out = table[DN]
        if (!d_out[b]) continue;
        LutApplyArgs la{};
        la.in = J.d_in[b]; la.out = d_out[b]; la.in_pitch = J.in_pitch; la.out_pitch = out_pitch;
        la.rows = (uint32_t)J.rows_local; la.cols = (uint32_t)J.cols;
        la.lut = ctx->luts.as<uint8_t>() + (size_t)b * 131072; // u8 final values, or u16 levels (65536 x 2 bytes) at the same offset
        la.dev_state = d_state; la.band = b; la.lut_cap = ctx->chain_levels_cap;
        KernelTimer t(ctx, "lut_apply_u16");
        HIPCHK(ctx, launch_lut_apply_u16(la, true, u16o, ctx->stream));
    }
    if (J.synrgb) {
        LutComposeArgs f{};
        for (int b = 0; b < 2; ++b) { f.in[b] = J.d_in[b]; f.lut[b] = ctx->luts.as<uint8_t>() + (size_t)b * 131072; }
        f.rgb = d_rgb; f.in_pitch = J.in_pitch; f.rgb_pitch_px = rgb_pitch_px;
        f.rows = (uint32_t)J.rows_local; f.cols = (uint32_t)J.cols;
        f.tables = ctx->tables.as<uint8_t>();
        f.dev_state = d_state;
        f.lut_cap = ctx->chain_levels_cap;
        KernelTimer t(ctx, "lut_compose_u16");
        HIPCHK(ctx, launch_lut_compose_u16(f, ctx->stream));
    }
    if (ctx->async_dev && J.allow_async && !stats_out && !u16o && !J.reduce) { // stream-ordered (u16 levels need their `uncertain` flag read back)
        ctx->async_pending = ctx->timing;
        return SARPRO_HIP_OK;
    }
    ChainBandState *h_state = ctx->h_small.as<ChainBandState>();
    HIPCHK(ctx, hipMemcpyAsync(h_state, d_state, sizeof(ChainBandState) * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the only synchronisation of the chain
    uint32_t hi = 0;
    bool uncertain = false;
    for (int b = 0; b < nb; ++b) {
        J.stats[b] = h_state[b].stats;
        if (stats_out) stats_out[b] = J.stats[b];
        hi = std::max(hi, h_state[b].win_hi);
        uncertain = uncertain || (u16o && h_state[b].uncertain);
    }
    if (u16o && ctx->attrs.on(A_FORCE_UNCERTAIN)) uncertain = true; // test hook: exercise the rerun
    if (uncertain) return kRerunOnHostRoute;
    // LDS capacity (bytes per band) of the NEXT scene's DN tables (speed only)
    ctx->chain_levels_cap = std::min<uint32_t>(16384, std::max<uint32_t>(2048, (hi + 1 + 1023) / 1024 * 1024));
    return SARPRO_HIP_OK;
}

int job_run_all(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px,
                       sarpro_hip_stats *stats_out) {
    timing_reset(J.ctx);
    J.ctx->spec_ran = false;
    RETCHK(job_init(J));
    if ((J.rows_local == 0 && !J.reduce) || J.cols == 0) { // a rank with an empty stripe still joins the reductions
        if (stats_out) std::memset(stats_out, 0, sizeof(*stats_out) * (size_t)J.nbands);
        return SARPRO_HIP_OK;
    }
    if (chain_eligible(J)) {
        HostTimer t(J.ctx, "host:chain(enqueue+final sync)");
        return job_run_chain(J, d_out, out_pitch, d_rgb, rgb_pitch_px, stats_out);
    }
    if (chain_levels_eligible(J, d_out, out_pitch, d_rgb, rgb_pitch_px)) {
        HostTimer t(J.ctx, "host:chain(enqueue+final sync)");
        const int rc = job_run_chain_levels(J, d_out, out_pitch, d_rgb, rgb_pitch_px, stats_out);
        if (rc != kRerunOnHostRoute) return rc;
    }
    { HostTimer t(J.ctx, "host:phase1_launch"); RETCHK(job_phase1(J)); }
    // (row-stripe mode without the device chain: the same phases with a synchronous all-reduce after each)
    if (J.reduce) RETCHK(sarpro_hip_comm_allreduce_sum_u64(J.ctx, J.ctx->ghist.as<uint64_t>(), 65536 * (size_t)J.nbands));
    { HostTimer t(J.ctx, "host:after_phase1(sync+stats+tables)"); RETCHK(job_after_phase1(J)); }
    { HostTimer t(J.ctx, "host:phase2_launch"); RETCHK(job_phase2(J)); }
    if (J.reduce && J.clahe()) RETCHK(sarpro_hip_comm_allreduce_sum_u64(J.ctx, J.ctx->tile_bins.as<uint64_t>(), 64 * 256 * (size_t)J.nbands));
    { HostTimer t(J.ctx, "host:phase3(sync+cdfs+launch)"); RETCHK(job_phase3(J, d_out, out_pitch)); }
    if (J.reduce && J.clahe() && J.u8_out()) RETCHK(sarpro_hip_comm_allreduce_sum_u64(J.ctx, J.ctx->level_hist.as<uint64_t>(), 256 * kMaxBands));
    { HostTimer t(J.ctx, "host:phase4(sync+tables+launch+sync)"); RETCHK(job_phase4(J, d_out, out_pitch, d_rgb, rgb_pitch_px, false)); }
    if (stats_out) for (int b = 0; b < J.nbands; ++b) stats_out[b] = J.stats[b];
    return SARPRO_HIP_OK;
}

} // namespace sarpro

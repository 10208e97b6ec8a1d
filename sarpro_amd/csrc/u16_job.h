// u16_job.h -- the u16 (integer-DN) flavour's job object and the steps the entry points, the row-stripe protocol and the
// device-resident chains share (api_u16_phases.cpp, api_u16_chain.cpp, api.cpp).
#pragma once
#include "chain_kernels.h"
#include "context.h"
#include "host_logic.h"
#include "internal.h"

namespace sarpro {

struct U16Job {
    sarpro_hip_ctx *ctx = nullptr;
    int nbands = 1;
    const uint16_t *d_in[kMaxBands] = {nullptr, nullptr};
    size_t rows_total = 0, cols = 0, row0 = 0, rows_local = 0, in_pitch = 0;
    int strategy = 0, bit_depth = 0, mode = 0;
    bool synrgb = false; // dual-pol JPEG branch (save.rs:317-367): always U8, Tamed uses tamed_synrgb
    int tamed_force = 0; // single band tamed_synrgb entry point: 1 copol, 2 crosspol
    bool vec = false;
    bool reduce = false; // row stripe of a multi-rank scene: histograms are all-reduced over ctx->comm, on the stream
    bool hist_done = false; // phase 1 already ran (streaming ingest: chunk by chunk, under the upload)
    bool clear_after_sum = false; // untiled chain: k_sum_tile_hists is the last reader of the tile histogram and zeroes it
    size_t tile_hist_bytes = 0;   // footprint of this job's histogram pass in ctx->tile_hist[0]
    bool allow_async = false; // the entry point may return once the device chain is enqueued (SARPRO_HIP_CTX_ASYNC_DEV)
    bool tables_only = false; // percentile chain: stop at the DN -> final u8 tables (band_u8_table_dev)
    StripePlan *plan = nullptr;
    // host-side state between phases
    sarpro_hip_stats stats[kMaxBands];
    DnLut lut[kMaxBands];
    uint8_t resc[kMaxBands][256];
    bool resc_identity[kMaxBands] = {true, true};
    uint64_t level_hist_h[kMaxBands][256];
    int floor_with_cushion = -1;
    // level rasters (u8) when an intermediate is needed
    uint8_t *d_levels[kMaxBands] = {nullptr, nullptr};
    size_t lvl_pitch = 0;

    bool clahe() const { return strategy == SARPRO_STRATEGY_CLAHE; }
    int tamed_kind(int band) const {
        if (tamed_force) return tamed_force;
        if ((synrgb || (tables_only && nbands == 2)) && strategy == SARPRO_STRATEGY_TAMED) return band == 0 ? kTamedCopol : kTamedCrosspol;
        return kNotTamedSynrgb;
    }
    bool u8_out() const { return synrgb || tamed_force || bit_depth == SARPRO_BITDEPTH_U8; }
};

inline bool ptr_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int job_init(U16Job &J);
void mark_tile_hist_clean(U16Job &J);
uint32_t *tile_hist_of(sarpro_hip_ctx *ctx, int band, int ntiles);
// the histogram pass can be issued in pieces (streaming ingest): see api_u16_phases.cpp
int job_phase1(U16Job &J, bool begin = true, int first = 0, int last = -1, bool end = true);
int job_after_phase1(U16Job &J);
void job_rescale_from_level_hist(U16Job &J, int b);
int job_phase2(U16Job &J);
int ensure_levels(U16Job &J);
int job_phase3(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch);
int job_phase4(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px, bool level_hist_reduced_on_device);
bool chain_eligible(const U16Job &J);
bool chain_levels_eligible(const U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, const uint8_t *d_rgb, size_t rgb_pitch_px);
constexpr int kRerunOnHostRoute = 1; // u16 levels with gamma != 1 that the device could not certify: the host route runs them again
int job_run_chain(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out);
int job_run_chain_levels(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out);
int job_run_all(U16Job &J, void *const d_out[kMaxBands], size_t out_pitch, uint8_t *d_rgb, size_t rgb_pitch_px, sarpro_hip_stats *stats_out);

} // namespace sarpro

// chain_kernels.h -- device-side "host half" of the CLAHE u8 chain (chain_kernels.hip).
#pragma once
#include "kernels.h"

namespace sarpro {

struct ChainBandState { // lives in device memory, one per band
    sarpro_hip_stats stats;
    uint32_t win_hi;    // first DN whose dB value reached the high clip (the bin table is constant above)
    uint32_t uncertain; // u16 levels with gamma != 1: some DN's level could not be certified (rerun on the host route)
};

struct ChainStatsPartial { // one per (band, 4096-DN slice): kernel A of the statistics step
    unsigned long long count;
    uint32_t min_dn, max_dn;
    double sum1, sum2;
};
constexpr int kChainStatsParts = 16;

struct ChainStatsArgs {
    const unsigned long long *ghist; // [nbands][65536]
    const double *db;                // [65536] dB value of every DN (host-built, glibc)
    ChainBandState *state;           // [nbands]
    uint8_t *binlut;                 // per band, binlut_stride bytes apart: CLAHE bin or u8 level of every DN
    size_t binlut_stride;
    // percentile strategies (levels mode): window by strategy, u8 level of every DN, level histogram
    int levels_mode;                 // 0: CLAHE bins (window p01..p99); 1: u8 levels of `strategy`; 2: u16 levels of `strategy`
    uint16_t *lut16;                 // levels_mode 2: [nbands][65536] u16 level of every DN
    int strategy;
    int tamed_kind[kMaxBands];       // 0 / 1 copol / 2 crosspol (autoscale.rs:721-727)
    unsigned long long total_px;     // pixels per band (level 0 also counts the invalid ones)
    unsigned long long *level_hist;  // [nbands][256]: cleared by kernel A (levels mode: filled by kernel C; CLAHE: by the apply kernel)
    unsigned long long *sample_valid;// [kSampleReplicas][kMaxBands] or null: cleared by kernel A together with the histogram's replicas (CLAHE chain with a
                                     // sampled level histogram: level_hist is then [kSampleReplicas][kMaxBands][256])
    const double *gamma_thr;         // [3][256]: x-thresholds of trunc(pow(x, g) * 255) for g = 0.8, 0.9, 1.1 (host-built)
    ChainStatsPartial *partials;     // scratch [nbands][kChainStatsParts]
    unsigned long long *bins4096;    // scratch [nbands][4096]
};

struct ChainFinishArgs {
    const unsigned long long *level_hist; // [nbands][256]; bin 0 is restored here
    unsigned long long total_px;          // pixels per band of the whole scene
    int nbands;
    uint8_t *resc_out;                    // [2][256] u8 rescale maps
    uint8_t *identity_out;                // [2] 1 when the band's rescale is the identity on the occupied levels
    uint8_t *tables;                      // compose tables R2|G2|B2 (nbands == 2), may be null
    const uint8_t *supp_rg;               // [41][512] suppressed lut_r|lut_g for every floor value
    const uint8_t *blue_pair_supp;        // [256][256]
    int *floor_out;                       // optional
    // levels mode (percentile strategies): bin 0 of level_hist is exact, the DN -> level tables become
    // DN -> FINAL u8 tables (rescale applied) and the compose tables are built without folding it again
    int levels_mode;
    int no_rescale[2];                    // tamed-synrgb bands have no u8 rescale (autoscale.rs:731-741)
    int suppressed;                       // 1: suppressed synRGB variant (Tamed / Clahe), 0: default variant
    uint8_t *dn_tables;                   // [2] x dn_table_stride: in: level of every DN, out: final u8 of every DN
    size_t dn_table_stride;
    const uint8_t *default_rg;            // [512] default lut_r | lut_g (synthetic_rgb.rs:22-29)
    const uint8_t *blue_pair_default;     // [256][256]
    const ChainSpecState *gate;           // speculative CLAHE chain: run only when the verdict refuted the speculation (null: always)
};

hipError_t launch_chain_stats(const ChainStatsArgs &a, int nbands, hipStream_t s);
hipError_t launch_chain_cdfs(const unsigned long long *tile_bins, double *cdfs, uint32_t rows, uint32_t cols, int nbands,
                             hipStream_t s);
hipError_t launch_chain_finish(const ChainFinishArgs &a, hipStream_t s);
// partial level histogram (apply kernel, ClaheApplyArgs::partial_hist): check that its exact bins suffice, else clear
// it and flag the band; the second kernel recounts a flagged band from its level raster
hipError_t launch_level_hist_guard(unsigned long long *level_hist, unsigned long long total_px, int nbands, uint32_t *d_flags,
                                   hipStream_t s);
struct LevelRecountArgs {
    const uint8_t *levels[kMaxBands];  // level rasters
    size_t pitch;
    uint32_t rows, cols;
    unsigned long long *level_hist;    // [nbands][256]
    const uint32_t *flags;             // [nbands], written by k_level_hist_guard
    const ChainSpecState *gate;        // set: flags are ignored, every band is recounted iff gate->verdict != 0
};
// CLAHE chain with a sampled level histogram: identity proof + predicted floor + compose tables (k_chain_predict)
struct ChainPredictArgs {
    const unsigned long long *sample_hist; // [kSampleReplicas][2][256] partial level histogram of the sampled rows (bin 0 implied), summed here
    unsigned long long *exact_hist;        // [2][256] cleared here: the gated recount adds into it
    ChainSpecState *spec;
    const ChainBandState *state;           // stats.valid_count per band
    unsigned long long total_px;           // pixels per band
    uint8_t *resc_out, *identity_out;      // [2][256] identity maps, [2] ones (they stand iff the verdict accepts)
    int *floor_out;
    uint8_t *tables;                       // compose tables R2|G2|B2
    const uint8_t *supp_rg;                // [41][512]
    const uint8_t *blue_pair_supp;         // [256][256]
    uint32_t force;                        // kSpecForce*
    uint32_t allow_rescaled;               // the consumer verifies a predicted lowest level (the fused CLAHE -> RGB pass does)
    const float *blue_pq;                  // P[256] | Q[256] (host-built, verified) or null
    float *blue_by_level;                  // out: Pv[256] | Qv[256] with Pv[v] = P[R2[v]], Qv[v] = Q[G2[v]] (the fused pass's LITE form reads these instead of B2)
    uint32_t second;                       // 1: the launch behind a refuted fused pass: runs iff ChainSpecState::retry_min (an undercut lowest level whose true value the
                                           // pass recorded), with that level instead of the sample's, and arms the retry kernel
};
hipError_t launch_chain_predict(const ChainPredictArgs &a, hipStream_t s);
struct ChainRepredictArgs {
    ChainSpecState *spec;
    const uint8_t *resc_in;                // [2][256] the bands' u8 rescale tables as k_chain_predict left them (identity, or the predicted rescale)
    int *floor_out;
    uint8_t *tables;                       // compose tables R2|G2|B2
    const uint8_t *supp_rg;                // [41][512]
    const uint8_t *blue_pair_supp;         // [256][256]
    const float *blue_pq;
    float *blue_by_level;
    uint32_t stripes;                      // 1: a row stripe (the counts are all-reduced behind the retry kernel whether it ran or not)
};
hipError_t launch_chain_repredict(const ChainRepredictArgs &a, hipStream_t s); // behind the fused pass's verdict, before launch_clahe_rgb_fused_retry
hipError_t launch_level_hist_if_flagged(const LevelRecountArgs &a, int nbands, hipStream_t s);
// dst = map[src] unless skip_flag && *skip_flag (device byte) is non-zero and src == dst
hipError_t launch_chain_remap(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, uint32_t rows,
                              uint32_t cols, const uint8_t *d_map, const uint8_t *d_skip_flag, hipStream_t s);

} // namespace sarpro

// fused_kernels.h -- the fused CLAHE pass of the headline chain (fused_kernels.hip): both DN rasters in,
// interleaved RGB out, no level rasters in between (save.rs:317-367 at native resolution:
// autoscale.rs:572-608 per band, then synthetic_rgb.rs:88-178).
#pragma once
#include "chain_kernels.h"

namespace sarpro {

// One piece of a workgroup's share of the scene: a strip of `1 << gx_log2` wave columns (256 px each) inside one
// interpolation cell x a row range.  The 16 waves of the workgroup stand gx wide x 16/gx tall on it.
struct FusedItem {
    int32_t r0, r1;   // local rows [r0, r1)
    int32_t c0, c1;   // columns this piece owns
    int32_t cstart;   // column of lane 0 of wave column 0: a multiple of 4, <= c0
    int32_t gx_log2;
    int32_t flags;    // bit 0: the cell extrapolates (negative blend weights)
    int32_t id[4];    // tiles t00, t01, t10, t11
    int32_t tile;     // the tile this cell lies in (strata of the sample pass)
};
static_assert(sizeof(FusedItem) == 48, "FusedItem layout");

enum FusedMode { kFusedSample = 0, kFusedSpec = 1, kFusedHist = 2, kFusedFinal = 3 };
constexpr int kFusedMaxGrid = 1024;
// queue capacity per workgroup (entries of 16 B), sized by the planner from the share's pixels: ~0.3 % of the pixels of an
// interior cell are uncertain, but in an extrapolating cell every pixel of a saturated bin is (its level is 254 or 255
// depending on the rounding of the weights) -- up to ~10 % in bright regions
constexpr double kFusedQueueRateInner = 0.015, kFusedQueueRateEdge = 0.16;
constexpr uint32_t kFusedQueueMin = 4096;    // a queue that still overflows has its share redone by the fixup
// test switches (FusedArgs::force, from SARPRO_HIP_FUSED_FORCE)
constexpr uint32_t kFusedForceNoSpec = 1u;      // preconditions "fail": histogram pass + exact tables + final pass
constexpr uint32_t kFusedForceMispredict = 2u;  // predicted floor + 1: the verification must refute it
constexpr uint32_t kFusedForceTwoLevel = 4u;    // windows "do not fit": DN -> bin from global memory
constexpr size_t kFusedHist3Words = 2ull * 64 * 256 * 32;
constexpr uint32_t kFusedForceTinyQueue = 8u;   // queue of 4 entries: the inline exact path takes the rest

struct FusedState { // device memory, one per context; reset by k_fused_prep
    uint32_t spec_ok;     // the speculative pass may run (see k_fused_prep)
    uint32_t direct;      // both windows fit the DN-indexed LDS tables
    uint32_t k_base[2];   // entry index (LDS byte offset / 16) of band b's DN = 0 entry
    uint32_t verdict;     // after the speculative pass: 0 = its RGB is final, 1 = prediction refuted
    int32_t floor_pred;   // predicted suppression floor (synthetic_rgb.rs:99-113), before the +3 cushion
    uint32_t fix_done[4]; // workgroups of each fixup launch that have finished
    uint32_t pad0[2];
    unsigned long long n_lt[2];        // speculative pass: kept band-pixels with level < floor_pred, < floor_pred + 1
    double cum_est[32];                // predicted cumulative count of band-pixels with level <= l (k_fused_predict)
    double unsampled;                  // valid pixels of strata the sample never hit
    uint32_t predict_done, pad1;
    unsigned long long total_px;       // pixels per band of the scene
    unsigned long long dbg[8];         // diagnostics of the fixup (speculative pass): see sarpro_hip_fused_report
    uint32_t dbg_n, dbg_pad; uint32_t dbg_samples[64][8];
    uint32_t qcount[4][kFusedMaxGrid]; // per pass and workgroup: queued (uncertain) pixels; bit 31: the queue overflowed
};

struct FusedArgs {
    const uint16_t *in[kMaxBands];
    size_t in_pitch;                    // elements, % 8 == 0
    uint8_t *rgb;
    size_t rgb_pitch_px;                // % 16 == 0
    const FusedItem *items;
    const int32_t *wg_first;            // [grid + 1]
    const double *cdfs[kMaxBands];      // [64][256]
    const uint8_t *binlut[kMaxBands];   // DN -> CLAHE bin, constant from win_hi on
    const ChainBandState *state;        // win_hi per band
    const RowWeight *row_w, *col_w;     // row_w indexed by GLOBAL row
    const float *row_wf, *col_wf;       // the same weights (dy, dx) rounded to f32
    int32_t row_off;
    FusedState *fs;
    const uint8_t *tables;              // R2[256] | G2[256] | B2[65536]
    uint4 *queue;                       // (row, column, DN1 | DN2 << 16, -) of the uncertain pixels, workgroup w owns [qoff[w], qoff[w+1])
    const uint32_t *qoff;               // [grid + 1]
    uint32_t *hist3;                    // sample pass: [2][64][256][32] sampled level counts per (band, tile, CLAHE bin)
    uint8_t *dump;                      // kSpecDumpBytes of write-only scratch
    unsigned long long *level_hist;     // [2][256], histogram pass (bin 0 stays implied)
    uint32_t sample_stride;             // sample pass: every sample_stride-th step
    uint32_t force;
};

struct FusedPrepArgs {
    FusedState *fs;
    const ChainBandState *state;
    const unsigned long long *tile_bins; // [2][64][256]
    const double *cdfs;                  // [2][64][256]
    unsigned long long total_px;
    uint32_t force;
};

struct FusedPredictArgs { // sample -> predicted floor
    FusedState *fs;
    const ChainBandState *state;
    const unsigned long long *tile_bins; // [2][64][256] exact valid-pixel counts per (band, tile, CLAHE bin)
    uint32_t *hist3;                     // [2][64][256][32], zeroed again here
    unsigned long long total_px;
    uint32_t force;
};

struct FusedTablesArgs { // predicted floor -> compose tables (identity rescale)
    FusedState *fs;
    uint8_t *tables;
    const uint8_t *supp_rg;        // [41][512]
    const uint8_t *blue_pair_supp; // [256][256]
    uint32_t force;
};

struct DnHistPiecesArgs { // per-tile DN histograms of both bands over the fused pass's pieces (k_dn_hist_pieces)
    const uint16_t *in[kMaxBands];
    uint32_t *tile_hist[kMaxBands]; // [64][65536], zeroed by the caller
    size_t pitch;                   // elements, % 4 == 0
    const FusedItem *items;
    const int32_t *wg_first;        // [grid + 1]
    uint32_t lds_bins;              // DN < lds_bins are privatised in LDS
};
hipError_t launch_dn_hist_pieces(const DnHistPiecesArgs &a, int grid, hipStream_t s);

hipError_t fused_configure(); // once per device: dynamic LDS opt-in of the fused kernels
hipError_t launch_fused_prep(const FusedPrepArgs &a, hipStream_t s);
hipError_t launch_fused_predict(const FusedPredictArgs &a, hipStream_t s);
hipError_t launch_fused_tables_predict(const FusedTablesArgs &a, hipStream_t s);
hipError_t launch_fused_main(const FusedArgs &a, int mode, int grid, hipStream_t s);
hipError_t launch_fused_fixup(const FusedArgs &a, int mode, int grid, unsigned long long total_px, hipStream_t s);

} // namespace sarpro

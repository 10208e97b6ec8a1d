// kernels.h -- launch interface of the gfx950 kernels (kernels.hip).  Plain structs and
// device pointers; every launcher enqueues on the given stream and returns hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "host_logic.h"

namespace sarpro {

constexpr int kMaxBands = 2;

// Work item of the position-dependent kernels: a column strip x row range that lies inside
// one CLAHE tile (histogram pass) or one interpolation cell (apply pass).  Rows/cols are in
// LOCAL raster coordinates (the stripe this rank holds).
struct Rect {
    int32_t r0, r1;         // rows [r0, r1)
    int32_t c0, c1;         // columns [c0, c1) this item owns
    int32_t cstart;         // c0 rounded down to the vector width: lane l starts at cstart + l*VEC
    int32_t id[4];          // hist: id[0] = tile index;  apply: tile indices t00, t01, t10, t11
    int32_t pad[3];
};
static_assert(sizeof(Rect) == 48, "Rect layout");

struct DnHistArgs {
    const uint16_t *in[kMaxBands];
    uint32_t *tile_hist[kMaxBands]; // [ntiles][65536], zeroed by the caller
    size_t pitch;                   // elements
    const Rect *rects;
    uint32_t lds_bins;              // DN < lds_bins are privatised in LDS
};

struct ClaheApplyArgs {
    const uint16_t *in[kMaxBands];
    void *out[kMaxBands];                   // u8 (levels 0..255) or u16
    size_t in_pitch, out_pitch;             // elements
    const Rect *rects;
    const double *cdfs[kMaxBands];          // [64][256]
    const uint8_t *binlut[kMaxBands];       // full 65536-entry DN -> bin table
    uint32_t win_lo[kMaxBands], win_hi[kMaxBands];
    uint32_t lut_in_lds;                    // window staged in LDS (else gathered from global)
    const RowWeight *row_w;                 // indexed by GLOBAL row (row_off + local row)
    const RowWeight *col_w;                 // indexed by column
    unsigned long long *level_hist[kMaxBands]; // [256] histogram of u8 levels, or null
    int32_t row_off;
    double max_val;                         // 255.0 or 65535.0
    const struct ChainBandState *dev_state;  // chain mode: win_hi is read from device memory (null: use win_hi[])
    uint32_t lut_cap;                       // speculative kernel: LDS capacity of the offset table (entries)
    uint32_t hist_mode;                     // speculative kernel: 0 full level histogram, 1 levels >= 64 only counted in bulk (see
                                            // k_level_hist_guard), 2 the bulk form on sampled rows only (see k_chain_predict)
    uint32_t sample_stride, sample_phase;   // hist_mode 2: row r is sampled iff (row_off + r) % sample_stride == sample_phase; stride > 4
    unsigned long long *sample_valid;       // hist_mode 2, 3: [kSampleReplicas][nbands] valid (DN != 0) pixels on the sampled rows, added to
                                            // (hist_mode 2, 3: level_hist[b] is replica 0 of [kSampleReplicas][kMaxBands][256] as well)
    const struct ChainSpecState *gate;      // set: the launch is the fused chain's fallback and runs only if gate->verdict != 0
    uint8_t *dump;                          // speculative kernel: kSpecDumpBytes of scratch that edge lanes' full-width stores go to
};

struct LutApplyArgs {
    const uint16_t *in;
    void *out;                  // u8 or u16
    size_t in_pitch, out_pitch; // elements
    uint32_t rows, cols;
    const void *lut;            // full 65536-entry table of u8 (OUT16 = false) or u16 entries
    uint32_t win_lo, win_hi;
    uint32_t lut_in_lds;
    const struct ChainBandState *dev_state; // chain mode: the window is [0, dev_state[band].win_hi], staged in LDS when it is below lut_cap entries
    int band;
    uint32_t lut_cap;
};

struct ComposeArgs {
    const uint8_t *b1, *b2;
    uint8_t *rgb;
    size_t in_pitch, rgb_pitch_px; // elements / pixels
    uint32_t rows, cols;
    const uint8_t *tables;         // R2[256] | G2[256] | B2[65536]
    struct ChainSpecState *spec;   // CLAHE chain with a predicted floor (chain_kernels.h), else null
    int speculative;               // 1: the speculative composition (runs iff spec->spec_ok, counts, decides spec->verdict);
                                   // 0 with spec set: the fallback composition (runs iff spec->verdict != 0)
};

// State of the CLAHE chain's speculation on the synRGB floor (device memory, one per context).  k_chain_predict writes it
// after the apply pass, the speculative compose pass adds its counts and the verdict, the gated exact kernels read it.
constexpr unsigned long long kSampleWeightOne = 4096;
// The sampling passes end with every workgroup adding its ~100 occupied bins to the SAME global words: 1760 workgroups on one word
// serialise for ~25 us (the sample-only pass took 0.035 ms with four rows per item and 0.011 with none).  They add into one of
// kSampleReplicas copies instead (by workgroup index); k_chain_predict sums the copies.
constexpr int kSampleReplicas = 16; // fixed-point 1.0 of the sampled histogram's per-item weights
constexpr int kSpecFloorCap = 37; // synthetic_rgb.rs:110-113: floor + 3 is capped at 40, so every floor >= 37 is the same floor
struct ChainSpecState {
    uint32_t spec_ok;              // kSpecIdentity: both bands hold level 0 and level 255 (=> the u8 rescale is the identity) and a floor was
                                   // predicted; kSpecRescaled: a band's lowest level is PREDICTED (min_pred > 0), its rescale folded into the
                                   // tables, and the fused pass verifies that too; 0: no speculation
    uint32_t verdict;              // 0: the speculative RGB is final; 1: refuted (or never composed): the exact kernels run
    int32_t floor_pred;            // predicted floor F (before the +3 cushion), kSpecFloorCap = "at least that"
    uint32_t done;                 // workgroups of the speculative compose pass that have added their counts
    unsigned long long n_lt[2];    // band-pixels with final level < F, < F + 1, counted by the speculative compose pass
    unsigned long long n_below_min; // kSpecRescaled: valid-raster band-pixels with level < min_pred of their band (0 or the prediction is refuted); follows n_lt (one all-reduce)
    unsigned long long below_hist[2][128]; // kSpecRescaled: per band, how often level l was the LOWEST byte a lane met below min_pred in a row (the rare branch of the
                                   // fused pass adds here; row stripes: summed with the counts -- a minimum does not travel through a sum, the presence of a level does)
    unsigned long long target;     // synthetic_rgb.rs:99-100
    uint32_t min_pred[2];          // kSpecRescaled: predicted lowest level of each band (the highest is 255: proven from the sample)
    uint32_t thr[2][2];            // kSpecRescaled: per band the lowest LEVEL whose final value is >= F, >= F + 1 (256: none)
    unsigned long long sample_valid[2]; // valid pixels on the sampled rows, per band, weighted like the histogram (summed by k_chain_predict)
    unsigned long long sample_valid_rep[kSampleReplicas * 2]; // [replica][band]: what the workgroups of the sampling pass add to
    double est_lt[2];              // the sample's estimate of n_lt (diagnostics)
    uint32_t force;                // test switches (kSpecForce*)
    uint32_t pool_overflow;        // the fused CLAHE -> RGB pass stepped aside: the bands' DN windows do not fit its LDS pool
    uint32_t next_item;            // the fused pass's work list: the next item to hand out (cleared by k_chain_predict; ends at items + workgroups)
    // A refuted floor gets ONE second fused pass before the exact kernels (round 6): the pass's own counts say on which side of the
    // prediction the floor lies (n_lt[0] >= target: below it; target > n_lt[1]: above it), the sample's error is rarely more than one level.
    int32_t retry_floor;           // written with the verdict: the floor to try next (-1: none -- accepted, a lowest level undercut, no room on that side)
    uint32_t retry_armed;          // k_chain_repredict: tables, thresholds and counters are set up for the second pass (the retry kernel runs iff 1)
    uint32_t retried;              // the second pass ran: `verdict` is its verdict, floor_pred the floor it tried
    int32_t floor_first;           // the floor the first pass tried (diagnostics)
    unsigned long long saved_counts[3]; // row stripes without a second pass: the summed counts, set aside while the second all-reduce sums zeros
    // A lowest level the sample missed (n_below_min != 0): the pass that finds such bytes records the lowest of them per band, and the
    // second pass runs with the TRUE lowest level (k_chain_predict again, its estimate rebuilt on the new rescale); one device only.
    uint32_t true_min[2];          // lowest level byte below min_pred the fused pass met, per band (256: none)
    uint32_t retry_min;            // written with the verdict: 1 = undercut, the true lowest levels are known: k_chain_predict's second launch takes them
};
constexpr uint32_t kSpecIdentity = 1u, kSpecRescaled = 2u;
constexpr size_t kSpecCountWords = 3 + 2 * 128; // n_lt[2] | n_below_min | below_hist: what a row stripe all-reduces behind a fused pass
constexpr uint32_t kSpecForceMispredict = 1u; // predicted floor + 1 (- 1 at the cap): the verification must refute it
constexpr uint32_t kSpecForceNoSpec = 2u;     // "level 0 or 255 missing": no speculative composition at all
constexpr uint32_t kSpecForceMinMispredict = 4u; // a predicted lowest level + 1: the verification must refute it
constexpr uint32_t kSpecForceNoRetry = 8u;        // no second fused pass: a refuted floor goes straight to the exact kernels (round 5's behaviour)
constexpr uint32_t kSpecForceMispredict2 = 16u;   // predicted floor + 2 (- 2 at the cap): the second pass's floor is wrong too

// The fused CLAHE -> RGB pass (kernels.hip 6a): both DN rasters in, interleaved RGB out.
struct ClaheRgbArgs {
    const uint16_t *in[kMaxBands];
    size_t in_pitch;                    // elements, % 8 == 0
    uint8_t *rgb;
    size_t rgb_pitch_px;                // % 16 == 0
    const Rect *rects;                  // interpolation-cell items (512 columns x rows), line-aligned strips
    int nrects;
    const double *cdfs[kMaxBands];      // [64][256]
    const uint8_t *binlut[kMaxBands];   // DN -> CLAHE bin, constant from win_hi on
    const RowWeight *row_w, *col_w;     // row_w indexed by GLOBAL row
    int32_t row_off;
    const struct ChainBandState *dev_state; // win_hi per band
    struct ChainSpecState *spec;        // spec_ok / floor_pred in, counts and verdict out
    const uint8_t *tables;              // R2[256] | G2[256] | B2[65536] for the predicted floor
    // Saturated bins (all four CDFs 1.0) in extrapolating cells: the reference's level is 254 or 255 by how (1 - d) + d rounds in
    // f64 -- a function of the pixel's row and column alone.  sat_col[c]: which of the (at most three) values T = fl((1 - dx) + dx)
    // column c produces; sat_row[r]: bit k set = a saturated pixel of row r in a column of class k gets level 255 (else 254).
    const uint8_t *sat_col, *sat_row;   // [cols rounded up to the pitch], [rows]; host-built with the plan (api.cpp)
    const float *blue_by_level;         // Pv[256] | Qv[256] (k_chain_predict) or null: the LITE form's blue = rne(Pv[level1] * Qv[level2]), 0 for water
    uint32_t no_verdict;                // a row stripe: the counts are summed over the ranks first, launch_spec_verdict takes the verdict
    uint32_t sat_cols;                  // entries of sat_col
    uint32_t sat_ok;                    // 0: tables absent (the geometry produced more than three classes): such pixels take the exact path
    uint32_t retry;                     // 1: the second pass of a refuted floor (launch_clahe_rgb_fused_retry): runs iff ChainSpecState::retry_armed
};
bool clahe_rgb_fused_supported(const ClaheRgbArgs &a);
hipError_t launch_clahe_rgb_fused(const ClaheRgbArgs &a, int grid /* persistent workgroups: one per CU */, hipStream_t s);
hipError_t launch_clahe_rgb_fused_retry(const ClaheRgbArgs &a, int grid, hipStream_t s); // behind launch_chain_repredict: one kernel, both forms, returns at once unless armed
hipError_t launch_spec_verdict(struct ChainSpecState *spec, const struct ChainBandState *state, hipStream_t s, int second = 0); // the fused pass's verdict from counts that were all-reduced (second: behind the retry's all-reduce)

constexpr size_t kSpecDumpBytes = 256 * 1024;
hipError_t launch_dn_hist_u16(const DnHistArgs &a, int nrects, int nbands, bool vec, hipStream_t s);
hipError_t launch_dn_hist_u16_interior(const DnHistArgs &a, int nrects, int nbands, hipStream_t s);
// untiled histogram (tile_hist[b] = one 65536-bin histogram per band) of a rows x cols raster whose pitch is a multiple of 8
hipError_t launch_dn_hist_u16_linear(const DnHistArgs &a, uint32_t rows, uint32_t cols, int nbands, hipStream_t s);
struct SumTileHistArgs {
    uint32_t *tile_hist[kMaxBands];       // [ntiles][65536]
    unsigned long long *out[kMaxBands];   // [65536]
    uint32_t clear;                       // last reader of the tile histograms: zero what was read (the next scene skips its fill)
};
hipError_t launch_sum_tile_hists(const SumTileHistArgs &a, int ntiles, int nbands, hipStream_t s);
struct TileBinHistArgs {
    uint32_t *tile_hist[kMaxBands];       // [ntiles][65536]
    const uint8_t *binlut[kMaxBands];     // [65536]
    unsigned long long *out[kMaxBands];   // [ntiles][256]
    uint32_t clear;                       // last reader of the tile histograms: zero what was read (the next scene skips its fill)
    double *cdfs_out[kMaxBands];          // [ntiles][256] or null: also clip / redistribute / CDF of the tile (autoscale.rs:271-302) -- the scene's
    uint32_t rows, cols;                  // shape, for the tile areas; only where no reduction over ranks sits between bins and CDFs
};
hipError_t launch_tile_bin_hist(const TileBinHistArgs &a, int ntiles, int nbands, hipStream_t s);
hipError_t launch_clahe_apply_u16(const ClaheApplyArgs &a, int nrects, int nbands, bool vec, bool out16,
                                  hipStream_t s);
bool clahe_apply_spec_ok(const ClaheApplyArgs &a, int nbands);
hipError_t launch_clahe_apply_u8_spec(ClaheApplyArgs a, int nrects, int nbands, hipStream_t s);
constexpr uint32_t kChainLutEntries = 8192; // LDS offset-table capacity when the window is only known on the device
hipError_t launch_lut_apply_u16(const LutApplyArgs &a, bool vec, bool out16, hipStream_t s);
hipError_t launch_compose_u8(const ComposeArgs &a, int vec, hipStream_t s);
struct LutComposeArgs {
    const uint16_t *in[kMaxBands];
    uint8_t *rgb;
    size_t in_pitch, rgb_pitch_px;   // elements / pixels; multiples of 16, 16-byte aligned bases
    uint32_t rows, cols;
    const uint8_t *lut[kMaxBands];   // full 65536-entry DN -> final u8 tables (entry 0 = invalid pixels)
    uint32_t win_hi[kMaxBands];      // the tables are constant from win_hi on
    const uint8_t *tables;           // R2[256] | G2[256] | B2[65536]
    const struct ChainBandState *dev_state; // chain mode: win_hi is read from device memory; lut_cap = LDS capacity per band
    uint32_t lut_cap;
};
bool lut_compose_fits(const LutComposeArgs &a);
hipError_t launch_lut_compose_u16(const LutComposeArgs &a, hipStream_t s);
hipError_t launch_polop_f32(int op, const float *a, const float *b, size_t n, float *out, hipStream_t s);
hipError_t launch_synth_scene_u16(uint64_t seed, int band, const uint16_t *d_q /*[4][65536]*/,
                                  size_t rows_total, size_t cols, size_t row0, size_t rows_local,
                                  uint16_t *d_out, size_t pitch, uint32_t flags, hipStream_t s);
hipError_t launch_remap_u8(uint8_t *buf, size_t pitch, uint32_t rows, uint32_t cols, const uint8_t *d_map256,
                           hipStream_t s);
hipError_t launch_hist256_u8(const uint8_t *in, size_t pitch, uint32_t rows, uint32_t cols,
                             unsigned long long *hist, hipStream_t s);

size_t clahe_apply_lds_bytes(const ClaheApplyArgs &a, int nbands);
// the conflict-free exact kernel with u16 levels out (kernels.hip 4a): persistent 1024-thread workgroups over items of <= 1024 rows
bool clahe_apply_u16_cf_supported(const ClaheApplyArgs &a, int nbands, size_t max_item_rows);
hipError_t launch_clahe_apply_u16_cf(const ClaheApplyArgs &a, const int32_t *first, int nwg, int nbands, hipStream_t s); // workgroup k of a band: items [first[k], first[k + 1]) of a.rects
// more than the default 64 KiB of dynamic LDS for `kernel` on the CURRENT device: once per (device, kernel), thread-safe, checked
hipError_t opt_in_dynamic_lds(const void *kernel);
constexpr uint32_t kLutLdsMaxBytes = 48 * 1024; // per-band window budget in LDS

} // namespace sarpro

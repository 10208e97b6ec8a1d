// host_logic.h -- host half of the MI355X raster core: everything that happens between
// kernels (statistics from device-produced histograms, window selection, LUT / CDF /
// geometry tables).  Pure C++ + glibc libm, no HIP: runs (and is tested) without a GPU.
//
// The device never evaluates log10/pow for integer-DN input: every per-pixel transcendental
// of the reference collapses into tables built here with the host's glibc, exactly as the
// Rust reference evaluates them (f64::log10 / f64::powf / f32::powf lower to libm).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/sarpro_hip.h"

namespace sarpro {

constexpr int kTiles = 8;          // CLAHE tiles per axis (autoscale.rs:593)
constexpr int kClaheBins = 256;    // autoscale.rs:593
constexpr double kClipLimit = 2.0; // autoscale.rs:593
constexpr int kStatBins = 4096;    // autoscale.rs:103

// db[DN] = 10*log10(max(DN,1e-10)) for every u16 DN (pipeline.rs:18-23); db[0] = -100.
const double *db_table_u16();

// 0 = not tamed-synrgb, 1 = co-pol, 2 = cross-pol (autoscale.rs:721-727)
enum TamedSynrgb { kNotTamedSynrgb = 0, kTamedCopol = 1, kTamedCrosspol = 2 };

int stats_from_dn_hist(const uint64_t *dn_hist, sarpro_hip_stats *out);
// stats over arbitrary sorted distinct (value-in-dB, count) support; used by the f32 path
int stats_from_bins4096(uint64_t count, double min_db, double max_db, double mean, double std_db,
                        const uint64_t *hist4096, sarpro_hip_stats *out);
int select_window(sarpro_hip_stats *s, int strategy, int tamed_synrgb);

// level of one dB value under the window (autoscale.rs:440-442 / 649-651 / 734-736)
uint16_t level_of_db(double db, double low_clip, double high_clip, double gamma, double max_val);
// CLAHE bin of one dB value (autoscale.rs:585-586 then 262-265 / 320)
uint8_t clahe_bin_of_db(double db, double low_clip, double high_clip);

// A per-DN table plus the DN window outside which it is constant:
// value(DN) = full[clamp(DN, win_lo, win_hi)] for DN >= 1, and 0 for DN = 0.
struct DnLut {
    std::vector<uint16_t> full; // 65536 entries; full[0] = 0
    uint32_t win_lo = 1, win_hi = 1;
};
void build_level_lut_u16(const sarpro_hip_stats &s, int bit_depth, int tamed_synrgb, DnLut *out);
void build_clahe_bin_lut_u16(const sarpro_hip_stats &s, DnLut *out);

// CLAHE geometry (autoscale.rs:235-236, 307-318)
struct RowWeight { double d, omd; int32_t t0, t1; }; // d = dy (or dx), omd = 1.0 - d
struct ClaheGeometry {
    size_t rows = 0, cols = 0, tile_h = 0, tile_w = 0;
    std::vector<RowWeight> row_w, col_w;
    // cells: maximal row (col) ranges with constant (t0,t1)
    std::vector<size_t> row_cell_start, col_cell_start; // size = ncells+1
};
bool clahe_shape_ok(size_t rows, size_t cols);
// A bin whose four tile CDFs are all 1.0, blended as autoscale.rs:327-329 does: top = bottom = T = fl(fl(1 - dx) + dx), out =
// fl(fl(T (1 - dy)) + fl(T dy)), level = (out clamped to [0, 1]) * 255 truncated -- 255 or, where the roundings fall short of 1.0 (they
// can in the first half tile row / column, where a weight is negative), 254: a function of the pixel's row and column alone.
// col_class[c] in {0, 1, 2} names the value T of column c; bit k of row_bits[r] is set when a pixel of row r whose column is of class
// k gets level 255.  False (tables untouched) when the geometry yields more than three values of T or a level outside {254, 255}.
bool clahe_saturated_levels(const ClaheGeometry &g, std::vector<uint8_t> *col_class, std::vector<uint8_t> *row_bits);
void build_clahe_geometry(size_t rows, size_t cols, ClaheGeometry *g);
void clahe_tile_cdf(uint64_t *hist /*[256], clobbered*/, size_t tile_rows, size_t tile_cols,
                    double *cdf /*[256]*/);
int clahe_cdfs(const uint64_t *tile_hists, size_t rows, size_t cols, double *cdfs_out);

void u8_rescale_lut(unsigned min_level, unsigned max_level, uint8_t *lut256);

// luts = lut_r[256] | lut_g[256] | lut_b[65536]
void synrgb_luts_default(uint8_t *luts);
int synrgb_floor_from_hist(const uint64_t *combined_hist256, uint64_t n_per_band);
void synrgb_luts_suppressed(int floor_with_cushion, uint8_t *luts);

// Compose tables with the per-band u8 rescale folded in:
// R2[v1] = lut_r[resc1[v1]], G2[v2] = lut_g[resc2[v2]],
// B2[v1<<8|v2] = lut_b[resc1[v1]<<8 | resc2[v2]] (with the suppressed water short-circuit
// synthetic_rgb.rs:161-166 folded into all three: when both rescaled values <= floor the
// pixel is (0,0,0); lut_r/lut_g are already 0 there, B2 is forced to 0).
// tables = R2[256] | G2[256] | B2[65536]
void fold_compose_tables(const uint8_t *luts, int floor_with_cushion /* -1: none */,
                         const uint8_t *resc1, const uint8_t *resc2, uint8_t *tables);

int stripe_plan(size_t rows, int nranks, size_t *row0, size_t *nrows);

// constant tables the device-resident chain uploads once (chain_kernels.hip)
const uint8_t *synrgb_supp_rg_tables(); // [41][512]: suppressed lut_r | lut_g for floor_with_cushion = 0..40
const uint8_t *synrgb_blue_pair_supp(); // [256][256]: blue of the suppressed variant per (r, g) pair
const float *synrgb_blue_factors_supp(); // P[256] | Q[256]: that table as rne(P[r] * Q[g]), verified for all pairs; nullptr when the check fails
const uint8_t *synrgb_blue_pair_default(); // [256][256]: blue of the default variant per (r, g) pair
// [3][256] doubles for gamma = 0.8, 0.9, 1.1: thr[k] = smallest x in [0,1] with trunc(clamp(pow(x, gamma) * 255)) >= k
// (autoscale.rs:441-442 / 650-651 at max_val 255); thr[0] = 0.  Found with glibc pow by bisection over doubles.
const double *gamma_level_thresholds_u8();

} // namespace sarpro

// ---------------------------------------------------------------------------------------
// f32-input flavour: every per-pixel decision of the reference is a monotone step function of
// the sample v (v -> max(v,1e-10) -> log10 -> affine -> clamp -> pow -> truncate), so each one
// is captured EXACTLY by a sorted table of f32 thresholds found on the host with glibc's
// log10/pow; the device only compares v against thresholds (no device libm in any decision).
// ---------------------------------------------------------------------------------------
namespace sarpro {
int bin4096_of_f32(float x, double min_db, double max_db);                                             // autoscale.rs:113-115
void build_bin4096_thresholds_range(double min_db, double max_db, int k0, int k1, float *thr_out);     // thresholds k0 .. k1 (1 <= k <= 4095)
double percentile_from_bin(double min_db, double max_db, int bin, uint64_t target, uint64_t cum_below, uint64_t in_bin); // autoscale.rs:120-140
uint64_t percentile_target(uint64_t n, double p);                                                      // autoscale.rs:121-122
double db_of_f32(float v);        // pipeline.rs:19-20
float valid_threshold_f32();      // smallest f32 v with db_of_f32(v) > -50 (pipeline.rs:22)
// thr[k], k = 1..4095: smallest valid f32 whose 4096-bin index (autoscale.rs:113-115) is >= k;
// +inf when no f32 reaches k.  thr[0] is unused (set to 0).
void build_bin4096_thresholds(double min_db, double max_db, float *thr4096);
// thr[k], k = 1..nlevels: smallest valid f32 whose level (autoscale.rs:440-442) is >= k.
void build_level_thresholds(const sarpro_hip_stats &s, int nlevels /*255 or 65535*/, float *thr);
float level_threshold_one(const sarpro_hip_stats &s, int nlevels, int k); // thr[k] of that table alone
// thr[k], k = 1..255: smallest valid f32 whose CLAHE bin (autoscale.rs:585-586, 262-265) is >= k.
void build_clahe_bin_thresholds(const sarpro_hip_stats &s, float *thr256);
} // namespace sarpro

// ---------------------------------------------------------------------------------------
// Lanczos3 resize (resize.rs:32-89 -> fast_image_resize `Convolution(Lanczos3)`; parity with the
// crate is UNPINNED) -- fixed-point coefficient tables for one axis.
// ---------------------------------------------------------------------------------------
namespace sarpro {
struct ResizeCoeffs {
    uint32_t in_size = 0, out_size = 0, window = 0;
    int precision = 0;                 // result = clamp((sum + (1 << (precision-1))) >> precision)
    std::vector<uint32_t> start, size; // [out_size] first input index / tap count
    std::vector<int32_t> k;            // [window][out_size] (tap-major so lanes over outputs coalesce)
};
void build_resize_coeffs(uint32_t in_size, uint32_t out_size, int elem_size /*1: i16 range, 2: i32 range*/, ResizeCoeffs *out);
void resize_bounds(uint32_t in_size, uint32_t out_size, uint32_t *window, std::vector<uint32_t> *start, std::vector<uint32_t> *size); // build_resize_coeffs' windows alone
void resize_dimensions(size_t original_cols, size_t original_rows, size_t target_size, size_t *new_cols, size_t *new_rows); // resize.rs:6-30
} // namespace sarpro

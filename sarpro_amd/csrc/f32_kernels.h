// f32_kernels.h -- launch interface of the f32-input kernels (f32_kernels.hip).
#pragma once
#include "kernels.h"

namespace sarpro {

struct F32Partial { // one per block of the pre-pass
    unsigned long long count;
    double sum, sumsq; // of dB over the valid samples
    float minv, maxv;  // of the valid samples
};

// Where a step table is linear in dB (4096-bin index, CLAHE bin, level with gamma = 1) the kernels first ESTIMATE
// the step from log2(x * inv_x0) * scale + bias (hardware v_log_f32) and then VERIFY it against the two neighbouring
// thresholds; only a failed check (samples within float error of a threshold, or use == 0) takes the binary search.
// The thresholds alone decide the result, so the estimate's quality only affects speed.
struct F32StepEstimate {
    float inv_x0, scale, bias; // t = log2(x * inv_x0) * scale (the clipped window maps to 0..1), step = t^gamma * nsteps + bias
    float gamma, nsteps;
    int use;
};

// Where the samples come from.  op < 0: the f32 raster `in` of the kernel.  op >= 0 (a sarpro_polop value): the samples are
// op(a, b) of two co-registered rasters, computed in registers as ops.rs:4-44 computes them (IEEE f32, correctly rounded
// divide) -- the f32 pol-op raster of io/sentinel1.rs:1501-1578 is never written or read.  u16 = 1: a and b hold u16 DN
// (exact as f32), else f32.
struct F32Pol {
    const void *a = nullptr, *b = nullptr;
    size_t pitch = 0; // elements
    int op = -1;
    int u16 = 0;
};

struct F32LevelArgs {
    const float *in;
    void *out; // u8 or u16
    size_t in_pitch, out_pitch;
    uint32_t rows, cols;
    float t_valid;
    const float *thr;                // [256] (u8) or [65536 + 1] (u16: thr[65536] = +inf sentinel); thr[0] unused
    unsigned long long *level_hist;  // [256], u8 only
    F32StepEstimate est;
    // u16 output: the level evaluated in f64 with the reference's expression (autoscale.rs:437-447 / 647-655) decides
    // unless it lies within 1e-6 of a level boundary (device libm vs glibc: differences of a few ulp, ~1e-10 levels);
    // those samples, a few per scene, are resolved against the exact thresholds.  t_first / t_last are thr[1] and
    // thr[65535]: samples outside them are level 0 / 65535 by definition of the table.
    double low, high, range, gamma, max_val; // range = max(high - low, 1)
    float t_first, t_last;
    int f64_levels;
    F32Pol pol;
};

struct F32TileHistArgs {
    const float *in;
    size_t pitch;
    const Rect *rects;
    float t_valid;
    const float *thr;                // [256]
    unsigned long long *tile_bins;   // [64][256], zeroed by the caller
    F32StepEstimate est;
    F32Pol pol;
};

struct F32ClaheApplyArgs {
    const float *in;
    void *out;
    size_t in_pitch, out_pitch;
    const Rect *rects;
    const double *cdfs;              // [64][256]
    float t_valid;
    const float *thr;                // [256]
    const RowWeight *row_w, *col_w;
    unsigned long long *level_hist;  // [256], u8 only
    double max_val;
    F32StepEstimate est;
    F32Pol pol;
};

int f32_prepass_grid(uint32_t rows, uint32_t cols, bool vec);
// moments = false: count / min / max only (no per-sample f64 log10: the pass is then memory-bound)
hipError_t launch_f32_prepass(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, bool moments,
                              F32Partial *d_partials, int grid, hipStream_t s, const F32Pol &pol = F32Pol());
hipError_t launch_f32_hist4096(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec,
                               const float *d_thr, unsigned long long *d_hist, F32StepEstimate est, hipStream_t s, const F32Pol &pol = F32Pol());
hipError_t launch_f32_level(const F32LevelArgs &a, bool vec, bool out16, hipStream_t s);
hipError_t launch_f32_tile_hist(const F32TileHistArgs &a, int nrects, bool vec, hipStream_t s);
hipError_t launch_f32_clahe_apply(const F32ClaheApplyArgs &a, int nrects, bool vec, bool out16, hipStream_t s);
hipError_t launch_db_mask_f32(const float *in, size_t n, float t_valid, double *db, uint8_t *mask, hipStream_t s);

} // namespace sarpro

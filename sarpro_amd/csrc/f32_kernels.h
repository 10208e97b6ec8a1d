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
    float a_mul, b_add;        // gamma == 1: step = log2(x) * a_mul + b_add, the same expression folded (one v_log, one fma, one clamp)
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
    // gamma == 1 (every strategy but the gamma-corrected ones): the whole expression is one line in log2(x),
    // y = clamp(lin_a * log2(x) + lin_b, 0, lin_ymax), lin_a = 10 log10(2) max_val / range, lin_b = -low max_val / range
    // (folded into the kernel's log2 table and series coefficients: 7 f64 operations instead of 20)
    int lin;
    double lin_a, lin_b, lin_ymax;
    // f64_levels == 2: no 65535-entry table at all -- a sample within 1e-6 of a level boundary is QUEUED (row, column, bits) and
    // the host settles it with the reference's own arithmetic (glibc); thr is then unused.  *uq_count > uq_cap: overflow.
    uint32_t *uq_count;
    uint4 *uq_entries;
    uint32_t uq_cap;
    F32Pol pol;
};
// out[row * pitch + col] = level for n patches (row, col, level, -)
hipError_t launch_patch_u16(uint16_t *out, size_t pitch, const uint4 *d_patches, uint32_t n, hipStream_t s);
hipError_t launch_patch_u8(uint8_t *out, size_t pitch, const uint4 *d_patches, uint32_t n, hipStream_t s);

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
    uint32_t no_spec;                // context attribute NO_SPEC: the f64 blend for every sample also at u8 output (cross-check)
};

// ---- percentiles without the 4096-bin sweep (f32_path.cpp: zone route) ----
// A row sample locates each percentile the strategy reads to within a few buckets of the float's leading 15 bits; the
// min / max pass then also counts the valid samples at or above every zone bound and appends the samples INSIDE a zone to a
// side buffer.  With the scene's min / max known, the few 4096-bin thresholds that fall inside a zone are counted against that
// buffer: the bin of the percentile, the count below it and the count in it come out exact (autoscale.rs:120-140).
constexpr int kMaxZones = 6;
constexpr int kMaxProbes = 2 * kMaxZones;
constexpr int kSampleKeys = 32768; // key = bits >> 16 of a positive f32: 8 exponent + 7 mantissa bits
constexpr int kSubKeys = 512;
constexpr int kZoneLutKeys = 8192; // keys the sweep's class table covers (64 octaves from work->kbase; values outside clamp to its ends)
constexpr int kZoneMaxThr = 1023;  // thresholds (zone bounds included) the count kernel takes: 2^m - 1      // the next 9 mantissa bits: sub-bucket of a probed key
struct F32ZoneWork { // device memory, written by the zone kernels
    uint32_t ns, kmin, kmax, nprobe;       // sample size, lowest / highest populated key
    uint32_t probe_key[kMaxProbes];        // key whose bucket holds the probe rank
    uint32_t probe_base[kMaxProbes];       // sampled values below that bucket
    uint32_t probe_rank[kMaxProbes];
    int32_t nz;                            // zones selected (0: the route steps aside)
    float bounds[2 * kMaxZones];           // lo_0 < hi_0 < lo_1 < hi_1 ...: zone j = [lo_j, hi_j); unused entries +inf
    float mass_est;                        // estimated share of the valid samples the sweep keeps (whole marked buckets)
    uint32_t kbase;                        // first key of the class table
    int32_t zone_run[kMaxZones];           // the run of marked buckets zone j lies in (runs in ascending order; gap g lies below run g)
    int32_t nrun;                          // runs of kept buckets (0: the sweep keeps nothing)
    uint32_t run_s[kMaxZones], run_e[kMaxZones]; // first / last key of each run
};
struct F32ZoneSelectArgs {
    F32ZoneWork *work;
    const uint32_t *key_hist;    // [kSampleKeys]
    uint32_t *sub_hist;          // [kMaxProbes][kSubKeys]
    double pcts[kMaxZones];      // the percentiles the strategy reads
    int npcts;
    float t_valid;
    float max_mass;              // zones heavier than this share of the samples: nz = 0
    float sample_fraction;       // sampled rows / rows (1: the "sample" is the scene, the probe ranks are exact)
    uint8_t *lut;                // [kZoneLutKeys]: the sweep's class table (written after the finalize kernel, from work->run_*)
};
struct F32ZoneArgs {
    const float *in;
    size_t pitch;
    uint32_t rows, cols;
    float t_valid;
    F32Pol pol;
    F32Partial *partials;            // [grid]
    const F32ZoneWork *work;         // nz, bounds, kbase
    const uint32_t *lut;             // [kZoneLutKeys / 4] class bytes, key - kbase: 0x80 | 56 = KEPT (the bucket touches a zone), else 8 x (runs of kept buckets below it)
    unsigned long long *gap_counts;  // [grid][8]: valid samples in the unkept buckets below run g (g = 0 .. 6; entry 7 unused), per workgroup
    float *zone_buf;                 // wave v of workgroup w appends to [w * cap + v * cap / 4, + cap / 4): cap % 4 == 0
    uint32_t cap;
    uint32_t *zone_n;                // [grid * 4]: samples each wave found inside zones (> cap / 4: overflow, the route is abandoned)
    uint32_t no_vec8;                // context attribute F32_NO_VEC8: four samples per lane also for u16 operands (cross-check)
};
// rows r = stride/2, stride/2 + stride, ...: the samples (pol-op applied) stored row by row in `sample` (nsrows x sample_pitch) and
// the histogram of their leading bits
hipError_t launch_f32_sample_keys(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, uint32_t row_stride,
                                  float *d_sample, uint32_t sample_pitch, uint32_t *d_key_hist /* zeroed */, hipStream_t s,
                                  const F32Pol &pol = F32Pol());
hipError_t launch_selftest_div_small_ints(unsigned long long *d_mismatches /* zeroed */, hipStream_t s);
hipError_t launch_f32_zone_pick(const F32ZoneSelectArgs &a, hipStream_t s);
hipError_t launch_f32_sample_sub(const float *d_sample, uint64_t n, float t_valid, const F32ZoneWork *work, uint32_t *d_sub_hist /* zeroed */, hipStream_t s);
hipError_t launch_f32_zone_finalize(const F32ZoneSelectArgs &a, hipStream_t s);
hipError_t launch_f32_prepass_zones(const F32ZoneArgs &a, bool vec, int grid, hipStream_t s);
// Results for the host WITHOUT copy commands: a one-workgroup kernel reduces / copies them into pinned (coherent) host memory and
// raises a sequence word there; the host spins on that word instead of sleeping in hipStreamSynchronize.  (Four small copies and
// a stream wait cost a turn 80 us; this costs it ~15.)
struct F32ZoneMail { // what the host reads after the min / max pass of the zone route
    float min_v, max_v;                        // +inf / -inf when there are no valid samples
    unsigned long long gap[8];                 // valid samples in the unkept buckets below run g
    unsigned long long kept;                   // samples in the side buffers
    uint32_t overflow, pad;                    // a wave's side buffer overflowed
    F32ZoneWork work;
};
hipError_t launch_f32_zone_post(const F32ZoneArgs &a, int grid, F32ZoneMail *mail, uint32_t *flag, uint32_t seq, hipStream_t s);
struct PostSegs { const void *src[4]; void *dst[4]; uint32_t bytes[4]; int n; }; // bytes: multiples of 4
hipError_t launch_post(const PostSegs &segs, uint32_t *flag, uint32_t seq, hipStream_t s);
// host words -> device (the source is pinned coherent memory the device reads directly) and zero fills, one launch: a kernel
// follows a kernel without the 5-17 us a copy / fill command puts between itself and the next kernel
struct PrepSegs { const void *src[3]; void *dst[3]; uint32_t bytes[3]; int n; void *zero[4]; uint32_t zbytes[4]; int nz; }; // bytes: multiples of 4
hipError_t launch_prep(const PrepSegs &segs, hipStream_t s);
// counts[i] += zone samples x with thr[i] <= x < thr[i + 1], i = 0 .. nthr (thr[0] = -inf implied below thr[1]; thr sorted, nthr <= kZoneMaxThr)
hipError_t launch_f32_zone_count(const float *zone_buf, const uint32_t *zone_n, uint32_t cap, int nregions, const float *d_thr, int nthr,
                                 unsigned long long *d_counts /* [kZoneMaxThr + 1], zeroed */, hipStream_t s);

int f32_prepass_grid(uint32_t rows, uint32_t cols, bool vec);
int f32_zone_grid(uint32_t rows, uint32_t cols, bool vec); // grid of launch_f32_prepass_zones (<= 2048)
// moments = false: count / min / max only (no per-sample f64 log10: the pass is then memory-bound)
hipError_t launch_f32_prepass(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, bool moments,
                              F32Partial *d_partials, int grid, hipStream_t s, const F32Pol &pol = F32Pol());
hipError_t launch_f32_hist4096(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec,
                               const float *d_thr, unsigned long long *d_hist, F32StepEstimate est, hipStream_t s, const F32Pol &pol = F32Pol());
// the 4096 bins without host-built thresholds: min / max from the pre-pass partials (on the device), f64 binning with a margin,
// samples within 1e-6 of a bin boundary queued as (row, column, bits) for the host (f32_kernels.hip b')
hipError_t launch_f32_hist4096_direct(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, const F32Partial *d_partials,
                                      int nparts, unsigned long long *d_hist /* zeroed */, uint32_t *d_uq_count /* zeroed */, uint4 *d_uq_entries,
                                      uint32_t uq_cap, hipStream_t s, const F32Pol &pol = F32Pol());
hipError_t launch_f32_level(const F32LevelArgs &a, bool vec, bool out16, hipStream_t s);
hipError_t launch_f32_tile_hist(const F32TileHistArgs &a, int nrects, bool vec, hipStream_t s);
hipError_t launch_f32_clahe_apply(const F32ClaheApplyArgs &a, int nrects, bool vec, bool out16, hipStream_t s);
hipError_t launch_db_mask_f32(const float *in, size_t n, float t_valid, double *db, uint8_t *mask, hipStream_t s);

} // namespace sarpro

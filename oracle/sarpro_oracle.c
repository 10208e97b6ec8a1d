/*
 * sarpro_oracle.c -- CPU ORACLE for the sarpro per-pixel raster core.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (sarpro_amd/, the
 * C-ABI library, bench.py's GPU legs) may import, link or execute this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use
 * it, and only as the checker / the timed CPU baseline.
 *
 * It is a literal, single-threaded restatement (plain C, IEEE-754, glibc libm)
 * of the reference's src/core/processing/{pipeline,autoscale,ops,
 * synthetic_rgb,padding,resize}.rs.  Every function cites the reference
 * file:line it follows.  The loops are kept in the reference's own order and
 * the arithmetic in the reference's own op order and precision, so results are
 * what the Rust code yields on Linux/glibc (Rust's f64::log10 / f64::powf /
 * f32::powf lower to libm log10 / pow / powf).
 *
 * PARITY UNPINNED: the reference has no tests, no golden vectors and no
 * known-answer fixtures for this path (Cargo.toml:50 empty dev-dependencies,
 * no #[test] in src/), and it cannot be compiled here (no cargo/rustc in the
 * image, no GDAL).  The oracle is pinned only by (i) being a literal
 * restatement and (ii) hand-derived known answers
 * checked in tests/test_oracle_kat.py.
 *
 * Rust semantics spelled out (helpers below):
 *   - `x as u8/u16/u32/usize/isize` from float: truncate toward zero,
 *     saturate, NaN -> 0.
 *   - f32/f64 `round()`: half away from zero (C round()/roundf()).
 *   - `f64::max/min`: ignore NaN (C fmax()/fmin()).
 *   - `clamp(lo,hi)`: `if x<lo {lo} else if x>hi {hi} else {x}`; NaN stays NaN.
 *   - usize subtraction that would underflow (CLAHE tiny images) panics in
 *     debug and wraps in release: the oracle refuses those shapes
 *     (ORACLE_ERR_UNSUPPORTED_SHAPE).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORACLE_OK 0
#define ORACLE_ERR_INVALID_ARG (-1)
#define ORACLE_ERR_UNSUPPORTED_SHAPE (-3)
#define ORACLE_ERR_OOM (-6)

/* types.rs:115-123 declaration order */
enum { STRAT_STANDARD = 0, STRAT_ROBUST, STRAT_ADAPTIVE, STRAT_EQUALIZED,
       STRAT_CLAHE, STRAT_TAMED, STRAT_DEFAULT };
/* types.rs:170-173 */
enum { BITDEPTH_U8 = 0, BITDEPTH_U16 = 1 };
/* types.rs:8-14 */
enum { OP_SUM = 0, OP_DIFF, OP_RATIO, OP_NDIFF, OP_LOGRATIO };

/* autoscale.rs:7-24 plus the branch parameters the reference only logs */
typedef struct {
    uint64_t valid_count;
    double min_db, max_db, mean_db, std_db, median_db;
    double p01, p02, p05, p10, p25, p75, p90, p95, p98, p99;
    double low_clip, high_clip, gamma; /* chosen window (autoscale.rs:404-429, 491-564) */
    /* Adaptive decision margins (autoscale.rs:503-519) so tests can see how far a
       branch was from flipping; 0 for other strategies */
    double skew_factor, tail_heaviness;
} oracle_stats;

/* ---------- Rust cast / clamp semantics ---------- */
static inline double clamp_f64(double x, double lo, double hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x; /* NaN propagates */
}
static inline float clamp_f32(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
static inline uint64_t f64_as_u64(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}
static inline int64_t f64_as_i64(double x) {
    if (!(x == x)) return 0;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    return (int64_t)x;
}
static inline uint32_t f64_as_u32(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}
static inline uint16_t f64_as_u16(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 65535.0) return 65535;
    return (uint16_t)x;
}
static inline uint8_t f64_as_u8(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}
static inline uint8_t f32_as_u8(float x) {
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

/* ====================================================================== */
/* pipeline.rs:8-40  process_scalar_data_inplace                           */
/* ====================================================================== */
int sarpro_oracle_db_mask_f32(const float *src, size_t n, double *db, uint8_t *mask) {
    if ((!src || !db || !mask) && n) return ORACLE_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        double magnitude = fmax((double)src[i], 1e-10); /* :19 */
        double db_val = 10.0 * log10(magnitude);        /* :20 */
        db[i] = db_val;
        mask[i] = db_val > -50.0;                       /* :22 */
    }
    return ORACLE_OK;
}

/* ====================================================================== */
/* autoscale.rs:35-160  compute_histogram_stats                            */
/* ====================================================================== */
#define NUM_BINS 4096

static double estimate_percentile(const uint64_t *hist, uint64_t n, double min_db,
                                  double max_db, double span, double p) {
    /* autoscale.rs:120-140 */
    uint64_t target = f64_as_u64(floor(p * (double)n));
    if (target >= n) target = n - 1;
    uint64_t cumsum = 0;
    for (size_t b = 0; b < NUM_BINS; ++b) {
        uint64_t h = hist[b];
        uint64_t next = cumsum + h;
        if (target < next) {
            uint64_t within = target >= cumsum ? target - cumsum : 0;
            double frac = h > 0 ? (double)within / (double)h : 0.0;
            double bin_width = span / (double)NUM_BINS;
            double bin_start = min_db + (double)b * bin_width;
            return bin_start + frac * bin_width;
        }
        cumsum = next;
    }
    return max_db;
}

int sarpro_oracle_stats(const double *db, const uint8_t *mask, size_t n, oracle_stats *s) {
    if (!s || ((!db || !mask) && n)) return ORACLE_ERR_INVALID_ARG;
    memset(s, 0, sizeof(*s));
    uint64_t count = 0;
    double min_db = INFINITY, max_db = -INFINITY, mean = 0.0, m2 = 0.0;
    for (size_t i = 0; i < n; ++i) { /* :43-55, row-major order */
        if (mask[i]) {
            double v = db[i];
            count += 1;
            if (v < min_db) min_db = v;
            if (v > max_db) max_db = v;
            double delta = v - mean;
            mean += delta / (double)count;
            double delta2 = v - mean;
            m2 += delta * delta2;
        }
    }
    if (count == 0) return ORACLE_OK; /* :57-76 all zeros */

    double std_db = count > 1 ? sqrt(m2 / (double)count) : 0.0; /* :78 */
    s->valid_count = count;
    s->min_db = min_db;
    s->max_db = max_db;
    s->mean_db = mean;
    s->std_db = std_db;

    if (fabs(max_db - min_db) < DBL_EPSILON) { /* :81-100 */
        s->median_db = s->p01 = s->p02 = s->p05 = s->p10 = s->p25 = min_db;
        s->p75 = s->p90 = s->p95 = s->p98 = s->p99 = max_db;
        return ORACLE_OK;
    }

    uint64_t *hist = (uint64_t *)calloc(NUM_BINS, sizeof(uint64_t));
    if (!hist) return ORACLE_ERR_OOM;
    double span = max_db - min_db;
    double inv_span = 1.0 / span;
    for (size_t i = 0; i < n; ++i) { /* :108-117 */
        if (!mask[i]) continue;
        double t = clamp_f64((db[i] - min_db) * inv_span, 0.0, 1.0);
        uint64_t idx = f64_as_u64(t * (double)NUM_BINS);
        if (idx >= NUM_BINS) idx = NUM_BINS - 1;
        hist[idx] += 1;
    }
    s->median_db = estimate_percentile(hist, count, min_db, max_db, span, 0.5);
    s->p01 = estimate_percentile(hist, count, min_db, max_db, span, 0.01);
    s->p02 = estimate_percentile(hist, count, min_db, max_db, span, 0.02);
    s->p05 = estimate_percentile(hist, count, min_db, max_db, span, 0.05);
    s->p10 = estimate_percentile(hist, count, min_db, max_db, span, 0.10);
    s->p25 = estimate_percentile(hist, count, min_db, max_db, span, 0.25);
    s->p75 = estimate_percentile(hist, count, min_db, max_db, span, 0.75);
    s->p90 = estimate_percentile(hist, count, min_db, max_db, span, 0.90);
    s->p95 = estimate_percentile(hist, count, min_db, max_db, span, 0.95);
    s->p98 = estimate_percentile(hist, count, min_db, max_db, span, 0.98);
    s->p99 = estimate_percentile(hist, count, min_db, max_db, span, 0.99);
    free(hist);
    return ORACLE_OK;
}

/* ====================================================================== */
/* autoscale.rs:220-345  clahe_equalize_normalized                         */
/* ====================================================================== */

/* Shapes for which `r1 - r0` (autoscale.rs:250) or `c1 - c0` (:254) underflows. */
int sarpro_oracle_clahe_shape_ok(size_t rows, size_t cols, size_t tiles_x, size_t tiles_y) {
    if (rows == 0 || cols == 0 || tiles_x == 0 || tiles_y == 0) return 1; /* early clone path */
    size_t tile_h = (rows + tiles_y - 1) / tiles_y;
    size_t tile_w = (cols + tiles_x - 1) / tiles_x;
    for (size_t ty = 0; ty < tiles_y; ++ty)
        if (ty * tile_h > rows) return 0;
    for (size_t tx = 0; tx < tiles_x; ++tx)
        if (tx * tile_w > cols) return 0;
    return 1;
}

/* autoscale.rs:271-302: clip, redistribute, CDF of ONE tile.  hist is modified. */
void sarpro_oracle_clahe_tile_cdf(uint32_t *hist, size_t num_bins, size_t tile_rows,
                                  size_t tile_cols, double clip_limit, double *cdf) {
    double avg = (double)(tile_rows * tile_cols) / (double)num_bins; /* :242-245 */
    double clip_threshold = fmax(clip_limit * avg, 1.0);             /* :273 */
    double excess = 0.0;
    for (size_t i = 0; i < num_bins; ++i) { /* :275-280 */
        if ((double)hist[i] > clip_threshold) {
            excess += (double)hist[i] - clip_threshold;
            hist[i] = f64_as_u32(clip_threshold);
        }
    }
    double add_per_bin = floor(excess / (double)num_bins); /* :282 */
    uint64_t remainder = f64_as_u64(round(excess - add_per_bin * (double)num_bins)); /* :283 */
    for (size_t i = 0; i < num_bins; ++i) /* :284-286 */
        hist[i] = f64_as_u32((double)hist[i] + add_per_bin);
    size_t b = 0;
    while (remainder > 0) { /* :287-292 */
        hist[b] += 1;
        b = (b + 1) % num_bins;
        remainder -= 1;
    }
    double total = 0.0; /* :295 iter().sum::<f64>() folds from 0.0 left to right */
    for (size_t i = 0; i < num_bins; ++i) total += (double)hist[i];
    total = fmax(total, 1.0);
    double acc = 0.0;
    for (size_t i = 0; i < num_bins; ++i) { /* :298-301 */
        acc += (double)hist[i];
        cdf[i] = clamp_f64(acc / total, 0.0, 1.0);
    }
}

/* Returns ORACLE_OK and fills `out` (rows*cols f64).  cdfs_out (optional) gets the
   tiles_y*tiles_x*num_bins CDF table for stage-level tests. */
int sarpro_oracle_clahe(const double *norm, const uint8_t *mask, size_t rows, size_t cols,
                        size_t tiles_x, size_t tiles_y, double clip_limit, size_t num_bins,
                        double *out, double *cdfs_out) {
    if (rows == 0 || cols == 0 || tiles_x == 0 || tiles_y == 0 || num_bins < 2) { /* :231-233 */
        if (rows != 0 && cols != 0) memcpy(out, norm, rows * cols * sizeof(double));
        return ORACLE_OK;
    }
    if (!sarpro_oracle_clahe_shape_ok(rows, cols, tiles_x, tiles_y))
        return ORACLE_ERR_UNSUPPORTED_SHAPE;

    size_t tile_h = (rows + tiles_y - 1) / tiles_y; /* :235 */
    size_t tile_w = (cols + tiles_x - 1) / tiles_x; /* :236 */
    double *cdfs = (double *)calloc(tiles_x * tiles_y * num_bins, sizeof(double));
    uint32_t *hist = (uint32_t *)malloc(num_bins * sizeof(uint32_t));
    if (!cdfs || !hist) { free(cdfs); free(hist); return ORACLE_ERR_OOM; }

    for (size_t ty = 0; ty < tiles_y; ++ty) { /* :247-304 */
        size_t r0 = ty * tile_h;
        size_t r1 = (ty + 1) * tile_h < rows ? (ty + 1) * tile_h : rows;
        size_t tile_rows = r1 - r0;
        for (size_t tx = 0; tx < tiles_x; ++tx) {
            size_t c0 = tx * tile_w;
            size_t c1 = (tx + 1) * tile_w < cols ? (tx + 1) * tile_w : cols;
            size_t tile_cols = c1 - c0;
            memset(hist, 0, num_bins * sizeof(uint32_t));
            for (size_t r = r0; r < r1; ++r) {
                for (size_t c = c0; c < c1; ++c) {
                    if (mask[r * cols + c]) { /* :261-267 */
                        double v = clamp_f64(norm[r * cols + c], 0.0, 1.0);
                        int64_t bin = f64_as_i64(round(v * ((double)num_bins - 1.0)));
                        if (bin < 0) bin = 0;
                        if ((size_t)bin >= num_bins) bin = (int64_t)num_bins - 1;
                        hist[bin] += 1;
                    }
                }
            }
            sarpro_oracle_clahe_tile_cdf(hist, num_bins, tile_rows, tile_cols, clip_limit,
                                         cdfs + (ty * tiles_x + tx) * num_bins);
        }
    }

    for (size_t r = 0; r < rows; ++r) { /* :332-342 with sample_cdf :307-330 */
        for (size_t c = 0; c < cols; ++c) {
            if (!mask[r * cols + c]) { out[r * cols + c] = 0.0; continue; }
            double val = norm[r * cols + c];
            double rf = (double)r / (double)tile_h - 0.5;
            double cf = (double)c / (double)tile_w - 0.5;
            int64_t ty = f64_as_i64(fmax(floor(rf), 0.0));
            int64_t tx = f64_as_i64(fmax(floor(cf), 0.0));
            double dy = rf - (double)ty;
            double dx = cf - (double)tx;
            int64_t tyl = (int64_t)tiles_y - 1, txl = (int64_t)tiles_x - 1;
            size_t ty0 = (size_t)(ty < 0 ? 0 : (ty > tyl ? tyl : ty));
            size_t tx0 = (size_t)(tx < 0 ? 0 : (tx > txl ? txl : tx));
            size_t ty1 = (size_t)(ty + 1 < 0 ? 0 : (ty + 1 > tyl ? tyl : ty + 1));
            size_t tx1 = (size_t)(tx + 1 < 0 ? 0 : (tx + 1 > txl ? txl : tx + 1));
            size_t bin_pos = (size_t)f64_as_u64(round(clamp_f64(val, 0.0, 1.0) * ((double)num_bins - 1.0)));
            double cdf00 = cdfs[(ty0 * tiles_x + tx0) * num_bins + bin_pos];
            double cdf01 = cdfs[(ty0 * tiles_x + tx1) * num_bins + bin_pos];
            double cdf10 = cdfs[(ty1 * tiles_x + tx0) * num_bins + bin_pos];
            double cdf11 = cdfs[(ty1 * tiles_x + tx1) * num_bins + bin_pos];
            double top = cdf00 * (1.0 - dx) + cdf01 * dx;
            double bottom = cdf10 * (1.0 - dx) + cdf11 * dx;
            out[r * cols + c] = top * (1.0 - dy) + bottom * dy;
        }
    }
    if (cdfs_out) memcpy(cdfs_out, cdfs, tiles_x * tiles_y * num_bins * sizeof(double));
    free(cdfs);
    free(hist);
    return ORACLE_OK;
}

/* ====================================================================== */
/* autoscale.rs:348-364  scale_u16_to_u8                                   */
/* ====================================================================== */
int sarpro_oracle_scale_u16_to_u8(const uint16_t *data, size_t n, uint8_t *out) {
    if (n == 0) return ORACLE_OK;
    uint16_t mn = data[0], mx = data[0];
    for (size_t i = 1; i < n; ++i) { if (data[i] < mn) mn = data[i]; if (data[i] > mx) mx = data[i]; }
    float min = (float)mn, max = (float)mx;
    float scale = max > min ? 255.0f / (max - min) : 1.0f; /* :356 */
    for (size_t i = 0; i < n; ++i) {
        float val = roundf(((float)data[i] - min) * scale); /* :360 */
        out[i] = f32_as_u8(clamp_f32(val, 0.0f, 255.0f));
    }
    return ORACLE_OK;
}

/* shared map loop: autoscale.rs:437-447 and :647-655 */
static void map_window(const double *db, const uint8_t *mask, size_t n, double low_clip,
                       double high_clip, double range, double gamma, double max_val,
                       uint16_t *out) {
    for (size_t i = 0; i < n; ++i) {
        if (mask[i]) {
            double clipped = fmin(fmax(db[i], low_clip), high_clip);
            double normalized = pow((clipped - low_clip) / range, gamma);
            out[i] = f64_as_u16(clamp_f64(normalized * max_val, 0.0, max_val));
        } else {
            out[i] = 0;
        }
    }
}

/* ====================================================================== */
/* autoscale.rs:368-448  autoscale_db_image (Standard)                     */
/* ====================================================================== */
int sarpro_oracle_autoscale_db_image(const double *db, const uint8_t *mask, size_t rows,
                                     size_t cols, int bit_depth, uint16_t *out,
                                     oracle_stats *stats_out) {
    size_t n = rows * cols;
    oracle_stats s;
    int rc = sarpro_oracle_stats(db, mask, n, &s);
    if (rc) return rc;
    if (s.valid_count == 0) { /* :376-378 */
        memset(out, 0, n * sizeof(uint16_t));
        if (stats_out) *stats_out = s;
        return ORACLE_OK;
    }
    double max_val = bit_depth == BITDEPTH_U8 ? 255.0 : 65535.0;
    double dynamic_range = s.max_db - s.min_db;
    double iqr = s.p75 - s.p25;
    double low_clip, high_clip, gamma;
    if (dynamic_range < 15.0) { /* :404-408 */
        double range = fmax(20.0, dynamic_range * 0.8);
        low_clip = s.median_db - range / 2.0;
        high_clip = s.median_db + range / 2.0;
        gamma = 1.1;
    } else if (iqr < 5.0) { /* :409-413 */
        double outlier_factor = 2.5;
        low_clip = s.p25 - outlier_factor * iqr;
        high_clip = s.p75 + outlier_factor * iqr;
        gamma = 1.0;
    } else if (dynamic_range > 40.0) { /* :414-419 */
        low_clip = fmax(s.p02, s.min_db + 0.02 * dynamic_range);
        high_clip = fmin(s.p98, s.max_db - 0.02 * dynamic_range);
        gamma = 0.9;
    } else { /* :420-424 */
        low_clip = s.p02;
        high_clip = s.p98;
        gamma = 1.0;
    }
    low_clip = fmax(low_clip, s.min_db);   /* :427 */
    high_clip = fmin(high_clip, s.max_db); /* :428 */
    double range = fmax(high_clip - low_clip, 1.0); /* :429 */
    s.low_clip = low_clip; s.high_clip = high_clip; s.gamma = gamma;
    map_window(db, mask, n, low_clip, high_clip, range, gamma, max_val, out);
    if (stats_out) *stats_out = s;
    return ORACLE_OK;
}

/* ====================================================================== */
/* autoscale.rs:452-659  autoscale_db_image_advanced                       */
/* (the use_local_enhancement arm :613-643 is dead: the flag is false in   */
/*  every strategy arm, so it is not restated)                             */
/* ====================================================================== */
int sarpro_oracle_autoscale_db_image_advanced(const double *db, const uint8_t *mask,
                                              size_t rows, size_t cols, int bit_depth,
                                              int strategy, uint16_t *out,
                                              oracle_stats *stats_out) {
    size_t n = rows * cols;
    double max_val = bit_depth == BITDEPTH_U8 ? 255.0 : 65535.0;
    oracle_stats s;
    int rc = sarpro_oracle_stats(db, mask, n, &s);
    if (rc) return rc;
    if (s.valid_count == 0) { /* :466-468 */
        memset(out, 0, n * sizeof(uint16_t));
        if (stats_out) *stats_out = s;
        return ORACLE_OK;
    }
    double iqr = s.p75 - s.p25;
    double low_clip, high_clip, gamma;
    switch (strategy) {
    case STRAT_ROBUST: { /* :492-499 */
        double outlier_threshold = 2.5 * iqr;
        low_clip = fmax(fmax(s.p25 - outlier_threshold, s.p01), s.min_db);
        high_clip = fmin(fmin(s.p75 + outlier_threshold, s.p99), s.max_db);
        gamma = 1.0;
        break;
    }
    case STRAT_ADAPTIVE: { /* :500-538 */
        double skew_factor = (s.mean_db - s.median_db) / fmax(fabs(s.std_db), 1.0);
        double tail_heaviness = (s.p99 - s.p95) / fmax(s.p95 - s.p75, 1.0);
        s.skew_factor = skew_factor;
        s.tail_heaviness = tail_heaviness;
        if (fabs(skew_factor) > 0.5) {
            if (skew_factor > 0.0) { low_clip = s.p02; high_clip = s.p98; gamma = 0.9; }
            else { low_clip = s.p05; high_clip = s.p95; gamma = 1.1; }
        } else if (tail_heaviness > 2.0) {
            low_clip = s.p10; high_clip = s.p90; gamma = 0.8;
        } else {
            low_clip = s.p05; high_clip = s.p95; gamma = 1.0;
        }
        break;
    }
    case STRAT_EQUALIZED: /* :539-543 */
    case STRAT_CLAHE:     /* :544-548 */
        low_clip = s.p01; high_clip = s.p99; gamma = 1.0; break;
    case STRAT_TAMED: /* :549-553 */
        low_clip = s.p25; high_clip = s.p99; gamma = 1.0; break;
    case STRAT_STANDARD: /* :554-557 */
    case STRAT_DEFAULT:  /* :558-561 */
        low_clip = s.p05; high_clip = s.p95; gamma = 1.0; break;
    default:
        return ORACLE_ERR_INVALID_ARG;
    }
    double range = fmax(high_clip - low_clip, 1.0); /* :564 */
    s.low_clip = low_clip; s.high_clip = high_clip; s.gamma = gamma;
    if (stats_out) *stats_out = s;

    if (strategy == STRAT_CLAHE) { /* :572-608 */
        if (!sarpro_oracle_clahe_shape_ok(rows, cols, 8, 8)) return ORACLE_ERR_UNSUPPORTED_SHAPE;
        double *norm = (double *)malloc((n ? n : 1) * sizeof(double));
        double *eq = (double *)malloc((n ? n : 1) * sizeof(double));
        if (!norm || !eq) { free(norm); free(eq); return ORACLE_ERR_OOM; }
        for (size_t i = 0; i < n; ++i) { /* :583-591 */
            if (mask[i]) {
                double clipped = fmin(fmax(db[i], low_clip), high_clip);
                norm[i] = (clipped - low_clip) / range;
            } else {
                norm[i] = 0.0;
            }
        }
        rc = sarpro_oracle_clahe(norm, mask, rows, cols, 8, 8, 2.0, 256, eq, NULL); /* :593 */
        if (rc == ORACLE_OK) {
            for (size_t i = 0; i < n; ++i) /* :600-606 */
                out[i] = mask[i] ? f64_as_u16(clamp_f64(eq[i], 0.0, 1.0) * max_val) : 0;
        }
        free(norm);
        free(eq);
        return rc;
    }
    map_window(db, mask, n, low_clip, high_clip, range, gamma, max_val, out); /* :647-655 */
    return ORACLE_OK;
}

/* ====================================================================== */
/* autoscale.rs:710-742  autoscale_db_image_tamed_synrgb_u8                */
/* ====================================================================== */
int sarpro_oracle_tamed_synrgb_u8(const double *db, const uint8_t *mask, size_t rows,
                                  size_t cols, int is_copol, uint8_t *out) {
    size_t n = rows * cols;
    oracle_stats s;
    int rc = sarpro_oracle_stats(db, mask, n, &s);
    if (rc) return rc;
    if (s.valid_count == 0) { memset(out, 0, n); return ORACLE_OK; }
    double low_clip = is_copol ? fmin(s.p02, s.p05) : s.p05; /* :721-727 */
    double high_clip = s.p99;
    double range = fmax(high_clip - low_clip, 1.0);
    for (size_t i = 0; i < n; ++i) {
        if (mask[i]) {
            double clipped = fmin(fmax(db[i], low_clip), high_clip);
            double normalized = (clipped - low_clip) / range;
            out[i] = f64_as_u8(clamp_f64(normalized * 255.0, 0.0, 255.0));
        } else {
            out[i] = 0;
        }
    }
    return ORACLE_OK;
}

/* ====================================================================== */
/* pipeline.rs:42-67  process_scalar_data_pipeline                         */
/* + autoscale.rs:662-704 bit-depth wrappers                               */
/* U8  -> out_u8 filled (out_u16 untouched);  U16 -> out_u16 filled.       */
/* db_out / mask_out optional (the reference returns them).                */
/* ====================================================================== */
int sarpro_oracle_pipeline_f32(const float *in, size_t rows, size_t cols, int bit_depth,
                               int strategy, uint8_t *out_u8, uint16_t *out_u16,
                               double *db_out, uint8_t *mask_out, oracle_stats *stats_out) {
    size_t n = rows * cols;
    if (bit_depth != BITDEPTH_U8 && bit_depth != BITDEPTH_U16) return ORACLE_ERR_INVALID_ARG;
    if (strategy < STRAT_STANDARD || strategy > STRAT_DEFAULT) return ORACLE_ERR_INVALID_ARG;
    double *db = db_out ? db_out : (double *)malloc((n ? n : 1) * sizeof(double));
    uint8_t *mask = mask_out ? mask_out : (uint8_t *)malloc(n ? n : 1);
    uint16_t *v = bit_depth == BITDEPTH_U16 ? out_u16 : (uint16_t *)malloc((n ? n : 1) * sizeof(uint16_t));
    int rc = ORACLE_ERR_OOM;
    if (db && mask && v) {
        rc = sarpro_oracle_db_mask_f32(in, n, db, mask);
        if (rc == ORACLE_OK) {
            if (strategy == STRAT_STANDARD) /* pipeline.rs:49-52 */
                rc = sarpro_oracle_autoscale_db_image(db, mask, rows, cols, bit_depth, v, stats_out);
            else /* :53-64 */
                rc = sarpro_oracle_autoscale_db_image_advanced(db, mask, rows, cols, bit_depth,
                                                               strategy, v, stats_out);
        }
        if (rc == ORACLE_OK && bit_depth == BITDEPTH_U8)
            rc = sarpro_oracle_scale_u16_to_u8(v, n, out_u8); /* autoscale.rs:670,693 */
    }
    if (!db_out) free(db);
    if (!mask_out) free(mask);
    if (bit_depth != BITDEPTH_U16) free(v);
    return rc;
}

/* ====================================================================== */
/* ops.rs:4-44                                                             */
/* ====================================================================== */
int sarpro_oracle_polop_f32(int op, const float *a, const float *b, size_t n, float *out) {
    switch (op) {
    case OP_SUM: for (size_t i = 0; i < n; ++i) out[i] = a[i] + b[i]; break;   /* :4 */
    case OP_DIFF: for (size_t i = 0; i < n; ++i) out[i] = a[i] - b[i]; break;  /* :7 */
    case OP_RATIO:     /* :10-19 */
    case OP_LOGRATIO:  /* :35-44 (same body) */
        for (size_t i = 0; i < n; ++i) out[i] = fabsf(b[i]) > 1e-10f ? a[i] / b[i] : 0.0f;
        break;
    case OP_NDIFF: /* :22-32 */
        for (size_t i = 0; i < n; ++i) {
            float denom = a[i] + b[i];
            out[i] = fabsf(denom) > 1e-10f ? (a[i] - b[i]) / denom : 0.0f;
        }
        break;
    default: return ORACLE_ERR_INVALID_ARG;
    }
    return ORACLE_OK;
}

/* ====================================================================== */
/* synthetic_rgb.rs:10-67  create_synthetic_rgb                            */
/* luts_out optional: lut_r[256] | lut_g[256] | lut_b[65536] (66048 bytes) */
/* ====================================================================== */
int sarpro_oracle_synrgb_default(const uint8_t *b1, const uint8_t *b2, size_t n, uint8_t *rgb,
                                 uint8_t *luts_out) {
    const float GAMMA_R = 0.7f, GAMMA_G = 0.9f, GAMMA_B = 0.1f, SCALE_255 = 255.0f, BLUE_SCALE = 0.24f;
    uint8_t lut_r[256], lut_g[256];
    uint8_t *lut_b = (uint8_t *)malloc(65536);
    if (!lut_b) return ORACLE_ERR_OOM;
    for (unsigned v = 0; v <= 255; ++v) { /* :22-29 */
        float vf = (float)v / SCALE_255;
        lut_r[v] = f32_as_u8(clamp_f32(roundf(powf(vf, GAMMA_R) * SCALE_255), 0.0f, 255.0f));
        lut_g[v] = f32_as_u8(clamp_f32(roundf(powf(vf, GAMMA_G) * SCALE_255), 0.0f, 255.0f));
    }
    for (unsigned x1 = 0; x1 <= 255; ++x1) { /* :35-52 */
        for (unsigned x2 = 0; x2 <= 255; ++x2) {
            uint8_t blue;
            if (x2 == 0) {
                blue = 0;
            } else {
                float r = (float)lut_r[x1], g = (float)lut_g[x2];
                float ratio = r / g;
                blue = f32_as_u8(roundf(clamp_f32(powf(ratio, GAMMA_B) * SCALE_255 * BLUE_SCALE, 0.0f, 255.0f)));
            }
            lut_b[(x1 << 8) | x2] = blue;
        }
    }
    for (size_t i = 0; i < n; ++i) { /* :55-64 */
        unsigned v1 = b1[i], v2 = b2[i];
        rgb[3 * i + 0] = lut_r[v1];
        rgb[3 * i + 1] = lut_g[v2];
        rgb[3 * i + 2] = lut_b[(v1 << 8) | v2];
    }
    if (luts_out) { memcpy(luts_out, lut_r, 256); memcpy(luts_out + 256, lut_g, 256); memcpy(luts_out + 512, lut_b, 65536); }
    free(lut_b);
    return ORACLE_OK;
}

/* ====================================================================== */
/* synthetic_rgb.rs:88-178  create_synthetic_rgb_suppressed                */
/* floor_out optional: floor_with_cushion                                  */
/* ====================================================================== */
int sarpro_oracle_synrgb_suppressed(const uint8_t *b1, const uint8_t *b2, size_t n, uint8_t *rgb,
                                    uint8_t *luts_out, int *floor_out) {
    uint32_t histogram[256];
    memset(histogram, 0, sizeof(histogram));
    for (size_t i = 0; i < n; ++i) if (histogram[b1[i]] != UINT32_MAX) histogram[b1[i]] += 1; /* :93-95 */
    for (size_t i = 0; i < n; ++i) if (histogram[b2[i]] != UINT32_MAX) histogram[b2[i]] += 1; /* :96-98 */
    uint32_t total_count = (uint32_t)(n + n); /* :99 `as u32` wraps */
    uint32_t target_count = f64_as_u32(round((double)total_count * 0.05)); /* :100 */
    uint32_t cumulative = 0;
    size_t floor_value = 0;
    for (size_t i = 0; i <= 255; ++i) { /* :103-109 */
        uint64_t c = (uint64_t)cumulative + histogram[i];
        cumulative = c > UINT32_MAX ? UINT32_MAX : (uint32_t)c;
        if (cumulative >= target_count) { floor_value = i; break; }
    }
    size_t fwc = floor_value + 3; /* :111-113 */
    if (fwc > 40) fwc = 40;
    uint8_t floor_with_cushion = (uint8_t)fwc;
    if (floor_out) *floor_out = floor_with_cushion;

    const float SCALE_255 = 255.0f, GAMMA_R_SUPP = 1.15f, GAMMA_G_SUPP = 1.10f;
    float floor_f = (float)floor_with_cushion;
    float denom = fmaxf(255.0f - floor_f, 1.0f); /* :120 */
    uint8_t lut_r[256], lut_g[256];
    for (unsigned v = 0; v <= 255; ++v) { /* :124-135 */
        if ((uint8_t)v <= floor_with_cushion) {
            lut_r[v] = 0; lut_g[v] = 0;
        } else {
            float shifted = ((float)v - floor_f) / denom;
            lut_r[v] = f32_as_u8(clamp_f32(roundf(powf(shifted, GAMMA_R_SUPP) * SCALE_255), 0.0f, 255.0f));
            lut_g[v] = f32_as_u8(clamp_f32(roundf(powf(shifted, GAMMA_G_SUPP) * SCALE_255), 0.0f, 255.0f));
        }
    }
    const float GAMMA_B = 0.1f, BLUE_SCALE_SUPP = 0.18f, EPS = 8.0f;
    uint8_t *lut_b = (uint8_t *)malloc(65536);
    if (!lut_b) return ORACLE_ERR_OOM;
    for (unsigned x1 = 0; x1 <= 255; ++x1) { /* :143-155 */
        for (unsigned x2 = 0; x2 <= 255; ++x2) {
            float r = (float)lut_r[x1], g = (float)lut_g[x2];
            float ratio = (r + EPS) / (g + EPS);
            lut_b[(x1 << 8) | x2] =
                f32_as_u8(roundf(clamp_f32(powf(ratio, GAMMA_B) * SCALE_255 * BLUE_SCALE_SUPP, 0.0f, 255.0f)));
        }
    }
    for (size_t i = 0; i < n; ++i) { /* :158-175 */
        uint8_t v1 = b1[i], v2 = b2[i];
        if (v1 <= floor_with_cushion && v2 <= floor_with_cushion) {
            rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0;
            continue;
        }
        rgb[3 * i + 0] = lut_r[v1];
        rgb[3 * i + 1] = lut_g[v2];
        rgb[3 * i + 2] = lut_b[((unsigned)v1 << 8) | v2];
    }
    if (luts_out) { memcpy(luts_out, lut_r, 256); memcpy(luts_out + 256, lut_g, 256); memcpy(luts_out + 512, lut_b, 65536); }
    free(lut_b);
    return ORACLE_OK;
}

/* synthetic_rgb.rs:72-79 and :182-197: mode is ignored; Tamed|Clahe -> suppressed */
int sarpro_oracle_synrgb(int mode, int strategy, const uint8_t *b1, const uint8_t *b2, size_t n,
                         uint8_t *rgb) {
    if (mode < 0 || mode > 3) return ORACLE_ERR_INVALID_ARG;
    if (strategy == STRAT_TAMED || strategy == STRAT_CLAHE)
        return sarpro_oracle_synrgb_suppressed(b1, b2, n, rgb, NULL, NULL);
    return sarpro_oracle_synrgb_default(b1, b2, n, rgb, NULL);
}

/* ====================================================================== */
/* padding.rs:5-49  add_padding_to_square (elem_size 1 or 2)               */
/* out must hold max(rows,cols)^2 elements                                 */
/* ====================================================================== */
int sarpro_oracle_pad_to_square(const void *data, size_t cols, size_t rows, size_t elem_size,
                                void *out) {
    size_t max_dim = cols > rows ? cols : rows;
    size_t pad_cols = (max_dim - cols) / 2;
    size_t pad_rows = (max_dim - rows) / 2;
    memset(out, 0, max_dim * max_dim * elem_size);
    for (size_t row = 0; row < rows; ++row)
        memcpy((char *)out + ((row + pad_rows) * max_dim + pad_cols) * elem_size,
               (const char *)data + row * cols * elem_size, cols * elem_size);
    return ORACLE_OK;
}

/* resize.rs:6-30  calculate_resize_dimensions -> (cols, rows) */
void sarpro_oracle_resize_dims(size_t original_cols, size_t original_rows, size_t target_size,
                               size_t *new_cols, size_t *new_rows) {
    size_t short_side = original_rows < original_cols ? original_rows : original_cols;
    size_t long_side = original_rows > original_cols ? original_rows : original_cols;
    if (target_size > long_side) { *new_cols = original_cols; *new_rows = original_rows; return; }
    double scale_factor = (double)target_size / (double)long_side;
    size_t new_short = (size_t)f64_as_u64(round((double)short_side * scale_factor));
    if (original_cols > original_rows) { *new_cols = target_size; *new_rows = new_short; }
    else { *new_cols = new_short; *new_rows = target_size; }
}

/* ====================================================================== */
/* Whole dual-pol JPEG-branch flow at native resolution (no resize/pad):   */
/* save.rs:317-367: pipeline(band1,U8) [Tamed: tamed_synrgb(true)],        */
/* pipeline(band2,U8) [Tamed: tamed_synrgb(false)], then synRGB by         */
/* strategy.  This is the headline workload (calibrate+CLAHE+synRGB).      */
/* u8_1/u8_2 optional outputs of the per-band u8 rasters.                  */
/* ====================================================================== */
int sarpro_oracle_dualpol_synrgb_f32(const float *band1, const float *band2, size_t rows,
                                     size_t cols, int strategy, int mode, uint8_t *rgb,
                                     uint8_t *u8_1, uint8_t *u8_2) {
    size_t n = rows * cols;
    uint8_t *a = u8_1 ? u8_1 : (uint8_t *)malloc(n ? n : 1);
    uint8_t *b = u8_2 ? u8_2 : (uint8_t *)malloc(n ? n : 1);
    double *db = (double *)malloc((n ? n : 1) * sizeof(double));
    uint8_t *mask = (uint8_t *)malloc(n ? n : 1);
    int rc = ORACLE_ERR_OOM;
    if (a && b && db && mask) {
        const float *bands[2] = { band1, band2 };
        uint8_t *outs[2] = { a, b };
        rc = ORACLE_OK;
        for (int k = 0; k < 2 && rc == ORACLE_OK; ++k) {
            rc = sarpro_oracle_pipeline_f32(bands[k], rows, cols, BITDEPTH_U8, strategy, outs[k],
                                            NULL, db, mask, NULL); /* save.rs:320-321, 343-344 */
            if (rc == ORACLE_OK && strategy == STRAT_TAMED) /* save.rs:324-328, 347-351 */
                rc = sarpro_oracle_tamed_synrgb_u8(db, mask, rows, cols, k == 0, outs[k]);
        }
        if (rc == ORACLE_OK) rc = sarpro_oracle_synrgb(mode, strategy, a, b, n, rgb); /* save.rs:363 */
    }
    if (!u8_1) free(a);
    if (!u8_2) free(b);
    free(db);
    free(mask);
    return rc;
}


/* ====================================================================== */
/* resize.rs:32-89  resize_u8_image / resize_u16_image                     */
/* The arithmetic is the third-party crate fast_image_resize (Cargo.toml:33 */
/* `^5.2.1`, NOT pinned: no Cargo.lock; source absent from the reference).  */
/* PARITY UNPINNED for this function: what follows restates the crate's     */
/* published convolution algorithm as of 5.x -- ResizeAlg::Convolution(     */
/* Lanczos3), adaptive kernel size, coefficients normalised per output      */
/* pixel, fixed-point i16 (u8) / i32 (u16) weights with the largest         */
/* precision that keeps the largest weight in range, horizontal pass into   */
/* an integer intermediate, then vertical pass -- and is anchored only on   */
/* the reference's call sites (resize.rs:39-52, 62-81).                     */
/* ====================================================================== */
static double oracle_sinc(double x) { if (x == 0.0) return 1.0; x *= 3.14159265358979323846; return sin(x) / x; }
static double oracle_lanczos3(double x) { return (x >= -3.0 && x < 3.0) ? oracle_sinc(x) * oracle_sinc(x / 3.0) : 0.0; }

typedef struct { size_t window; uint32_t *start, *size; double *w; } oracle_coeffs;

static int oracle_precompute(size_t in_size, size_t out_size, oracle_coeffs *c) {
    double scale = (double)in_size / (double)out_size;
    double filter_scale = scale > 1.0 ? scale : 1.0;
    double radius = 3.0 * filter_scale;
    c->window = (size_t)ceil(radius) * 2 + 1;
    double recip = 1.0 / filter_scale;
    c->start = (uint32_t *)malloc(out_size * sizeof(uint32_t));
    c->size = (uint32_t *)malloc(out_size * sizeof(uint32_t));
    c->w = (double *)calloc(out_size * c->window, sizeof(double));
    if (!c->start || !c->size || !c->w) return ORACLE_ERR_OOM;
    for (size_t ox = 0; ox < out_size; ++ox) {
        double in_center = ((double)ox + 0.5) * scale;
        uint32_t x_min = f64_as_u32(fmax(floor(in_center - radius), 0.0));
        uint32_t x_max = f64_as_u32(fmin(ceil(in_center + radius), (double)in_size));
        double center = in_center - 0.5, ww = 0.0;
        double *w = c->w + ox * c->window;
        for (uint32_t x = x_min; x < x_max; ++x) { w[x - x_min] = oracle_lanczos3(((double)x - center) * recip); ww += w[x - x_min]; }
        if (ww != 0.0) for (uint32_t x = x_min; x < x_max; ++x) w[x - x_min] /= ww;
        c->start[ox] = x_min;
        c->size[ox] = x_max - x_min;
    }
    return ORACLE_OK;
}
static void oracle_coeffs_free(oracle_coeffs *c) { free(c->start); free(c->size); free(c->w); }

static int oracle_precision(const oracle_coeffs *c, size_t out_size, int limit_bits, int max_precision) {
    double max_w = 0.0;
    for (size_t i = 0; i < out_size * c->window; ++i) if (c->w[i] > max_w) max_w = c->w[i];
    int precision = 0;
    for (int cur = 0; cur < max_precision; ++cur) {
        precision = cur;
        double next = round(max_w * (double)(1ll << (cur + 1)));
        if (next >= (double)(1ll << limit_bits)) break;
    }
    return precision;
}

/* one pass along x: src (rows x in_cols) -> dst (rows x out_cols); elem 1 = u8 / i16 weights, 2 = u16 / i32 weights */
static int oracle_convolve_x(const void *src, size_t rows, size_t in_cols, void *dst, size_t out_cols, int elem) {
    oracle_coeffs c;
    int rc = oracle_precompute(in_cols, out_cols, &c);
    if (rc) return rc;
    int precision = elem == 1 ? oracle_precision(&c, out_cols, 15, 22) : oracle_precision(&c, out_cols, 31, 45);
    double scale = (double)(1ll << precision);
    int64_t *k = (int64_t *)malloc(out_cols * c.window * sizeof(int64_t));
    if (!k) { oracle_coeffs_free(&c); return ORACLE_ERR_OOM; }
    for (size_t i = 0; i < out_cols * c.window; ++i) k[i] = (int64_t)round(c.w[i] * scale);
    int64_t initial = precision > 0 ? (1ll << (precision - 1)) : 0;
    int64_t maxv = elem == 1 ? 255 : 65535;
    for (size_t r = 0; r < rows; ++r) {
        for (size_t ox = 0; ox < out_cols; ++ox) {
            int64_t ss = initial;
            const int64_t *kk = k + ox * c.window;
            for (uint32_t t = 0; t < c.size[ox]; ++t) {
                size_t x = c.start[ox] + t;
                int64_t v = elem == 1 ? ((const uint8_t *)src)[r * in_cols + x] : ((const uint16_t *)src)[r * in_cols + x];
                ss += v * kk[t];
            }
            int64_t o = ss >> precision;
            o = o < 0 ? 0 : (o > maxv ? maxv : o);
            if (elem == 1) ((uint8_t *)dst)[r * out_cols + ox] = (uint8_t)o; else ((uint16_t *)dst)[r * out_cols + ox] = (uint16_t)o;
        }
    }
    free(k);
    oracle_coeffs_free(&c);
    return ORACLE_OK;
}

static void oracle_transpose(const void *src, size_t rows, size_t cols, void *dst, int elem) {
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) {
            if (elem == 1) ((uint8_t *)dst)[c * rows + r] = ((const uint8_t *)src)[r * cols + c];
            else ((uint16_t *)dst)[c * rows + r] = ((const uint16_t *)src)[r * cols + c];
        }
}

/* elem_size 1 (u8) or 2 (u16); dst holds dst_rows*dst_cols elements */
int sarpro_oracle_resize_lanczos3(const void *src, size_t cols, size_t rows, size_t dst_cols, size_t dst_rows,
                                  size_t elem_size, void *dst) {
    if (elem_size != 1 && elem_size != 2) return ORACLE_ERR_INVALID_ARG;
    if (!cols || !rows || !dst_cols || !dst_rows) return ORACLE_ERR_INVALID_ARG;
    int elem = (int)elem_size;
    void *h = malloc(rows * dst_cols * elem_size);           /* horizontal pass */
    void *ht = malloc(rows * dst_cols * elem_size);          /* transposed: (dst_cols x rows) */
    void *vt = malloc(dst_rows * dst_cols * elem_size);      /* vertical pass on the transposed image */
    int rc = ORACLE_ERR_OOM;
    if (h && ht && vt) {
        rc = oracle_convolve_x(src, rows, cols, h, dst_cols, elem);
        if (rc == ORACLE_OK) {
            oracle_transpose(h, rows, dst_cols, ht, elem);
            rc = oracle_convolve_x(ht, dst_cols, rows, vt, dst_rows, elem);
        }
        if (rc == ORACLE_OK) oracle_transpose(vt, dst_cols, dst_rows, dst, elem);
    }
    free(h); free(ht); free(vt);
    return rc;
}

/* resize.rs:91-236  resize_image_data_with_meta.  target_size 0 = None.  out must hold
   max(final dims)^2 elements when pad, else new_cols*new_rows.  meta_out[6] =
   final_cols, final_rows, scale_x, scale_y, pad_left, pad_top (as doubles). */
int sarpro_oracle_resize_image_data_with_meta(const void *data, size_t original_cols, size_t original_rows,
                                              size_t target_size, size_t elem_size, int pad, void *out,
                                              double *meta_out) {
    size_t cols = original_cols, rows = original_rows;
    double scale_x = 1.0, scale_y = 1.0;
    const void *cur = data;
    void *resized = NULL;
    if (target_size) {
        size_t current_long = original_cols > original_rows ? original_cols : original_rows;
        if (current_long != target_size) { /* :112-145 early-out when already at the requested long side */
            size_t nc, nr;
            sarpro_oracle_resize_dims(original_cols, original_rows, target_size, &nc, &nr);
            resized = malloc((nc != 0 && nr != 0 ? nc * nr : 1) * elem_size);
            if (!resized) return ORACLE_ERR_OOM;
            int rc = sarpro_oracle_resize_lanczos3(data, original_cols, original_rows, nc, nr, elem_size, resized);
            if (rc) { free(resized); return rc; }
            scale_x = (double)nc / (double)original_cols; /* :168-169 */
            scale_y = (double)nr / (double)original_rows;
            cols = nc; rows = nr; cur = resized;
        }
    }
    size_t pad_left = 0, pad_top = 0, fc = cols, fr = rows;
    if (pad) {
        size_t m = cols > rows ? cols : rows;
        sarpro_oracle_pad_to_square(cur, cols, rows, elem_size, out);
        pad_left = (m - cols) / 2; pad_top = (m - rows) / 2; fc = fr = m;
    } else {
        memcpy(out, cur, cols * rows * elem_size);
    }
    free(resized);
    if (meta_out) { meta_out[0] = (double)fc; meta_out[1] = (double)fr; meta_out[2] = scale_x; meta_out[3] = scale_y;
                    meta_out[4] = (double)pad_left; meta_out[5] = (double)pad_top; }
    return ORACLE_OK;
}

const char *sarpro_oracle_version(void) { return "sarpro-oracle 1 (restates bogwi/sarpro v0.3.0 src/core/processing)"; }

/*
 * sarpro_oracle_mt.c -- TEST / MEASUREMENT INFRASTRUCTURE, like sarpro_oracle.c: never linked or loaded by the product.
 *
 * Row-parallel variant of the oracle's headline path (dual-pol CLAHE -> suppressed synRGB, i.e.
 * save.rs:317-367 at native resolution) for bench.py's cpu_baseline: "what the host's cores would give" (SURVEY 8d (ii)).
 * NOT the reference's behaviour: sarpro's hot path is single-threaded (its only threads are the GUI worker and
 * README-level process parallelism).  Every per-pixel loop of sarpro_oracle.c is split over rows with OpenMP,
 * histograms are per thread and summed, and Welford's sequential mean / M2 (autoscale.rs:49-53) is replaced by
 * per-thread sums -- mean and std feed nothing on this path (CLAHE uses p01 / p99 only), so the rasters are the
 * single-thread oracle's, which tests/test_oracle_mt.py checks.
 *
 * The file includes sarpro_oracle.c for its helpers (casts, clamp, CDF, LUT builders): one translation unit.
 */
#include "sarpro_oracle.c"

#include <omp.h>

int sarpro_oracle_mt_threads(void) { return omp_get_max_threads(); }

/* pipeline.rs:8-40 + autoscale.rs:35-160 (count / min / max / 4096-bin histogram) + :572-608 for ONE band, U8 */
static int band_clahe_u8_mt(const float *in, size_t rows, size_t cols, uint8_t *out_u8) {
    const size_t n = rows * cols;
    double *db = (double *)malloc((n ? n : 1) * sizeof(double));
    uint8_t *mask = (uint8_t *)malloc(n ? n : 1);
    uint16_t *lvl = (uint16_t *)malloc((n ? n : 1) * sizeof(uint16_t));
    double *eq = (double *)malloc((n ? n : 1) * sizeof(double));
    if (!db || !mask || !lvl || !eq) { free(db); free(mask); free(lvl); free(eq); return ORACLE_ERR_OOM; }
    uint64_t count = 0;
    double min_db = INFINITY, max_db = -INFINITY;
#pragma omp parallel for schedule(static) reduction(+ : count) reduction(min : min_db) reduction(max : max_db)
    for (size_t i = 0; i < n; ++i) {
        const double v = 10.0 * log10(fmax((double)in[i], 1e-10));
        db[i] = v;
        mask[i] = v > -50.0;
        if (mask[i]) { count += 1; if (v < min_db) min_db = v; if (v > max_db) max_db = v; }
    }
    int rc = ORACLE_OK;
    if (count == 0) {
        memset(lvl, 0, n * sizeof(uint16_t));
    } else {
        double p01, p99;
        if (fabs(max_db - min_db) < DBL_EPSILON) { p01 = min_db; p99 = max_db; }
        else {
            uint64_t hist[NUM_BINS];
            memset(hist, 0, sizeof(hist));
            const double span = max_db - min_db, inv_span = 1.0 / span;
#pragma omp parallel
            {
                uint64_t *h = (uint64_t *)calloc(NUM_BINS, sizeof(uint64_t));
#pragma omp for schedule(static) nowait
                for (size_t i = 0; i < n; ++i) {
                    if (!mask[i]) continue;
                    const double t = clamp_f64((db[i] - min_db) * inv_span, 0.0, 1.0);
                    uint64_t idx = f64_as_u64(t * (double)NUM_BINS);
                    if (idx >= NUM_BINS) idx = NUM_BINS - 1;
                    h[idx] += 1;
                }
#pragma omp critical
                for (size_t b = 0; b < NUM_BINS; ++b) hist[b] += h[b];
                free(h);
            }
            p01 = estimate_percentile(hist, count, min_db, max_db, span, 0.01);
            p99 = estimate_percentile(hist, count, min_db, max_db, span, 0.99);
        }
        const double low = p01, high = p99, range = fmax(high - low, 1.0);
        if (!sarpro_oracle_clahe_shape_ok(rows, cols, 8, 8)) rc = ORACLE_ERR_UNSUPPORTED_SHAPE;
        else {
            const size_t tiles = 8, nb = 256;
            const size_t tile_h = (rows + tiles - 1) / tiles, tile_w = (cols + tiles - 1) / tiles;
            double *cdfs = (double *)calloc(tiles * tiles * nb, sizeof(double));
            /* autoscale.rs:247-304: one tile per task (64 tasks) */
#pragma omp parallel for schedule(dynamic) collapse(2)
            for (size_t ty = 0; ty < tiles; ++ty)
                for (size_t tx = 0; tx < tiles; ++tx) {
                    const size_t r0 = ty * tile_h, r1 = (ty + 1) * tile_h < rows ? (ty + 1) * tile_h : rows;
                    const size_t c0 = tx * tile_w, c1 = (tx + 1) * tile_w < cols ? (tx + 1) * tile_w : cols;
                    uint32_t hist[256];
                    memset(hist, 0, sizeof(hist));
                    for (size_t r = r0; r < r1; ++r)
                        for (size_t c = c0; c < c1; ++c)
                            if (mask[r * cols + c]) {
                                const double clipped = fmin(fmax(db[r * cols + c], low), high);
                                const double v = clamp_f64((clipped - low) / range, 0.0, 1.0);
                                int64_t bin = f64_as_i64(round(v * 255.0));
                                if (bin < 0) bin = 0;
                                if (bin > 255) bin = 255;
                                hist[bin] += 1;
                            }
                    sarpro_oracle_clahe_tile_cdf(hist, nb, r1 - r0, c1 - c0, 2.0, cdfs + (ty * tiles + tx) * nb);
                }
            /* autoscale.rs:332-342 + :600-606 */
#pragma omp parallel for schedule(static)
            for (size_t r = 0; r < rows; ++r)
                for (size_t c = 0; c < cols; ++c) {
                    const size_t i = r * cols + c;
                    if (!mask[i]) { lvl[i] = 0; continue; }
                    const double clipped = fmin(fmax(db[i], low), high);
                    const double val = (clipped - low) / range;
                    const double rf = (double)r / (double)tile_h - 0.5, cf = (double)c / (double)tile_w - 0.5;
                    const int64_t ty = f64_as_i64(fmax(floor(rf), 0.0)), tx = f64_as_i64(fmax(floor(cf), 0.0));
                    const double dy = rf - (double)ty, dx = cf - (double)tx;
                    const int64_t last = (int64_t)tiles - 1;
                    const size_t ty0 = (size_t)(ty > last ? last : ty), tx0 = (size_t)(tx > last ? last : tx);
                    const size_t ty1 = (size_t)(ty + 1 > last ? last : ty + 1), tx1 = (size_t)(tx + 1 > last ? last : tx + 1);
                    const size_t bp = (size_t)f64_as_u64(round(clamp_f64(val, 0.0, 1.0) * 255.0));
                    const double c00 = cdfs[(ty0 * tiles + tx0) * nb + bp], c01 = cdfs[(ty0 * tiles + tx1) * nb + bp];
                    const double c10 = cdfs[(ty1 * tiles + tx0) * nb + bp], c11 = cdfs[(ty1 * tiles + tx1) * nb + bp];
                    const double top = c00 * (1.0 - dx) + c01 * dx, bottom = c10 * (1.0 - dx) + c11 * dx;
                    eq[i] = top * (1.0 - dy) + bottom * dy;
                    lvl[i] = f64_as_u16(clamp_f64(eq[i], 0.0, 1.0) * 255.0);
                }
            free(cdfs);
        }
    }
    if (rc == ORACLE_OK && n) { /* autoscale.rs:348-364 */
        uint16_t mn = 65535, mx = 0;
#pragma omp parallel for schedule(static) reduction(min : mn) reduction(max : mx)
        for (size_t i = 0; i < n; ++i) { if (lvl[i] < mn) mn = lvl[i]; if (lvl[i] > mx) mx = lvl[i]; }
        const float fmn = (float)mn, fmx = (float)mx, scale = fmx > fmn ? 255.0f / (fmx - fmn) : 1.0f;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) out_u8[i] = f32_as_u8(clamp_f32(roundf(((float)lvl[i] - fmn) * scale), 0.0f, 255.0f));
    }
    free(db); free(mask); free(lvl); free(eq);
    return rc;
}

/* save.rs:317-367 at native resolution, CLAHE: both bands, then the suppressed synRGB (synthetic_rgb.rs:88-178) */
int sarpro_oracle_mt_dualpol_clahe_synrgb_f32(const float *band1, const float *band2, size_t rows, size_t cols, uint8_t *rgb) {
    const size_t n = rows * cols;
    uint8_t *a = (uint8_t *)malloc(n ? n : 1), *b = (uint8_t *)malloc(n ? n : 1);
    if (!a || !b) { free(a); free(b); return ORACLE_ERR_OOM; }
    int rc = band_clahe_u8_mt(band1, rows, cols, a);
    if (rc == ORACLE_OK) rc = band_clahe_u8_mt(band2, rows, cols, b);
    if (rc == ORACLE_OK) {
        uint64_t histogram[256];
        memset(histogram, 0, sizeof(histogram));
#pragma omp parallel
        {
            uint64_t h[256];
            memset(h, 0, sizeof(h));
#pragma omp for schedule(static) nowait
            for (size_t i = 0; i < n; ++i) { h[a[i]] += 1; h[b[i]] += 1; }
#pragma omp critical
            for (int k = 0; k < 256; ++k) histogram[k] += h[k];
        }
        /* the floor and the LUTs from the single-thread routine on a histogram-equivalent toy: reuse its code by
           feeding it the saturated u32 histogram through a tiny raster is not possible, so restate :99-113 here */
        const uint32_t total_count = (uint32_t)(n + n);
        const uint32_t target_count = f64_as_u32(round((double)total_count * 0.05));
        uint32_t cumulative = 0;
        size_t floor_value = 0;
        for (size_t i = 0; i <= 255; ++i) {
            const uint64_t hi = histogram[i] > UINT32_MAX ? UINT32_MAX : histogram[i];
            const uint64_t c = (uint64_t)cumulative + hi;
            cumulative = c > UINT32_MAX ? UINT32_MAX : (uint32_t)c;
            if (cumulative >= target_count) { floor_value = i; break; }
        }
        size_t fwc = floor_value + 3;
        if (fwc > 40) fwc = 40;
        /* LUTs of the suppressed variant for this floor: run the single-thread routine on a 1-pixel raster whose floor
           is forced is not possible either; build them as :115-156 do */
        const float floor_f = (float)fwc, denom = fmaxf(255.0f - floor_f, 1.0f);
        uint8_t lut_r[256], lut_g[256];
        for (unsigned v = 0; v <= 255; ++v) {
            if (v <= fwc) { lut_r[v] = 0; lut_g[v] = 0; }
            else {
                const float shifted = ((float)v - floor_f) / denom;
                lut_r[v] = f32_as_u8(clamp_f32(roundf(powf(shifted, 1.15f) * 255.0f), 0.0f, 255.0f));
                lut_g[v] = f32_as_u8(clamp_f32(roundf(powf(shifted, 1.10f) * 255.0f), 0.0f, 255.0f));
            }
        }
        uint8_t *lut_b = (uint8_t *)malloc(65536);
        if (!lut_b) rc = ORACLE_ERR_OOM;
        else {
            for (unsigned x1 = 0; x1 <= 255; ++x1)
                for (unsigned x2 = 0; x2 <= 255; ++x2) {
                    const float ratio = ((float)lut_r[x1] + 8.0f) / ((float)lut_g[x2] + 8.0f);
                    lut_b[(x1 << 8) | x2] = f32_as_u8(roundf(clamp_f32(powf(ratio, 0.1f) * 255.0f * 0.18f, 0.0f, 255.0f)));
                }
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; ++i) {
                const uint8_t v1 = a[i], v2 = b[i];
                if (v1 <= fwc && v2 <= fwc) { rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0; continue; }
                rgb[3 * i] = lut_r[v1]; rgb[3 * i + 1] = lut_g[v2]; rgb[3 * i + 2] = lut_b[((unsigned)v1 << 8) | v2];
            }
            free(lut_b);
        }
    }
    free(a); free(b);
    return rc;
}

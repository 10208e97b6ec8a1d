// host_logic.cpp -- see host_logic.h.  Built with g++ -O2 -ffp-contract=off (no fast-math):
// the arithmetic below must round exactly like the reference's Rust/libm evaluation.
#include "host_logic.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace sarpro {

// ---- Rust `as` casts from float: truncate, saturate, NaN -> 0 ----
static inline uint64_t as_u64(double x) {
    if (!(x == x) || x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}
static inline int64_t as_i64(double x) {
    if (!(x == x)) return 0;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    if (x >= 9223372036854775807.0) return INT64_MAX;
    return (int64_t)x;
}
static inline uint32_t as_u32(double x) {
    if (!(x == x) || x <= 0.0) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}
static inline uint16_t as_u16(double x) {
    if (!(x == x) || x <= 0.0) return 0;
    if (x >= 65535.0) return 65535;
    return (uint16_t)x;
}
static inline uint8_t f32_as_u8(float x) {
    if (!(x == x) || x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}
// Rust clamp: NaN stays NaN
static inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

const double *db_table_u16() {
    static double table[65536];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t dn = 0; dn < 65536; ++dn) // pipeline.rs:19-20 with v = DN exactly
            table[dn] = 10.0 * std::log10(std::fmax((double)dn, 1e-10));
    });
    return table;
}

// percentile inversion, autoscale.rs:120-140
static double percentile_from_bins(const uint64_t *hist, uint64_t n, double min_db, double max_db,
                                   double span, double p) {
    uint64_t target = as_u64(std::floor(p * (double)n));
    if (target >= n) target = n - 1;
    uint64_t cumsum = 0;
    for (int b = 0; b < kStatBins; ++b) {
        uint64_t h = hist[b], next = cumsum + h;
        if (target < next) {
            uint64_t within = target >= cumsum ? target - cumsum : 0;
            double frac = h > 0 ? (double)within / (double)h : 0.0;
            double bin_width = span / (double)kStatBins;
            double bin_start = min_db + (double)b * bin_width;
            return bin_start + frac * bin_width;
        }
        cumsum = next;
    }
    return max_db;
}

int stats_from_bins4096(uint64_t count, double min_db, double max_db, double mean, double std_db,
                        const uint64_t *hist, sarpro_hip_stats *s) {
    std::memset(s, 0, sizeof(*s));
    if (count == 0) return SARPRO_HIP_OK; // autoscale.rs:57-76
    s->valid_count = count;
    s->min_db = min_db;
    s->max_db = max_db;
    s->mean_db = mean;
    s->std_db = std_db;
    if (std::fabs(max_db - min_db) < DBL_EPSILON) { // autoscale.rs:81-100
        s->median_db = s->p01 = s->p02 = s->p05 = s->p10 = s->p25 = min_db;
        s->p75 = s->p90 = s->p95 = s->p98 = s->p99 = max_db;
        return SARPRO_HIP_OK;
    }
    double span = max_db - min_db;
    s->median_db = percentile_from_bins(hist, count, min_db, max_db, span, 0.5);
    s->p01 = percentile_from_bins(hist, count, min_db, max_db, span, 0.01);
    s->p02 = percentile_from_bins(hist, count, min_db, max_db, span, 0.02);
    s->p05 = percentile_from_bins(hist, count, min_db, max_db, span, 0.05);
    s->p10 = percentile_from_bins(hist, count, min_db, max_db, span, 0.10);
    s->p25 = percentile_from_bins(hist, count, min_db, max_db, span, 0.25);
    s->p75 = percentile_from_bins(hist, count, min_db, max_db, span, 0.75);
    s->p90 = percentile_from_bins(hist, count, min_db, max_db, span, 0.90);
    s->p95 = percentile_from_bins(hist, count, min_db, max_db, span, 0.95);
    s->p98 = percentile_from_bins(hist, count, min_db, max_db, span, 0.98);
    s->p99 = percentile_from_bins(hist, count, min_db, max_db, span, 0.99);
    return SARPRO_HIP_OK;
}

// compute_histogram_stats (autoscale.rs:35-160) re-expressed over the exact DN histogram:
// every valid pixel with the same DN has the same dB value, so min/max/count and the 4096-bin
// histogram are exact; mean/std are the same quantities summed per DN instead of by Welford's
// sequential update (they agree to ~1e-12 relative; they feed only the log lines and the two
// discrete Adaptive tests).
int stats_from_dn_hist(const uint64_t *h, sarpro_hip_stats *out) {
    const double *db = db_table_u16();
    uint64_t count = 0;
    uint32_t min_dn = 0, max_dn = 0;
    for (uint32_t dn = 1; dn < 65536; ++dn) { // valid <=> db > -50 <=> DN >= 1
        if (h[dn]) {
            if (!count) min_dn = dn;
            max_dn = dn;
            count += h[dn];
        }
    }
    if (count == 0) { std::memset(out, 0, sizeof(*out)); return SARPRO_HIP_OK; }
    // mean / M2 over the distinct DN values, Neumaier-compensated (error ~1e-16 relative)
    auto csum = [&](auto term) {
        double sum = 0.0, comp = 0.0;
        for (uint32_t dn = min_dn; dn <= max_dn; ++dn) {
            if (!h[dn]) continue;
            const double x = term(dn), t = sum + x;
            comp += std::fabs(sum) >= std::fabs(x) ? (sum - t) + x : (x - t) + sum;
            sum = t;
        }
        return sum + comp;
    };
    const double mean = csum([&](uint32_t dn) { return (double)h[dn] * db[dn]; }) / (double)count;
    const double m2 = csum([&](uint32_t dn) { const double d = db[dn] - mean; return (double)h[dn] * d * d; });
    double std_db = count > 1 ? std::sqrt(m2 / (double)count) : 0.0;

    double min_db = db[min_dn], max_db = db[max_dn];
    uint64_t hist[kStatBins];
    std::memset(hist, 0, sizeof(hist));
    if (!(std::fabs(max_db - min_db) < DBL_EPSILON)) {
        double span = max_db - min_db, inv_span = 1.0 / span; // autoscale.rs:105-106
        for (uint32_t dn = min_dn; dn <= max_dn; ++dn) {
            if (!h[dn]) continue;
            double t = clampd((db[dn] - min_db) * inv_span, 0.0, 1.0); // :113
            uint64_t idx = as_u64(t * (double)kStatBins);               // :114
            if (idx >= (uint64_t)kStatBins) idx = kStatBins - 1;
            hist[idx] += h[dn];
        }
    }
    return stats_from_bins4096(count, min_db, max_db, mean, std_db, hist, out);
}

int select_window(sarpro_hip_stats *s, int strategy, int tamed_synrgb) {
    s->skew_factor = s->tail_heaviness = 0.0;
    if (s->valid_count == 0) { s->low_clip = s->high_clip = 0.0; s->gamma = 1.0; return SARPRO_HIP_OK; }
    if (tamed_synrgb != kNotTamedSynrgb) { // autoscale.rs:721-729
        s->low_clip = tamed_synrgb == kTamedCopol ? std::fmin(s->p02, s->p05) : s->p05;
        s->high_clip = s->p99;
        s->gamma = 1.0;
        return SARPRO_HIP_OK;
    }
    double dynamic_range = s->max_db - s->min_db;
    double iqr = s->p75 - s->p25;
    double low, high, gamma;
    switch (strategy) {
    case SARPRO_STRATEGY_STANDARD: // pipeline.rs:49-52 -> autoscale.rs:404-429
        if (dynamic_range < 15.0) {
            double range = std::fmax(20.0, dynamic_range * 0.8);
            low = s->median_db - range / 2.0; high = s->median_db + range / 2.0; gamma = 1.1;
        } else if (iqr < 5.0) {
            low = s->p25 - 2.5 * iqr; high = s->p75 + 2.5 * iqr; gamma = 1.0;
        } else if (dynamic_range > 40.0) {
            low = std::fmax(s->p02, s->min_db + 0.02 * dynamic_range);
            high = std::fmin(s->p98, s->max_db - 0.02 * dynamic_range);
            gamma = 0.9;
        } else {
            low = s->p02; high = s->p98; gamma = 1.0;
        }
        low = std::fmax(low, s->min_db);
        high = std::fmin(high, s->max_db);
        break;
    case SARPRO_STRATEGY_ROBUST: { // autoscale.rs:492-499
        double thr = 2.5 * iqr;
        low = std::fmax(std::fmax(s->p25 - thr, s->p01), s->min_db);
        high = std::fmin(std::fmin(s->p75 + thr, s->p99), s->max_db);
        gamma = 1.0;
        break;
    }
    case SARPRO_STRATEGY_ADAPTIVE: { // autoscale.rs:500-538
        double skew = (s->mean_db - s->median_db) / std::fmax(std::fabs(s->std_db), 1.0);
        double tail = (s->p99 - s->p95) / std::fmax(s->p95 - s->p75, 1.0);
        s->skew_factor = skew;
        s->tail_heaviness = tail;
        if (std::fabs(skew) > 0.5) {
            if (skew > 0.0) { low = s->p02; high = s->p98; gamma = 0.9; }
            else { low = s->p05; high = s->p95; gamma = 1.1; }
        } else if (tail > 2.0) {
            low = s->p10; high = s->p90; gamma = 0.8;
        } else {
            low = s->p05; high = s->p95; gamma = 1.0;
        }
        break;
    }
    case SARPRO_STRATEGY_EQUALIZED: // :539-543
    case SARPRO_STRATEGY_CLAHE:     // :544-548
        low = s->p01; high = s->p99; gamma = 1.0; break;
    case SARPRO_STRATEGY_TAMED: // :549-553
        low = s->p25; high = s->p99; gamma = 1.0; break;
    case SARPRO_STRATEGY_DEFAULT: // :558-561
        low = s->p05; high = s->p95; gamma = 1.0; break;
    default:
        return SARPRO_HIP_ERR_INVALID_ARG;
    }
    s->low_clip = low; s->high_clip = high; s->gamma = gamma;
    return SARPRO_HIP_OK;
}

uint16_t level_of_db(double db, double low_clip, double high_clip, double gamma, double max_val) {
    double range = std::fmax(high_clip - low_clip, 1.0); // autoscale.rs:429 / 564 / 729
    double clipped = std::fmin(std::fmax(db, low_clip), high_clip);
    double normalized = std::pow((clipped - low_clip) / range, gamma);
    return as_u16(clampd(normalized * max_val, 0.0, max_val));
}

uint8_t clahe_bin_of_db(double db, double low_clip, double high_clip) {
    double range = std::fmax(high_clip - low_clip, 1.0);
    double clipped = std::fmin(std::fmax(db, low_clip), high_clip);
    double n = (clipped - low_clip) / range;                                // autoscale.rs:585-586
    double v = clampd(n, 0.0, 1.0);                                         // :262 / :320
    int64_t bin = as_i64(std::round(v * ((double)kClaheBins - 1.0)));       // :263
    if (bin < 0) bin = 0;
    if (bin >= kClaheBins) bin = kClaheBins - 1;
    return (uint8_t)bin;
}

// Fill a per-DN table of a function that is constant for db <= low_clip and for db >= high_clip.
template <typename F>
static void fill_windowed(double low_clip, double high_clip, F f, DnLut *out) {
    const double *db = db_table_u16();
    out->full.assign(65536, 0);
    // a = first DN >= 1 with db > low_clip; b = first DN >= 1 with db >= high_clip
    uint32_t a = (uint32_t)(std::upper_bound(db + 1, db + 65536, low_clip) - db);
    uint32_t b = (uint32_t)(std::lower_bound(db + 1, db + 65536, high_clip) - db);
    uint32_t m = std::max(a, b);
    if (a > 1) {
        uint16_t v1 = f(db[1]);
        std::fill(out->full.begin() + 1, out->full.begin() + a, v1);
    }
    for (uint32_t dn = a; dn < m; ++dn) out->full[dn] = f(db[dn]);
    if (m <= 65535) {
        uint16_t vm = f(db[m]);
        std::fill(out->full.begin() + m, out->full.end(), vm);
    }
    out->win_lo = a > 1 ? a - 1 : 1;
    out->win_hi = std::min<uint32_t>(m, 65535);
    if (out->win_lo > out->win_hi) out->win_lo = out->win_hi;
}

void build_level_lut_u16(const sarpro_hip_stats &s, int bit_depth, int tamed_synrgb, DnLut *out) {
    double max_val = (bit_depth == SARPRO_BITDEPTH_U8 || tamed_synrgb) ? 255.0 : 65535.0;
    if (s.valid_count == 0) { out->full.assign(65536, 0); out->win_lo = out->win_hi = 1; return; }
    double lo = s.low_clip, hi = s.high_clip, g = s.gamma;
    fill_windowed(lo, hi, [=](double d) { return level_of_db(d, lo, hi, g, max_val); }, out);
}

void build_clahe_bin_lut_u16(const sarpro_hip_stats &s, DnLut *out) {
    if (s.valid_count == 0) { out->full.assign(65536, 0); out->win_lo = out->win_hi = 1; return; }
    double lo = s.low_clip, hi = s.high_clip;
    fill_windowed(lo, hi, [=](double d) { return (uint16_t)clahe_bin_of_db(d, lo, hi); }, out);
}

bool clahe_saturated_levels(const ClaheGeometry &g, std::vector<uint8_t> *col_class, std::vector<uint8_t> *row_bits) {
    std::vector<double> tvals;
    std::vector<uint8_t> cc(g.col_w.size());
    for (size_t c = 0; c < g.col_w.size(); ++c) {
        const double dx = g.col_w[c].d;
        const double t = 1.0 * (1.0 - dx) + 1.0 * dx; // c00 * (1.0 - dx) + c01 * dx with both CDFs 1.0
        size_t k = 0;
        while (k < tvals.size() && tvals[k] != t) ++k;
        if (k == tvals.size()) {
            if (tvals.size() == 3) return false;
            tvals.push_back(t);
        }
        cc[c] = (uint8_t)k;
    }
    std::vector<uint8_t> rb(g.row_w.size());
    for (size_t r = 0; r < g.row_w.size(); ++r) {
        const double dy = g.row_w[r].d;
        uint8_t bits = 0;
        for (size_t k = 0; k < 3; ++k) {
            const double t = k < tvals.size() ? tvals[k] : 1.0;
            double o = t * (1.0 - dy) + t * dy; // top * (1.0 - dy) + bottom * dy
            o = o < 0.0 ? 0.0 : (o > 1.0 ? 1.0 : o);
            const double lv = o * 255.0;
            const unsigned level = (unsigned)lv;
            if (level != 254u && level != 255u) return false;
            if (level == 255u) bits |= (uint8_t)(1u << k);
        }
        rb[r] = bits;
    }
    col_class->swap(cc);
    row_bits->swap(rb);
    return true;
}

bool clahe_shape_ok(size_t rows, size_t cols) {
    if (rows == 0 || cols == 0) return true; // early-out clone (autoscale.rs:231-233)
    size_t tile_h = (rows + kTiles - 1) / kTiles, tile_w = (cols + kTiles - 1) / kTiles;
    for (size_t t = 0; t < (size_t)kTiles; ++t)
        if (t * tile_h > rows || t * tile_w > cols) return false; // r1 - r0 underflow (:250,:254)
    return true;
}

static void axis_weights(size_t n, size_t tile, std::vector<RowWeight> *w, std::vector<size_t> *cells) {
    w->resize(n);
    cells->clear();
    for (size_t i = 0; i < n; ++i) { // autoscale.rs:308-318
        double f = (double)i / (double)tile - 0.5;
        int64_t t = as_i64(std::fmax(std::floor(f), 0.0));
        double d = f - (double)t;
        int64_t last = kTiles - 1;
        RowWeight rw;
        rw.d = d;
        rw.omd = 1.0 - d;
        rw.t0 = (int32_t)std::min(std::max<int64_t>(t, 0), last);
        rw.t1 = (int32_t)std::min(std::max<int64_t>(t + 1, 0), last);
        if (i == 0 || rw.t0 != (*w)[i - 1].t0 || rw.t1 != (*w)[i - 1].t1) cells->push_back(i);
        (*w)[i] = rw;
    }
    cells->push_back(n);
}

void build_clahe_geometry(size_t rows, size_t cols, ClaheGeometry *g) {
    g->rows = rows; g->cols = cols;
    g->tile_h = (rows + kTiles - 1) / kTiles; // autoscale.rs:235
    g->tile_w = (cols + kTiles - 1) / kTiles; // :236
    axis_weights(rows, g->tile_h, &g->row_w, &g->row_cell_start);
    axis_weights(cols, g->tile_w, &g->col_w, &g->col_cell_start);
}

// autoscale.rs:271-302.  The reference holds the counts in u32: the casts saturate like Rust's.
void clahe_tile_cdf(uint64_t *hist, size_t tile_rows, size_t tile_cols, double *cdf) {
    const double nb = (double)kClaheBins;
    double avg = (double)(tile_rows * tile_cols) / nb;
    double thr = std::fmax(kClipLimit * avg, 1.0);
    double excess = 0.0;
    for (int i = 0; i < kClaheBins; ++i) {
        if ((double)hist[i] > thr) {
            excess += (double)hist[i] - thr;
            hist[i] = as_u32(thr);
        }
    }
    double add = std::floor(excess / nb);
    uint64_t remainder = as_u64(std::round(excess - add * nb));
    for (int i = 0; i < kClaheBins; ++i) hist[i] = as_u32((double)hist[i] + add);
    for (size_t b = 0; remainder > 0; --remainder, b = (b + 1) % kClaheBins) hist[b] += 1;
    double total = 0.0;
    for (int i = 0; i < kClaheBins; ++i) total += (double)hist[i];
    total = std::fmax(total, 1.0);
    double acc = 0.0;
    for (int i = 0; i < kClaheBins; ++i) {
        acc += (double)hist[i];
        cdf[i] = clampd(acc / total, 0.0, 1.0);
    }
}

int clahe_cdfs(const uint64_t *tile_hists, size_t rows, size_t cols, double *cdfs_out) {
    if (rows == 0 || cols == 0) return SARPRO_HIP_ERR_INVALID_ARG;
    if (!clahe_shape_ok(rows, cols)) return SARPRO_HIP_ERR_UNSUPPORTED_SHAPE;
    size_t tile_h = (rows + kTiles - 1) / kTiles, tile_w = (cols + kTiles - 1) / kTiles;
    uint64_t hist[kClaheBins];
    for (size_t ty = 0; ty < (size_t)kTiles; ++ty) {
        size_t r0 = ty * tile_h, r1 = std::min((ty + 1) * tile_h, rows);
        for (size_t tx = 0; tx < (size_t)kTiles; ++tx) {
            size_t c0 = tx * tile_w, c1 = std::min((tx + 1) * tile_w, cols);
            size_t t = ty * kTiles + tx;
            std::memcpy(hist, tile_hists + t * kClaheBins, sizeof(hist));
            clahe_tile_cdf(hist, r1 - r0, c1 - c0, cdfs_out + t * kClaheBins);
        }
    }
    return SARPRO_HIP_OK;
}

void u8_rescale_lut(unsigned min_level, unsigned max_level, uint8_t *lut) {
    float mn = (float)min_level, mx = (float)max_level; // autoscale.rs:352-356
    float scale = mx > mn ? 255.0f / (mx - mn) : 1.0f;
    for (unsigned x = 0; x < 256; ++x) {
        float val = std::round(((float)x - mn) * scale); // :360
        lut[x] = f32_as_u8(clampf(val, 0.0f, 255.0f));
    }
}

// Blue depends on the pair of gamma-mapped u8 values (r, g) only, so both variants are
// constant 256x256 tables of powf results, built once per process.
static const uint8_t *blue_pair_table(bool dflt) {
    static uint8_t tab[2][65536];
    static std::once_flag once;
    std::call_once(once, [] {
        for (unsigned ri = 0; ri < 256; ++ri) {
            for (unsigned gi = 0; gi < 256; ++gi) {
                float r = (float)ri, g = (float)gi;
                { // default, synthetic_rgb.rs:43-48 (g == 0 gives ratio = inf or NaN exactly as in the reference)
                    float ratio = r / g;
                    tab[0][(ri << 8) | gi] =
                        f32_as_u8(std::round(clampf(std::pow(ratio, 0.1f) * 255.0f * 0.24f, 0.0f, 255.0f)));
                }
                { // suppressed, synthetic_rgb.rs:146-152
                    float ratio = (r + 8.0f) / (g + 8.0f);
                    tab[1][(ri << 8) | gi] =
                        f32_as_u8(std::round(clampf(std::pow(ratio, 0.1f) * 255.0f * 0.18f, 0.0f, 255.0f)));
                }
            }
        }
    });
    return tab[dflt ? 0 : 1];
}

static void blue_lut(const uint8_t *lut_r, const uint8_t *lut_g, bool dflt, uint8_t *lut_b) {
    const uint8_t *pair = blue_pair_table(dflt);
    for (unsigned x1 = 0; x1 < 256; ++x1) {
        const uint8_t *row = pair + ((unsigned)lut_r[x1] << 8);
        uint8_t *out = lut_b + (x1 << 8);
        for (unsigned x2 = 0; x2 < 256; ++x2) out[x2] = row[lut_g[x2]];
        if (dflt) out[0] = 0; // `if b2 == 0 { blue = 0 }` on the raw band value (synthetic_rgb.rs:38-40)
    }
}

void synrgb_luts_default(uint8_t *luts) {
    uint8_t *lut_r = luts, *lut_g = luts + 256, *lut_b = luts + 512;
    for (unsigned v = 0; v < 256; ++v) { // synthetic_rgb.rs:22-29
        float vf = (float)v / 255.0f;
        lut_r[v] = f32_as_u8(clampf(std::round(std::pow(vf, 0.7f) * 255.0f), 0.0f, 255.0f));
        lut_g[v] = f32_as_u8(clampf(std::round(std::pow(vf, 0.9f) * 255.0f), 0.0f, 255.0f));
    }
    blue_lut(lut_r, lut_g, true, lut_b);
}

int synrgb_floor_from_hist(const uint64_t *hist, uint64_t n_per_band) {
    // synthetic_rgb.rs:92-113; the reference's counters are saturating u32
    uint32_t total = (uint32_t)(n_per_band + n_per_band);              // `as u32` wraps
    uint32_t target = as_u32(std::round((double)total * 0.05));
    uint32_t cumulative = 0;
    size_t floor_value = 0;
    for (size_t i = 0; i < 256; ++i) {
        uint64_t hi = std::min<uint64_t>(hist[i], UINT32_MAX);
        uint64_t c = (uint64_t)cumulative + hi;
        cumulative = c > UINT32_MAX ? UINT32_MAX : (uint32_t)c;
        if (cumulative >= target) { floor_value = i; break; }
    }
    return (int)std::min<size_t>(floor_value + 3, 40);
}

void synrgb_luts_suppressed(int fwc, uint8_t *luts) {
    uint8_t *lut_r = luts, *lut_g = luts + 256, *lut_b = luts + 512;
    float floor_f = (float)fwc;
    float denom = std::fmax(255.0f - floor_f, 1.0f); // synthetic_rgb.rs:119-120
    for (unsigned v = 0; v < 256; ++v) {              // :124-135
        if ((int)v <= fwc) { lut_r[v] = 0; lut_g[v] = 0; continue; }
        float shifted = ((float)v - floor_f) / denom;
        lut_r[v] = f32_as_u8(clampf(std::round(std::pow(shifted, 1.15f) * 255.0f), 0.0f, 255.0f));
        lut_g[v] = f32_as_u8(clampf(std::round(std::pow(shifted, 1.10f) * 255.0f), 0.0f, 255.0f));
    }
    blue_lut(lut_r, lut_g, false, lut_b);
}

const uint8_t *synrgb_blue_pair_supp() { return blue_pair_table(false); }

// The suppressed variant's blue as a PRODUCT (synthetic_rgb.rs:139-151): round(powf((r + 8) / (g + 8), 0.1) * 255 * 0.18) takes the values
// 32..65 only, and round-to-nearest of the f32 product P[r] * Q[g] with P[r] = f32(45.9 (r + 8)^0.1), Q[g] = f32((g + 8)^-0.1) equals the
// reference's table entry for ALL 65 536 (r, g) pairs -- checked here against blue_pair_table, entry by entry, every time the tables
// are built (the nearest a product comes to a rounding boundary is 1.4e-6, three f32 steps).  Returns nullptr when the check fails
// (another libm): the fused pass then keeps its 64-KB pair table.  out: P[256] | Q[256].
const float *synrgb_blue_factors_supp() {
    static float tab[512];
    static bool ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        for (unsigned i = 0; i < 256; ++i) {
            tab[i] = (float)(45.9 * std::pow((double)i + 8.0, 0.1));
            tab[256 + i] = (float)std::pow((double)i + 8.0, -0.1);
        }
        const uint8_t *pair = blue_pair_table(false);
        ok = true;
        for (unsigned r = 0; r < 256 && ok; ++r)
            for (unsigned g = 0; g < 256; ++g) {
                const float prod = tab[r] * tab[256 + g];
                if ((unsigned)std::nearbyintf(prod) != pair[(r << 8) | g]) { ok = false; break; } // (default rounding mode: to nearest even, as v_cvt_pk_u8_f32)
            }
    });
    return ok ? tab : nullptr;
}
const uint8_t *synrgb_blue_pair_default() { return blue_pair_table(true); }

const double *gamma_level_thresholds_u8() {
    static double tab[3][256];
    static std::once_flag once;
    std::call_once(once, [] {
        const double gammas[3] = {0.8, 0.9, 1.1};
        for (int g = 0; g < 3; ++g) {
            auto level = [&](double x) { return (int)as_u16(clampd(std::pow(x, gammas[g]) * 255.0, 0.0, 255.0)); };
            tab[g][0] = 0.0;
            uint64_t lo_bits = 0; // bit patterns of non-negative doubles order like the doubles
            for (int k = 1; k < 256; ++k) {
                uint64_t lo = lo_bits, hi;
                const double one = 1.0;
                std::memcpy(&hi, &one, 8);
                // invariant: level(lo) < k <= level(hi)   (level(0) = 0, level(1) = 255)
                while (hi - lo > 1) {
                    const uint64_t mid = lo + (hi - lo) / 2;
                    double xm;
                    std::memcpy(&xm, &mid, 8);
                    if (level(xm) >= k) hi = mid; else lo = mid;
                }
                std::memcpy(&tab[g][k], &hi, 8);
                lo_bits = lo;
            }
        }
    });
    return &tab[0][0];
}

const uint8_t *synrgb_supp_rg_tables() {
    static uint8_t tab[41][512];
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<uint8_t> luts(66048);
        for (int fwc = 0; fwc <= 40; ++fwc) {
            synrgb_luts_suppressed(fwc, luts.data());
            std::memcpy(tab[fwc], luts.data(), 512);
        }
    });
    return &tab[0][0];
}

void fold_compose_tables(const uint8_t *luts, int fwc, const uint8_t *resc1, const uint8_t *resc2,
                         uint8_t *tables) {
    const uint8_t *lut_r = luts, *lut_g = luts + 256, *lut_b = luts + 512;
    uint8_t *R2 = tables, *G2 = tables + 256, *B2 = tables + 512;
    for (unsigned v = 0; v < 256; ++v) {
        R2[v] = lut_r[resc1[v]];
        G2[v] = lut_g[resc2[v]];
    }
    for (unsigned v1 = 0; v1 < 256; ++v1) {
        unsigned r1 = resc1[v1];
        for (unsigned v2 = 0; v2 < 256; ++v2) {
            unsigned r2 = resc2[v2];
            bool water = fwc >= 0 && (int)r1 <= fwc && (int)r2 <= fwc; // synthetic_rgb.rs:161-166
            B2[(v1 << 8) | v2] = water ? 0 : lut_b[(r1 << 8) | r2];
        }
    }
    if (fwc >= 0) { // the short-circuit also zeroes R and G; lut_r/lut_g are already 0 at <= floor
        for (unsigned v = 0; v < 256; ++v) {
            if ((int)resc1[v] <= fwc) R2[v] = 0;
            if ((int)resc2[v] <= fwc) G2[v] = 0;
        }
    }
}

int stripe_plan(size_t rows, int nranks, size_t *row0, size_t *nrows) {
    if (nranks <= 0) return SARPRO_HIP_ERR_INVALID_ARG;
    for (int k = 0; k < nranks; ++k) {
        size_t a = (size_t)(((unsigned __int128)rows * (unsigned)k) / (unsigned)nranks);
        size_t b = (size_t)(((unsigned __int128)rows * (unsigned)(k + 1)) / (unsigned)nranks);
        row0[k] = a;
        nrows[k] = b - a;
    }
    return SARPRO_HIP_OK;
}

} // namespace sarpro

// ---------------- C ABI: host half ----------------
using namespace sarpro;

extern "C" {

int sarpro_hip_host_stats_from_dn_hist(const uint64_t dn_hist[65536], sarpro_hip_stats *out) {
    if (!dn_hist || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    return stats_from_dn_hist(dn_hist, out);
}

int sarpro_hip_host_window(sarpro_hip_stats *stats, int strategy, int tamed_synrgb) {
    if (!stats || tamed_synrgb < 0 || tamed_synrgb > 2) return SARPRO_HIP_ERR_INVALID_ARG;
    return select_window(stats, strategy, tamed_synrgb);
}

int sarpro_hip_host_level_lut_u16(const sarpro_hip_stats *stats, int bit_depth, int tamed_synrgb,
                                  uint16_t lut_out[65536]) {
    if (!stats || !lut_out || (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16))
        return SARPRO_HIP_ERR_INVALID_ARG;
    DnLut lut;
    build_level_lut_u16(*stats, bit_depth, tamed_synrgb, &lut);
    std::memcpy(lut_out, lut.full.data(), 65536 * sizeof(uint16_t));
    return SARPRO_HIP_OK;
}

int sarpro_hip_host_clahe_bin_lut_u16(const sarpro_hip_stats *stats, uint8_t lut_out[65536]) {
    if (!stats || !lut_out) return SARPRO_HIP_ERR_INVALID_ARG;
    DnLut lut;
    build_clahe_bin_lut_u16(*stats, &lut);
    for (int i = 0; i < 65536; ++i) lut_out[i] = (uint8_t)lut.full[i];
    return SARPRO_HIP_OK;
}

int sarpro_hip_host_clahe_cdfs(const uint64_t *tile_hists, size_t rows, size_t cols, double *cdfs_out) {
    if (!tile_hists || !cdfs_out) return SARPRO_HIP_ERR_INVALID_ARG;
    return clahe_cdfs(tile_hists, rows, cols, cdfs_out);
}

int sarpro_hip_host_u8_rescale_lut(unsigned min_level, unsigned max_level, uint8_t lut_out[256]) {
    if (!lut_out || min_level > 65535 || max_level > 65535) return SARPRO_HIP_ERR_INVALID_ARG;
    u8_rescale_lut(min_level, max_level, lut_out);
    return SARPRO_HIP_OK;
}

int sarpro_hip_host_synrgb_luts(int strategy, const uint64_t combined_hist[256], uint64_t n_per_band,
                                uint8_t *luts_out, int *floor_out) {
    if (!luts_out || strategy < 0 || strategy > SARPRO_STRATEGY_DEFAULT) return SARPRO_HIP_ERR_INVALID_ARG;
    if (strategy == SARPRO_STRATEGY_TAMED || strategy == SARPRO_STRATEGY_CLAHE) { // synthetic_rgb.rs:188-194
        if (!combined_hist) return SARPRO_HIP_ERR_INVALID_ARG;
        int fwc = synrgb_floor_from_hist(combined_hist, n_per_band);
        synrgb_luts_suppressed(fwc, luts_out);
        if (floor_out) *floor_out = fwc;
    } else {
        synrgb_luts_default(luts_out);
        if (floor_out) *floor_out = -1;
    }
    return SARPRO_HIP_OK;
}

int sarpro_hip_host_clahe_shape_ok(size_t rows, size_t cols) { return clahe_shape_ok(rows, cols) ? 1 : 0; }
int sarpro_hip_host_clahe_saturated_levels(size_t rows, size_t cols, uint8_t *col_class, uint8_t *row_bits) {
    if (!clahe_shape_ok(rows, cols) || !col_class || !row_bits) return SARPRO_HIP_ERR_INVALID_ARG;
    ClaheGeometry g;
    build_clahe_geometry(rows, cols, &g);
    std::vector<uint8_t> cc, rb;
    if (!clahe_saturated_levels(g, &cc, &rb)) return SARPRO_HIP_ERR_UNSUPPORTED_SHAPE;
    std::memcpy(col_class, cc.data(), cols);
    std::memcpy(row_bits, rb.data(), rows);
    return SARPRO_HIP_OK;
}

int sarpro_hip_host_stripe_plan(size_t rows, int nranks, size_t *row0_out, size_t *nrows_out) {
    if (!row0_out || !nrows_out) return SARPRO_HIP_ERR_INVALID_ARG;
    return stripe_plan(rows, nranks, row0_out, nrows_out);
}

} // extern "C"

// ======================= f32-input flavour: threshold tables =======================
namespace sarpro {

double db_of_f32(float v) { return 10.0 * std::log10(std::fmax((double)v, 1e-10)); }

static inline float bits_to_f32(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
static inline uint32_t f32_to_bits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }
static const uint32_t kMaxFiniteBits = 0x7F7FFFFFu; // FLT_MAX

// Smallest b in [lo, hi] with pred(b) (pred monotone false -> true over positive-float bit
// patterns, which order like the floats); hi + 1 when none.  `guess` only speeds it up.
template <typename P>
static uint32_t find_first_bits(P pred, uint32_t lo, uint32_t hi, uint32_t guess) {
    if (!pred(hi)) return hi + 1;
    if (pred(lo)) return lo;
    // invariant: !pred(lo) && pred(hi)
    uint32_t g = std::min(std::max(guess, lo + 1), hi);
    if (g != hi) {
        if (pred(g)) { // gallop down
            hi = g;
            uint32_t step = 1;
            while (hi - lo > step && pred(hi - step)) { hi -= step; step *= 2; }
            if (hi - lo > step) lo = hi - step;
        } else { // gallop up
            lo = g;
            uint32_t step = 1;
            while (hi - lo > step && !pred(lo + step)) { lo += step; step *= 2; }
            if (hi - lo > step) hi = lo + step;
        }
    }
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (pred(mid)) hi = mid; else lo = mid;
    }
    return hi;
}

// The same for a caller who knows pred(hi) holds and whose guess is good to a few steps (the thresholds of a step function
// that is uniform in dB: the previous threshold times a constant ratio): the guess is tried first, pred(lo) only if the search
// ends next to an untouched lo -- 3-4 evaluations of a libm logarithm instead of 6-7.
template <typename P>
static uint32_t find_first_bits_near(P pred, uint32_t lo, uint32_t hi, uint32_t guess) {
    if (lo >= hi) return pred(lo) ? lo : hi; // (pred(hi) holds)
    bool lo_false = false; // pred(lo) known to be false
    uint32_t g = std::min(std::max(guess, lo + 1), hi);
    if (g == hi || pred(g)) { // gallop down from g (pred(hi) holds by contract)
        hi = g;
        uint32_t step = 1;
        while (hi - lo > step && pred(hi - step)) { hi -= step; step *= 2; }
        if (hi - lo > step) { lo = hi - step; lo_false = true; }
    } else { // gallop up
        lo = g; lo_false = true;
        uint32_t step = 1;
        while (hi - lo > step && !pred(lo + step)) { lo += step; step *= 2; }
        if (hi - lo > step) hi = lo + step;
    }
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (pred(mid)) hi = mid; else { lo = mid; lo_false = true; }
    }
    if (!lo_false && pred(lo)) return lo;
    return hi;
}

float valid_threshold_f32() {
    static float thr = [] {
        uint32_t b = find_first_bits([](uint32_t x) { return db_of_f32(bits_to_f32(x)) > -50.0; }, 1u, kMaxFiniteBits,
                                     f32_to_bits(1e-5f));
        return bits_to_f32(b);
    }();
    return thr;
}

// Thresholds of a monotone integer-valued function `fn(db)` of the sample, for k = 1..n.
// inv(k) gives an approximate dB value where fn first reaches k (initial guess only).
template <typename F, typename G>
static void step_thresholds(F fn, G inv_db, int n, float *thr) {
    const uint32_t lo0 = f32_to_bits(valid_threshold_f32());
    thr[0] = 0.0f;
    const int64_t top = fn(db_of_f32(bits_to_f32(kMaxFiniteBits)));
    auto run = [&](int k0, int k1) { // thresholds k0..k1-1; the lower bracket restarts at the validity threshold
        uint32_t lo = lo0;
        double prev_db = 0.0, prev_step = 0.0, ratio = 0.0, prev_v = 0.0; // guess of threshold k: threshold k - 1 times 10^(dB step / 10)
        for (int k = k0; k < k1; ++k) {
            if ((int64_t)k > top) { thr[k] = INFINITY; continue; }
            const double gdb = inv_db(k);
            double gv;
            if (k > k0 && prev_v > 0.0) {
                const double step = gdb - prev_db;
                if (!(std::fabs(step - prev_step) < 1e-9)) { ratio = std::pow(10.0, step / 10.0); prev_step = step; }
                gv = prev_v * ratio;
            } else gv = std::pow(10.0, gdb / 10.0);
            prev_db = gdb;
            uint32_t guess = (gv > 0.0 && gv < 3.0e38) ? f32_to_bits((float)gv) : lo;
            // (k <= top: the predicate holds at the largest finite float)
            uint32_t b = find_first_bits_near([&](uint32_t x) { return fn(db_of_f32(bits_to_f32(x))) >= (int64_t)k; }, lo, kMaxFiniteBits, guess);
            thr[k] = bits_to_f32(b);
            prev_v = (b <= kMaxFiniteBits) ? (double)bits_to_f32(b) : 0.0;
            lo = b; // thresholds are non-decreasing in k
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = n >= 8192 ? (int)std::min<unsigned>(16u, std::max<unsigned>(1u, hw)) : 1; // the 65535-level tables
    if (nthreads <= 1) { run(1, n + 1); return; }
    std::vector<std::thread> pool;
    const int per = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        const int k0 = 1 + t * per, k1 = std::min(n + 1, k0 + per);
        if (k0 < k1) pool.emplace_back(run, k0, k1);
    }
    for (auto &th : pool) th.join();
}

void build_bin4096_thresholds(double min_db, double max_db, float *thr) {
    const double span = max_db - min_db, inv_span = 1.0 / span; // autoscale.rs:105-106
    auto idx = [=](double db) -> int64_t {
        double t = clampd((db - min_db) * inv_span, 0.0, 1.0);
        uint64_t i = as_u64(t * (double)kStatBins);
        return (int64_t)(i >= (uint64_t)kStatBins ? kStatBins - 1 : i);
    };
    step_thresholds(idx, [=](int k) { return min_db + span * ((double)k / kStatBins); }, kStatBins - 1, thr);
}

// the same index function and thresholds for a few bins only (the zone route of f32_path.cpp)
int bin4096_of_f32(float x, double min_db, double max_db) {
    const double span = max_db - min_db, inv_span = 1.0 / span;
    double t = clampd((db_of_f32(x) - min_db) * inv_span, 0.0, 1.0);
    uint64_t i = as_u64(t * (double)kStatBins);
    return (int)(i >= (uint64_t)kStatBins ? kStatBins - 1 : i);
}

void build_bin4096_thresholds_range(double min_db, double max_db, int k0, int k1, float *thr_out) {
    const double span = max_db - min_db;
    const uint32_t lo0 = f32_to_bits(valid_threshold_f32());
    const int top = bin4096_of_f32(bits_to_f32(kMaxFiniteBits), min_db, max_db);
    uint32_t lo = lo0;
    for (int k = k0; k <= k1; ++k) {
        if (k > top) { thr_out[k - k0] = INFINITY; continue; }
        const double gv = std::pow(10.0, (min_db + span * ((double)k / kStatBins)) / 10.0);
        const uint32_t guess = (gv > 0.0 && gv < 3.0e38) ? f32_to_bits((float)gv) : lo;
        const uint32_t b = find_first_bits([&](uint32_t x) { return bin4096_of_f32(bits_to_f32(x), min_db, max_db) >= k; }, lo, kMaxFiniteBits, guess);
        thr_out[k - k0] = bits_to_f32(b);
        lo = b;
    }
}

// autoscale.rs:120-140 from the three numbers it reads: the bin the target rank falls in, the count below the bin, the count in it
double percentile_from_bin(double min_db, double max_db, int bin, uint64_t target, uint64_t cum_below, uint64_t in_bin) {
    const double span = max_db - min_db;
    const uint64_t within = target >= cum_below ? target - cum_below : 0;
    const double frac = in_bin > 0 ? (double)within / (double)in_bin : 0.0;
    const double bin_width = span / (double)kStatBins;
    const double bin_start = min_db + (double)bin * bin_width;
    return bin_start + frac * bin_width;
}

uint64_t percentile_target(uint64_t n, double p) {
    uint64_t target = as_u64(std::floor(p * (double)n));
    return target >= n ? n - 1 : target;
}

void build_level_thresholds(const sarpro_hip_stats &s, int nlevels, float *thr) {
    const double lo = s.low_clip, hi = s.high_clip, g = s.gamma, max_val = (double)nlevels;
    const double range = std::fmax(hi - lo, 1.0);
    auto lvl = [=](double db) -> int64_t { return (int64_t)level_of_db(db, lo, hi, g, max_val); };
    step_thresholds(lvl, [=](int k) { return lo + range * std::pow((double)k / max_val, 1.0 / g); }, nlevels, thr);
}

// one entry of that table: the smallest f32 whose level is >= k
float level_threshold_one(const sarpro_hip_stats &s, int nlevels, int k) {
    const double lo = s.low_clip, hi = s.high_clip, g = s.gamma, max_val = (double)nlevels;
    const double range = std::fmax(hi - lo, 1.0);
    auto lvl = [=](double db) -> int64_t { return (int64_t)level_of_db(db, lo, hi, g, max_val); };
    if ((int64_t)k > lvl(db_of_f32(bits_to_f32(kMaxFiniteBits)))) return INFINITY;
    const double gv = std::pow(10.0, (lo + range * std::pow((double)k / max_val, 1.0 / g)) / 10.0);
    const uint32_t lo0 = f32_to_bits(valid_threshold_f32());
    const uint32_t guess = (gv > 0.0 && gv < 3.0e38) ? f32_to_bits((float)gv) : lo0;
    return bits_to_f32(find_first_bits([&](uint32_t x) { return lvl(db_of_f32(bits_to_f32(x))) >= (int64_t)k; }, lo0, kMaxFiniteBits, guess));
}

void build_clahe_bin_thresholds(const sarpro_hip_stats &s, float *thr) {
    const double lo = s.low_clip, hi = s.high_clip;
    const double range = std::fmax(hi - lo, 1.0);
    auto bin = [=](double db) -> int64_t { return (int64_t)clahe_bin_of_db(db, lo, hi); };
    step_thresholds(bin, [=](int k) { return lo + range * (((double)k - 0.5) / 255.0); }, kClaheBins - 1, thr);
}

} // namespace sarpro

// ======================= resize =======================
namespace sarpro {

static double sinc_pi(double x) { if (x == 0.0) return 1.0; x *= 3.14159265358979323846; return std::sin(x) / x; }
static double lanczos3(double x) { return (x >= -3.0 && x < 3.0) ? sinc_pi(x) * sinc_pi(x / 3.0) : 0.0; }

// the windows of build_resize_coeffs without their weights (the same expressions): which input indices output ox reads
void resize_bounds(uint32_t in_size, uint32_t out_size, uint32_t *window, std::vector<uint32_t> *start, std::vector<uint32_t> *size) {
    const double scale = (double)in_size / (double)out_size;
    const double radius = 3.0 * (scale > 1.0 ? scale : 1.0);
    *window = (uint32_t)std::ceil(radius) * 2 + 1;
    start->assign(out_size, 0);
    size->assign(out_size, 0);
    for (uint32_t ox = 0; ox < out_size; ++ox) {
        const double in_center = ((double)ox + 0.5) * scale;
        const uint32_t x_min = as_u32(std::fmax(std::floor(in_center - radius), 0.0));
        const uint32_t x_max = as_u32(std::fmin(std::ceil(in_center + radius), (double)in_size));
        (*start)[ox] = x_min;
        (*size)[ox] = x_max - x_min;
    }
}

void build_resize_coeffs(uint32_t in_size, uint32_t out_size, int elem_size, ResizeCoeffs *c) {
    c->in_size = in_size; c->out_size = out_size;
    const double scale = (double)in_size / (double)out_size;
    const double filter_scale = scale > 1.0 ? scale : 1.0; // adaptive kernel size
    const double radius = 3.0 * filter_scale;
    c->window = (uint32_t)std::ceil(radius) * 2 + 1;
    const double recip = 1.0 / filter_scale;
    c->start.assign(out_size, 0);
    c->size.assign(out_size, 0);
    std::vector<double> w((size_t)out_size * c->window, 0.0);
    double max_w = 0.0;
    for (uint32_t ox = 0; ox < out_size; ++ox) {
        const double in_center = ((double)ox + 0.5) * scale;
        const uint32_t x_min = as_u32(std::fmax(std::floor(in_center - radius), 0.0));
        const uint32_t x_max = as_u32(std::fmin(std::ceil(in_center + radius), (double)in_size));
        const double center = in_center - 0.5;
        double ww = 0.0;
        double *row = w.data() + (size_t)ox * c->window;
        for (uint32_t x = x_min; x < x_max; ++x) { row[x - x_min] = lanczos3(((double)x - center) * recip); ww += row[x - x_min]; }
        if (ww != 0.0) for (uint32_t x = x_min; x < x_max; ++x) row[x - x_min] /= ww;
        c->start[ox] = x_min;
        c->size[ox] = x_max - x_min;
    }
    for (double v : w) max_w = std::max(max_w, v);
    const int limit_bits = elem_size == 1 ? 15 : 31, max_precision = elem_size == 1 ? 22 : 45;
    int precision = 0;
    for (int cur = 0; cur < max_precision; ++cur) { // largest precision that keeps the largest weight in range
        precision = cur;
        if (std::round(max_w * (double)(1ll << (cur + 1))) >= (double)(1ll << limit_bits)) break;
    }
    c->precision = precision;
    const double fx = (double)(1ll << precision);
    c->k.assign((size_t)out_size * c->window, 0);
    for (uint32_t ox = 0; ox < out_size; ++ox)
        for (uint32_t t = 0; t < c->window; ++t)
            c->k[(size_t)t * out_size + ox] = (int32_t)std::round(w[(size_t)ox * c->window + t] * fx);
}

void resize_dimensions(size_t original_cols, size_t original_rows, size_t target_size, size_t *new_cols, size_t *new_rows) {
    const size_t short_side = std::min(original_rows, original_cols), long_side = std::max(original_rows, original_cols);
    if (target_size > long_side) { *new_cols = original_cols; *new_rows = original_rows; return; } // resize.rs:14-20
    const double scale_factor = (double)target_size / (double)long_side;
    const size_t new_short = (size_t)as_u64(std::round((double)short_side * scale_factor));
    if (original_cols > original_rows) { *new_cols = target_size; *new_rows = new_short; }
    else { *new_cols = new_short; *new_rows = target_size; }
}

} // namespace sarpro

// Built with -fsanitize=address,undefined by tests/test_native_sanitizers.py: the host half of the raster core
// (statistics, windows, per-DN tables, CDFs, thresholds, compose tables, resize coefficients, stripe plans) driven
// over degenerate and random inputs.  The float -> integer casts that restate Rust's saturating `as` are the main
// customers of UBSan here.  A clean exit is the test.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_logic.h"

using namespace sarpro;

static uint64_t s_ = 0x243F6A8885A308D3ull;
static uint64_t rnd() { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }

static void run_hist(std::vector<uint64_t> &h) {
    sarpro_hip_stats st;
    if (stats_from_dn_hist(h.data(), &st) != SARPRO_HIP_OK) std::abort();
    for (int strategy = 0; strategy <= SARPRO_STRATEGY_DEFAULT; ++strategy)
        for (int tamed = 0; tamed <= 2; ++tamed) {
            sarpro_hip_stats w = st;
            select_window(&w, strategy, tamed);
            DnLut lut;
            for (int depth = 0; depth <= 1; ++depth) build_level_lut_u16(w, depth, tamed, &lut);
            build_clahe_bin_lut_u16(w, &lut);
            if (w.valid_count) {
                std::vector<float> thr(65537);
                build_level_thresholds(w, 255, thr.data());
                build_clahe_bin_thresholds(w, thr.data());
                if (w.max_db > w.min_db) build_bin4096_thresholds(w.min_db, w.max_db, thr.data());
            }
        }
}

int main() {
    std::vector<uint64_t> h(65536);
    // degenerate histograms: empty, only no-data, one DN, two DNs, extremes, huge counts
    run_hist(h);
    h[0] = 12345; run_hist(h);
    h[777] = 1; run_hist(h);
    h[778] = 1ull << 40; run_hist(h);
    h[1] = 5; h[65535] = 7; run_hist(h);
    for (int rep = 0; rep < 6; ++rep) { // random supports of different density
        std::fill(h.begin(), h.end(), 0);
        const int n = 1 << (2 * rep + 1);
        for (int i = 0; i < n; ++i) h[rnd() % 65536] += rnd() % 100000;
        run_hist(h);
    }
    // 65535-level thresholds once (threaded path)
    {
        std::fill(h.begin(), h.end(), 0);
        for (int i = 1; i < 30000; ++i) h[i] = 1 + rnd() % 50;
        sarpro_hip_stats st;
        stats_from_dn_hist(h.data(), &st);
        select_window(&st, SARPRO_STRATEGY_STANDARD, 0);
        std::vector<float> thr(65537);
        build_level_thresholds(st, 65535, thr.data());
        for (int k = 2; k <= 65535; ++k) if (thr[k] < thr[k - 1]) std::abort(); // non-decreasing
    }
    // CLAHE: geometry for supported shapes, CDFs from random / zero / saturating tile histograms
    for (size_t rows : {42u, 43u, 100u, 257u, 1000u})
        for (size_t cols : {42u, 57u, 640u}) {
            if (!clahe_shape_ok(rows, cols)) continue;
            ClaheGeometry g;
            build_clahe_geometry(rows, cols, &g);
            std::vector<uint64_t> th(64 * 256);
            std::vector<double> cdfs(64 * 256);
            for (int mode = 0; mode < 3; ++mode) {
                for (auto &x : th) x = mode == 0 ? 0 : (mode == 1 ? rnd() % 1000 : (rnd() % 3 ? 0 : 0xFFFFFFFFFFull));
                if (clahe_cdfs(th.data(), rows, cols, cdfs.data()) != SARPRO_HIP_OK) std::abort();
                for (double c : cdfs) if (!(c >= 0.0 && c <= 1.0)) std::abort();
            }
        }
    // u8 rescale, synRGB tables, floor search, compose folding
    std::vector<uint8_t> luts(66048), tables(66048);
    uint8_t r1[256], r2[256];
    for (unsigned mn = 0; mn < 256; mn += 51)
        for (unsigned mx = mn; mx < 256; mx += 17) { u8_rescale_lut(mn, mx, r1); u8_rescale_lut(mx, mn, r2); }
    synrgb_luts_default(luts.data());
    for (int i = 0; i < 256; ++i) { r1[i] = (uint8_t)i; r2[i] = (uint8_t)(255 - i); }
    fold_compose_tables(luts.data(), -1, r1, r2, tables.data());
    uint64_t comb[256];
    for (int rep = 0; rep < 8; ++rep) {
        uint64_t total = 0;
        for (auto &c : comb) { c = rep == 0 ? 0 : (rnd() % (rep == 7 ? (1ull << 40) : 1000)); total += c; }
        const int fwc = synrgb_floor_from_hist(comb, total / 2);
        if (fwc < 0 || fwc > 40) std::abort();
        synrgb_luts_suppressed(fwc, luts.data());
        fold_compose_tables(luts.data(), fwc, r1, r2, tables.data());
    }
    (void)synrgb_supp_rg_tables(); (void)synrgb_blue_pair_supp(); (void)synrgb_blue_pair_default(); (void)gamma_level_thresholds_u8();
    // resize coefficient tables and dimension rules; stripe plans
    for (uint32_t in : {1u, 2u, 7u, 100u, 2048u, 20000u})
        for (uint32_t out : {1u, 3u, 64u, 1024u, 4096u})
            for (int esz = 1; esz <= 2; ++esz) { ResizeCoeffs c; build_resize_coeffs(in, out, esz, &c); }
    for (size_t c : {1u, 17u, 20000u}) for (size_t r : {1u, 23u, 16000u}) for (size_t t : {1u, 512u, 2048u}) { size_t nc, nr; resize_dimensions(c, r, t, &nc, &nr); }
    for (size_t rows : {0u, 1u, 7u, 8u, 20000u}) for (int n : {1, 2, 3, 8}) { size_t r0[8], nr[8]; stripe_plan(rows, n, r0, nr); }
    // f32 helpers at the edges of the float range
    for (float v : {0.0f, 1e-38f, 1e-10f, 1.0f, 3.4e38f, -1.0f}) (void)db_of_f32(v);
    (void)valid_threshold_f32();
    printf("host logic done\n");
    return 0;
}

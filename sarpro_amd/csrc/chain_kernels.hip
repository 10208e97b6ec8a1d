// chain_kernels.hip -- the host half of the CLAHE u8 path, moved onto the device so the headline
// chain (histogram -> statistics -> CLAHE bins -> CDFs -> apply -> rescale / synRGB tables ->
// compose) runs with NO host synchronisation between kernels.
//
// Everything here is +, -, *, /, sqrt, floor, round and integer<->float conversion in IEEE f64/f32
// (built with -ffp-contract=off, correctly rounded division): the results are bit-identical to
// host_logic.cpp, which remains the implementation for the other strategies, the f32 flavour and
// the row-stripe protocol.  Transcendentals never run here: the dB value of every DN, the powf
// tables of the suppressed synRGB LUTs (one per possible floor) and the blue pair tables are
// constant tables built once on the host with glibc and uploaded at context creation.
#include "chain_kernels.h"
#include "device_cdf.h"

#include <cfloat>

namespace sarpro {
namespace {

constexpr int kStatsBlock = 1024;      // k_chain_finish
constexpr int kTileDns = 256;          // DNs per wave-private staging tile of the statistics kernels

__device__ inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ------------------------------------------------------------------------------------
// compute_histogram_stats (autoscale.rs:35-160) over the DN histogram + the window of the strategy + the table
// of every DN (CLAHE bin: autoscale.rs:583-591, 262-265; u8 level: 440-442 / 649-651 / 734-736).
// Three kernels, kStatsParts workgroups per band each (a 65536-entry sweep by ONE workgroup runs at ~45 GB/s:
// 23 us; as one kernel the three dependent sweeps took 80-100 us):
//   A  partial count / min / max / dB moments of each 4096-DN slice; clears the scratch of B and C
//   B  the 4096-bin histogram (autoscale.rs:103-117): every workgroup reduces the partials to min / max, bins its
//      slice in LDS (run-length aggregated) and adds the occupied bins to the band's global bins
//   C  every workgroup scans the bins, inverts the 11 percentiles and selects the window (identical arithmetic in
//      all of them), then writes its slice of the table; workgroup 0 publishes the statistics
// The f64 sums are taken in a fixed order (thread-strided inside a slice, butterfly + wave order, slice order).
// ------------------------------------------------------------------------------------
constexpr int kStatsParts = kChainStatsParts;       // workgroups per band
constexpr int kPartBlock = 256;                    // threads per workgroup
constexpr int kPartDns = 65536 / kStatsParts;      // 4096 DNs per workgroup
constexpr int kPartWaveDns = kPartDns / (kPartBlock / 64);

template <typename T, typename Op>
__device__ T part_reduce(T v, T *scratch /*[>= 4]*/, Op op) { // deterministic reduction over kPartBlock threads
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = op(v, __shfl_xor(v, m, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = scratch[0];
#pragma unroll
    for (int w = 1; w < kPartBlock / 64; ++w) r = op(r, scratch[w]);
    return r;
}

struct BandTotals { unsigned long long count; uint32_t min_dn, max_dn; double sum1, sum2; };
__device__ BandTotals reduce_partials(const ChainStatsPartial *p) { // the same order in every workgroup
    BandTotals r{0ull, 0xFFFFFFFFu, 0u, 0.0, 0.0};
    for (int k = 0; k < kStatsParts; ++k) {
        r.count += p[k].count; r.min_dn = min(r.min_dn, p[k].min_dn); r.max_dn = max(r.max_dn, p[k].max_dn);
        r.sum1 += p[k].sum1; r.sum2 += p[k].sum2;
    }
    return r;
}

__global__ __launch_bounds__(kPartBlock) void k_chain_stats_a(ChainStatsArgs a) {
    __shared__ unsigned long long scr_u64[4];
    __shared__ double scr_f64[4];
    __shared__ uint32_t scr_u32[4];
    const int band = blockIdx.x, part = blockIdx.y, t = threadIdx.x;
    const unsigned long long *__restrict__ h = a.ghist + (size_t)band * 65536;
    const double *__restrict__ db = a.db;
    if (part == 0) { // scratch of the next two kernels
        for (int i = t; i < kStatBins; i += kPartBlock) a.bins4096[(size_t)band * kStatBins + i] = 0ull;
        if (a.level_hist) a.level_hist[(size_t)band * 256 + t] = 0ull; // levels mode: filled by kernel C; CLAHE: by the apply kernel
        if (a.sample_valid) { // CLAHE chain with a sampled histogram: every replica of the histogram and of the valid counts
            for (int r = 1; r < kSampleReplicas; ++r) a.level_hist[((size_t)r * kMaxBands + band) * 256 + t] = 0ull;
            if (t < kSampleReplicas) a.sample_valid[t * kMaxBands + band] = 0ull;
        }
        if (t == 0) { a.state[band].win_hi = 65535u; a.state[band].uncertain = 0u; }
    }
    unsigned long long cnt = 0;
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    double s1 = 0.0, s2 = 0.0;
    constexpr int kPer = kPartDns / kPartBlock; // 16 DNs per thread, all loads in flight at once
    unsigned long long hv[kPer];
    double dv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const uint32_t dn = part * kPartDns + k * kPartBlock + t; hv[k] = h[dn]; dv[k] = db[dn]; }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint32_t dn = part * kPartDns + k * kPartBlock + t;
        if (dn && hv[k]) {
            cnt += hv[k]; mn = min(mn, dn); mx = max(mx, dn);
            const double w = (double)hv[k];
            s1 += w * dv[k];
            s2 += w * dv[k] * dv[k];
        }
    }
    ChainStatsPartial r;
    r.count = part_reduce(cnt, scr_u64, [](unsigned long long x, unsigned long long y) { return x + y; });
    r.min_dn = part_reduce(mn, scr_u32, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
    r.max_dn = part_reduce(mx, scr_u32, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
    r.sum1 = part_reduce(s1, scr_f64, [](double x, double y) { return x + y; });
    r.sum2 = part_reduce(s2, scr_f64, [](double x, double y) { return x + y; });
    if (t == 0) a.partials[band * kStatsParts + part] = r;
}

__global__ __launch_bounds__(kPartBlock) void k_chain_stats_b(ChainStatsArgs a) {
    __shared__ unsigned long long hist[kStatBins];
    // a lane walks CONSECUTIVE DNs (run-length aggregation); the tables reach it through wave-private tiles so the
    // global reads stay coalesced: 256 DNs are loaded 4 per lane at stride 64 and consumed 4-contiguous per lane
    __shared__ unsigned long long tile_h[kPartBlock / 64][kTileDns];
    __shared__ double tile_d[kPartBlock / 64][kTileDns];
    const int band = blockIdx.x, part = blockIdx.y, t = threadIdx.x;
    const unsigned long long *__restrict__ h = a.ghist + (size_t)band * 65536;
    const double *__restrict__ db = a.db;
    const BandTotals tot = reduce_partials(a.partials + band * kStatsParts);
    if (tot.count == 0) return;
    const double min_db = db[tot.min_dn], max_db = db[tot.max_dn];
    if (fabs(max_db - min_db) < DBL_EPSILON) return; // autoscale.rs:81-100: no histogram in the degenerate case
    for (int i = t; i < kStatBins; i += kPartBlock) hist[i] = 0;
    __syncthreads();
    const double span = max_db - min_db, inv_span = 1.0 / span; // autoscale.rs:105-106
    // the bin index is monotone in DN, so equal indices form runs that are summed in a register and flushed with ONE
    // LDS atomic per run (bright DNs crowd into the top bins: per-DN atomics from 64 lanes would all hit one word)
    unsigned long long run = 0, run_idx = ~0ull;
    const int w = t >> 6, lane = t & 63;
    const uint32_t wave_base = (uint32_t)part * kPartDns + (uint32_t)w * kPartWaveDns;
    for (uint32_t tb = wave_base; tb < wave_base + kPartWaveDns; tb += kTileDns) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < kTileDns / 64; ++k) { tile_h[w][k * 64 + lane] = h[tb + k * 64 + lane]; tile_d[w][k * 64 + lane] = db[tb + k * 64 + lane]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < kTileDns / 64; ++m) {
            const uint32_t dn = tb + lane * (kTileDns / 64) + m;
            const unsigned long long hv = tile_h[w][lane * (kTileDns / 64) + m];
            if (!dn || !hv) continue;
            const double tt = clampd((tile_d[w][lane * (kTileDns / 64) + m] - min_db) * inv_span, 0.0, 1.0);
            unsigned long long idx = (unsigned long long)(tt * (double)kStatBins);
            if (idx >= (unsigned long long)kStatBins) idx = kStatBins - 1;
            if (idx != run_idx) {
                if (run) atomicAdd(&hist[run_idx], run);
                run = 0;
                run_idx = idx;
            }
            run += hv;
        }
    }
    if (run) atomicAdd(&hist[run_idx], run);
    __syncthreads();
    for (int i = t; i < kStatBins; i += kPartBlock)
        if (hist[i]) atomicAdd(&a.bins4096[(size_t)band * kStatBins + i], hist[i]);
}

__global__ __launch_bounds__(kPartBlock) void k_chain_stats_c(ChainStatsArgs a) {
    __shared__ unsigned long long scr_u64[4];
    __shared__ uint32_t scr_u32[4];
    __shared__ double pct[11];
    __shared__ unsigned long long lh[256];
    __shared__ unsigned long long tile_h[kPartBlock / 64][kTileDns];
    __shared__ double tile_d[kPartBlock / 64][kTileDns];
    const int band = blockIdx.x, part = blockIdx.y, t = threadIdx.x;
    const unsigned long long *__restrict__ h = a.ghist + (size_t)band * 65536;
    const double *__restrict__ db = a.db;
    ChainBandState *out = a.state + band;
    uint8_t *binlut = a.binlut + (size_t)band * a.binlut_stride;
    // CLAHE: this workgroup's dB values and its share of the 4096 bins are requested before anything depends on them (the kernel is a
    // chain of dependent round trips to L2: partials -> bins -> dB values; two of the three now overlap the first)
    constexpr int kPerC = kPartDns / kPartBlock, kOwnC = kStatBins / kPartBlock;
    double dv_pre[kPerC];
    unsigned long long own_pre[kOwnC];
    if (!a.levels_mode) {
#pragma unroll
        for (int k = 0; k < kPerC; ++k) dv_pre[k] = db[(uint32_t)part * kPartDns + k * kPartBlock + t];
    }
#pragma unroll
    for (int k = 0; k < kOwnC; ++k) own_pre[k] = a.bins4096[(size_t)band * kStatBins + kOwnC * t + k];
    const BandTotals tot = reduce_partials(a.partials + band * kStatsParts);
    const unsigned long long count = tot.count;

    sarpro_hip_stats st;
    st.valid_count = count;
    st.min_db = st.max_db = st.mean_db = st.std_db = st.median_db = 0.0;
    st.p01 = st.p02 = st.p05 = st.p10 = st.p25 = st.p75 = st.p90 = st.p95 = st.p98 = st.p99 = 0.0;
    st.low_clip = st.high_clip = 0.0;
    st.gamma = 1.0;
    st.skew_factor = st.tail_heaviness = 0.0;

    if (count == 0) { // no valid pixel: every DN is invalid, the table is irrelevant (autoscale.rs:466-468)
        for (uint32_t dn = (uint32_t)part * kPartDns + t; dn < (uint32_t)(part + 1) * kPartDns; dn += kPartBlock) binlut[dn] = 0;
        if (part == 0) {
            if (a.levels_mode == 1 && t == 0) a.level_hist[(size_t)band * 256] = a.total_px; // the other bins were cleared by kernel A
            if (a.levels_mode == 2) for (uint32_t dn = (uint32_t)part * kPartDns + t; dn < (uint32_t)(part + 1) * kPartDns; dn += kPartBlock) a.lut16[(size_t)band * 65536 + dn] = 0;
            if (t == 0) { out->stats = st; out->win_hi = 1; }
        }
        return;
    }
    const double min_db = db[tot.min_dn], max_db = db[tot.max_dn];
    // ---- mean / std (informational): var = E[x^2] - mean^2 ----
    const double mean = tot.sum1 / (double)count;
    st.min_db = min_db; st.max_db = max_db; st.mean_db = mean;
    st.std_db = count > 1 ? sqrt(fmax(tot.sum2 / (double)count - mean * mean, 0.0)) : 0.0;

    if (fabs(max_db - min_db) < DBL_EPSILON) { // autoscale.rs:81-100
        st.median_db = st.p01 = st.p02 = st.p05 = st.p10 = st.p25 = min_db;
        st.p75 = st.p90 = st.p95 = st.p98 = st.p99 = max_db;
    } else {
        const double span = max_db - min_db;
        // ---- exclusive prefix over the bins: thread t owns bins 16t .. 16t+15 ----
        constexpr int kOwn = kStatBins / kPartBlock;
        unsigned long long own[kOwn], totb = 0;
#pragma unroll
        for (int k = 0; k < kOwn; ++k) { own[k] = own_pre[k]; totb += own[k]; }
        unsigned long long excl;
        { // block exclusive scan (integers: order-free)
            const int lane = t & 63, wave = t >> 6;
            unsigned long long incl = totb;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned long long up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            __syncthreads();
            if (lane == 63) scr_u64[wave] = incl;
            __syncthreads();
            unsigned long long base = 0;
            for (int w = 0; w < wave; ++w) base += scr_u64[w];
            excl = base + incl - totb;
        }
        // ---- percentile inversion (autoscale.rs:120-140) ----
        const double ps[11] = {0.5, 0.01, 0.02, 0.05, 0.10, 0.25, 0.75, 0.90, 0.95, 0.98, 0.99};
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            unsigned long long target = (unsigned long long)floor(ps[q] * (double)count);
            if (target >= count) target = count - 1;
            unsigned long long e = excl;
#pragma unroll
            for (int k = 0; k < kOwn; ++k) {
                if (own[k] && target >= e && target < e + own[k]) {
                    const double frac = (double)(target - e) / (double)own[k];
                    const double bin_width = span / (double)kStatBins;
                    const double bin_start = min_db + (double)(kOwn * t + k) * bin_width;
                    pct[q] = bin_start + frac * bin_width;
                }
                e += own[k];
            }
        }
        __syncthreads();
        st.median_db = pct[0]; st.p01 = pct[1]; st.p02 = pct[2]; st.p05 = pct[3]; st.p10 = pct[4]; st.p25 = pct[5];
        st.p75 = pct[6]; st.p90 = pct[7]; st.p95 = pct[8]; st.p98 = pct[9]; st.p99 = pct[10];
    }
    const uint32_t dn0 = (uint32_t)part * kPartDns;
    if (!a.levels_mode) {
        // ---- CLAHE window + this workgroup's slice of the DN -> bin table ----
        const double low = st.p01, high = st.p99;
        st.low_clip = low; st.high_clip = high; st.gamma = 1.0;
        const double range = fmax(high - low, 1.0);
        uint32_t first_hi = 65535u; // first DN >= 1 whose dB value reached the high clip: the table is constant from there
        constexpr int kPer = kPartDns / kPartBlock;
        double dv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) dv[k] = dv_pre[k];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t dn = dn0 + k * kPartBlock + t;
            uint8_t bin = 0;
            if (dn) {
                const double d = dv[k];
                const double clipped = fmin(fmax(d, low), high);
                const double v = clampd((clipped - low) / range, 0.0, 1.0);
                long long b = (long long)round(v * 255.0);
                b = b < 0 ? 0 : (b > 255 ? 255 : b);
                bin = (uint8_t)b;
                if (d >= high) first_hi = min(first_hi, dn);
            }
            binlut[dn] = bin;
        }
        const uint32_t wh = part_reduce(first_hi, scr_u32, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
        if (t == 0) {
            if (wh != 65535u) atomicMin(&out->win_hi, wh);
            if (part == 0) out->stats = st;
        }
        return;
    }
    // ---- percentile strategies: window (autoscale.rs:404-429, 491-564, 721-729) ----
    {
        const double dynamic_range = st.max_db - st.min_db, iqr = st.p75 - st.p25;
        double low, high, gamma = 1.0;
        const int tamed = a.tamed_kind[band];
        if (tamed) {
            low = tamed == 1 ? fmin(st.p02, st.p05) : st.p05; high = st.p99;
        } else switch (a.strategy) {
        case SARPRO_STRATEGY_STANDARD:
            if (dynamic_range < 15.0) {
                const double range = fmax(20.0, dynamic_range * 0.8);
                low = st.median_db - range / 2.0; high = st.median_db + range / 2.0; gamma = 1.1;
            } else if (iqr < 5.0) {
                low = st.p25 - 2.5 * iqr; high = st.p75 + 2.5 * iqr;
            } else if (dynamic_range > 40.0) {
                low = fmax(st.p02, st.min_db + 0.02 * dynamic_range);
                high = fmin(st.p98, st.max_db - 0.02 * dynamic_range);
                gamma = 0.9;
            } else { low = st.p02; high = st.p98; }
            low = fmax(low, st.min_db);
            high = fmin(high, st.max_db);
            break;
        case SARPRO_STRATEGY_ROBUST: {
            const double thr = 2.5 * iqr;
            low = fmax(fmax(st.p25 - thr, st.p01), st.min_db);
            high = fmin(fmin(st.p75 + thr, st.p99), st.max_db);
            break;
        }
        case SARPRO_STRATEGY_ADAPTIVE: {
            const double skew = (st.mean_db - st.median_db) / fmax(fabs(st.std_db), 1.0);
            const double tail = (st.p99 - st.p95) / fmax(st.p95 - st.p75, 1.0);
            st.skew_factor = skew; st.tail_heaviness = tail;
            if (fabs(skew) > 0.5) {
                if (skew > 0.0) { low = st.p02; high = st.p98; gamma = 0.9; }
                else { low = st.p05; high = st.p95; gamma = 1.1; }
            } else if (tail > 2.0) { low = st.p10; high = st.p90; gamma = 0.8; }
            else { low = st.p05; high = st.p95; }
            break;
        }
        case SARPRO_STRATEGY_EQUALIZED: low = st.p01; high = st.p99; break;
        case SARPRO_STRATEGY_TAMED: low = st.p25; high = st.p99; break;
        default: low = st.p05; high = st.p95; break; // Default (Standard never reaches the advanced arm)
        }
        st.low_clip = low; st.high_clip = high; st.gamma = gamma;
    }
    if (a.levels_mode == 2) {
        // ---- u16 level of every DN of the slice (autoscale.rs:440-442 / 649-651 at max_val 65535; no rescale follows).
        //      gamma = 1: the same f64 operations as the host, exact.  gamma != 1: the device pow is a few ulp from
        //      glibc's (~1e-11 levels); a DN whose value comes within 1e-7 of a level boundary marks the band
        //      `uncertain` and the caller reruns it on the host-orchestrated route (odds ~1e-4 per band). ----
        const double low = st.low_clip, high = st.high_clip, gamma = st.gamma;
        const double range = fmax(high - low, 1.0);
        uint16_t *lut16 = a.lut16 + (size_t)band * 65536;
        uint32_t first_hi = 65535u, uncertain = 0u;
        constexpr int kPer = kPartDns / kPartBlock;
        double dv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) dv[k] = db[dn0 + k * kPartBlock + t];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t dn = dn0 + k * kPartBlock + t;
            uint32_t level = 0;
            if (dn) {
                const double d = dv[k];
                const double clipped = fmin(fmax(d, low), high);
                const double x = (clipped - low) / range;
                const double y = clampd((gamma == 1.0 ? x : pow(x, gamma)) * 65535.0, 0.0, 65535.0);
                level = (y == y) ? (uint32_t)y : 0u;
                // (x = 0 and x = 1 are exact in every libm: pow(0, g) = 0, pow(1, g) = 1)
                if (gamma != 1.0 && x > 0.0 && x < 1.0 && fabs(y - rint(y)) < 1e-7 && rint(y) >= 1.0) uncertain = 1u;
                if (d >= high) first_hi = min(first_hi, dn);
            }
            lut16[dn] = (uint16_t)level;
        }
        const uint32_t wh = part_reduce(first_hi, scr_u32, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
        const uint32_t unc = part_reduce(uncertain, scr_u32, [](uint32_t x, uint32_t y) { return x | y; });
        if (t == 0) {
            if (wh != 65535u) atomicMin(&out->win_hi, wh);
            if (unc) atomicOr(&out->uncertain, 1u);
            if (part == 0) out->stats = st;
        }
        return;
    }
    // ---- u8 level of every DN of the slice (autoscale.rs:440-442 / 649-651 / 734-736) + the level histogram.
    //      gamma != 1: trunc(pow(x, g) * 255) is resolved against host-built thresholds of x (glibc pow),
    //      so no pow runs here and the level is exactly the reference's. ----
    {
        const double low = st.low_clip, high = st.high_clip, gamma = st.gamma;
        const double range = fmax(high - low, 1.0);
        const double *gthr = gamma == 0.8 ? a.gamma_thr : (gamma == 0.9 ? a.gamma_thr + 256 : (gamma == 1.1 ? a.gamma_thr + 512 : nullptr));
        lh[t] = 0;
        __syncthreads();
        uint32_t first_hi = 65535u;
        unsigned long long run = 0;
        uint32_t run_level = 0xFFFFFFFFu;
        const int w = t >> 6, lane = t & 63;
        const uint32_t wave_base = dn0 + (uint32_t)w * kPartWaveDns;
        for (uint32_t tb = wave_base; tb < wave_base + kPartWaveDns; tb += kTileDns) { // a lane walks increasing DNs: levels form runs
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < kTileDns / 64; ++k) { tile_h[w][k * 64 + lane] = h[tb + k * 64 + lane]; tile_d[w][k * 64 + lane] = db[tb + k * 64 + lane]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            static_assert(kTileDns / 64 == 4, "one packed 4-byte table store per lane and tile");
            uint32_t lvl4[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t dn = tb + lane * 4 + m;
                uint32_t level = 0;
                if (dn) {
                    const double d = tile_d[w][lane * 4 + m];
                    const double clipped = fmin(fmax(d, low), high);
                    const double x = (clipped - low) / range;
                    if (gthr) {
                        uint32_t idx = 0; // number of k in 1..255 with x >= thr[k]  (NaN / negative x -> 0)
#pragma unroll
                        for (uint32_t step = 128; step; step >>= 1)
                            if (x >= gthr[idx + step]) idx += step;
                        level = idx;
                    } else {
                        const double y = clampd(x * 255.0, 0.0, 255.0);
                        level = (y == y) ? (uint32_t)y : 0u;
                    }
                    if (d >= high) first_hi = min(first_hi, dn);
                }
                lvl4[m] = level;
                const unsigned long long n = dn ? tile_h[w][lane * 4 + m] : (a.total_px - count); // DN = 0: every invalid pixel is level 0
                if (n) {
                    if (level != run_level) { if (run) atomicAdd(&lh[run_level], run); run = 0; run_level = level; }
                    run += n;
                }
            }
            // 4 consecutive table bytes in one store: a wave writes 256 contiguous bytes
            *reinterpret_cast<uint32_t *>(binlut + tb + lane * 4) = lvl4[0] | (lvl4[1] << 8) | (lvl4[2] << 16) | (lvl4[3] << 24);
        }
        if (run) atomicAdd(&lh[run_level], run);
        const uint32_t wh = part_reduce(first_hi, scr_u32, [](uint32_t x, uint32_t y) { return x < y ? x : y; });
        __syncthreads();
        if (lh[t]) atomicAdd(&a.level_hist[(size_t)band * 256 + t], lh[t]);
        if (t == 0) {
            if (wh != 65535u) atomicMin(&out->win_hi, wh);
            if (part == 0) out->stats = st;
        }
    }
}

// ------------------------------------------------------------------------------------
// clip / redistribute / CDF of every tile (autoscale.rs:271-302).  One block per (tile, band).
// The clip threshold is a multiple of 2^-7 and the counts are integers < 2^32, so every f64 sum
// below is exact and order-free.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chain_cdfs(const unsigned long long *__restrict__ tile_bins,
                                                    double *__restrict__ cdfs, uint32_t rows, uint32_t cols) {
    __shared__ double scr[256];
    __shared__ unsigned long long cum[256];
    const int tile = blockIdx.x, band = blockIdx.y, b = threadIdx.x;
    const size_t base = ((size_t)band * kTiles * kTiles + tile) * 256;
    const double c = clahe_tile_cdf_entry(tile_bins[base + b], tile, b, true, rows, cols, scr, cum);
    cdfs[base + b] = c;
}

// ------------------------------------------------------------------------------------
// After the apply pass: u8 rescale of each band (autoscale.rs:348-364), suppressed-synRGB floor
// from the combined histogram of the FINAL u8 bands (synthetic_rgb.rs:92-113) and the compose
// tables with both folded in (host_logic.cpp: fold_compose_tables).  Every block recomputes the
// (tiny) maps and builds its slice of the 65536-entry blue table; block 0 also publishes the maps.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kStatsBlock) void k_chain_finish(ChainFinishArgs a) {
    if (a.gate && a.gate->verdict == 0) return; // the speculative composition stands (k_chain_predict): nothing to finish
    __shared__ unsigned long long lh[2][256];
    __shared__ uint8_t resc[2][256];
    __shared__ unsigned long long combined[256];
    __shared__ int s_fwc;
    const int t = threadIdx.x;
    for (int i = t; i < 512; i += kStatsBlock) {
        const int b = i >> 8, k = i & 255;
        lh[b][k] = (b < a.nbands) ? a.level_hist[(size_t)b * 256 + k] : 0ull;
    }
    if (t < 256) combined[t] = 0;
    __syncthreads();
    // waves 0 and 1 take one band each (lane-parallel over the 256 levels, 4 per lane) -- as serial loops of one
    // thread per band these few hundred dependent steps were a third of the kernel
    __shared__ unsigned s_mn[2], s_mx[2];
    const int wb = t >> 6, ln = t & 63;
    if (wb < a.nbands) {
        unsigned long long v[4], others = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = lh[wb][ln * 4 + k]; if (ln * 4 + k) others += v[k]; }
        if (!a.levels_mode) { // level 0 is not counted by the apply kernel: it is what is left of the scene
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) others += __shfl_xor(others, m, 64);
            if (ln == 0) { v[0] = a.total_px - others; lh[wb][0] = v[0]; }
        }
        unsigned mn = 256u, mx = 0u; // lowest / highest occupied level (autoscale.rs:349-352 over the raster = over its histogram)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (v[k]) { mn = min(mn, (unsigned)(ln * 4 + k)); mx = max(mx, (unsigned)(ln * 4 + k)); }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, m, 64)); mx = max(mx, (unsigned)__shfl_xor((int)mx, m, 64)); }
        if (mn == 256u) mn = 0u; // empty histogram
        if (ln == 0) { s_mn[wb] = mn; s_mx[wb] = mx; }
        const float fmn = (float)mn, fmx = (float)mx;
        const float scale = fmx > fmn ? 255.0f / (fmx - fmn) : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned x = ln * 4 + k;
            float val = roundf(((float)x - fmn) * scale);
            val = val < 0.0f ? 0.0f : (val > 255.0f ? 255.0f : val);
            resc[wb][x] = a.no_rescale[wb] ? (uint8_t)x : (uint8_t)val;
        }
    }
    __syncthreads();
    if (t < 256) for (int b = 0; b < a.nbands; ++b) atomicAdd(&combined[resc[b][t]], lh[b][t]);
    const bool lead = blockIdx.x == 0;
    if (lead && t < 512 && a.resc_out) a.resc_out[t] = (t >> 8) < a.nbands ? resc[t >> 8][t & 255] : (uint8_t)(t & 255);
    if (lead && wb < a.nbands && a.identity_out) {
        bool ident = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (lh[wb][ln * 4 + k] && resc[wb][ln * 4 + k] != ln * 4 + k) ident = false;
        const bool all = __builtin_amdgcn_ballot_w64(!ident) == 0ull;
        if (ln == 0) a.identity_out[wb] = all ? 1 : 0;
    }
    __syncthreads();
    if (a.levels_mode) // DN -> level tables become DN -> final u8 tables; the compose tables then take final values
        for (int i = blockIdx.x * kStatsBlock + t; i < 2 * 65536; i += gridDim.x * kStatsBlock) {
            const int b = i >> 16;
            if (b < a.nbands) { uint8_t *p = a.dn_tables + (size_t)b * a.dn_table_stride + (i & 65535); *p = resc[b][*p]; }
        }
    if (a.nbands < 2 || !a.tables) return;
    if (t == 0 && a.suppressed) { // synthetic_rgb.rs:99-113 with the reference's saturating u32 counters
        const uint32_t total = (uint32_t)(a.total_px + a.total_px);
        const double tc = round((double)total * 0.05);
        const uint32_t target = tc >= 4294967295.0 ? 4294967295u : (uint32_t)tc;
        uint32_t cumulative = 0;
        int floor_value = 0;
        for (int i = 0; i < 256; ++i) {
            const unsigned long long hi = combined[i] > 4294967295ull ? 4294967295ull : combined[i];
            const unsigned long long c = (unsigned long long)cumulative + hi;
            cumulative = c > 4294967295ull ? 4294967295u : (uint32_t)c;
            if (cumulative >= target) { floor_value = i; break; }
        }
        const int fwc = floor_value + 3 < 40 ? floor_value + 3 : 40;
        s_fwc = fwc;
        if (lead && a.floor_out) *a.floor_out = fwc;
    }
    __syncthreads();
    const int fwc = a.suppressed ? s_fwc : -1;
    const uint8_t *lut_r = a.suppressed ? a.supp_rg + (size_t)fwc * 512 : a.default_rg, *lut_g = lut_r + 256; // host-built powf tables
    const uint8_t *pair = a.suppressed ? a.blue_pair_supp : a.blue_pair_default;
    uint8_t *R2 = a.tables, *G2 = a.tables + 256, *B2 = a.tables + 512;
    const bool fold = !a.levels_mode; // levels mode: the values that reach the compose tables are already final
    if (lead && t < 256) {
        const int r1 = fold ? resc[0][t] : t, r2 = fold ? resc[1][t] : t;
        R2[t] = r1 <= fwc ? 0 : lut_r[r1];
        G2[t] = r2 <= fwc ? 0 : lut_g[r2];
    }
    for (int i = blockIdx.x * kStatsBlock + t; i < 65536; i += gridDim.x * kStatsBlock) {
        const int v1 = i >> 8, v2 = i & 255;
        const int r1 = fold ? resc[0][v1] : v1, r2 = fold ? resc[1][v2] : v2;
        const bool water = r1 <= fwc && r2 <= fwc;                 // suppressed only (fwc = -1 otherwise)
        const bool b2zero = !a.suppressed && r2 == 0;             // default variant: `if b2 == 0 { blue = 0 }` (synthetic_rgb.rs:38-40)
        B2[i] = (water || b2zero) ? 0 : pair[((size_t)lut_r[r1] << 8) | lut_g[r2]];
    }
}

// ------------------------------------------------------------------------------------
// Speculation on the two global quantities that stand between the CLAHE blend and the composition (dual-pol u8 scene on one
// device).  The apply pass counted its level histogram on SAMPLED rows only (kernels.hip, HIST == 2: the per-pixel LDS atomics
// were 15 % of the dominant kernel), so the histogram is an estimate.  This kernel turns it into
//   * a PROOF that the u8 rescale of autoscale.rs:348-364 is the identity: a level that occurs in the sample occurs in the
//     raster; with level 0 (any invalid pixel of the scene, or a sampled valid one) and level 255 present in both bands,
//     min = 0 and max = 255;
//   * a PREDICTION F of the suppressed-synRGB floor (synthetic_rgb.rs:99-113): per band the invalid pixels (level 0, their
//     number is exact: pixels - valid) plus the sampled valid pixels' level counts scaled by valid / sampled valid;
// and builds the compose tables for (identity, F).  The compose pass then counts the band-pixels below F and F + 1 exactly
// while it composes and accepts or refutes F (kernels.hip, k_compose_u8<16, true>); refuted, or without the proof, the gated
// exact kernels run: level recount -> k_chain_finish -> composition.  The raster is the reference's either way; the
// prediction only decides which kernels produce it.  Every block computes the (tiny) prediction and builds its slice of the
// blue table; block 0 publishes the state.
//   A band WITHOUT level 0 (a crop with no invalid pixel whose darkest cells start above the CDF's foot) used to have no proof and
// took the exact kernels.  With allow_rescaled its lowest sampled level becomes a third prediction: the rescale (min_pred, 255) is
// folded into the tables, the floor is predicted on the rescaled levels, and the fused pass also counts the level bytes below
// min_pred -- any such byte refutes (spec_ok = kSpecRescaled; kernels.hip, clahe_rgb_fused_body<true>).  Level 255 stays a proof.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kStatsBlock) void k_chain_predict(ChainPredictArgs a) {
    // The second launch (behind a fused pass): only after an undercut lowest level whose true value the pass recorded, on a chain that
    // has not had its second pass yet (nothing this kernel writes is read by this test -- `retried` is the retry kernel's).
    if (a.second && !(a.spec->spec_ok == kSpecRescaled && a.spec->verdict == 1u && !a.spec->pool_overflow && a.spec->retry_min == 1u &&
                      !a.spec->retried && !(a.spec->force & kSpecForceNoRetry))) return;
    __shared__ double est[2][256], fin[2][256]; // per band: estimated pixels per level, per FINAL level (after the predicted rescale)
    __shared__ uint8_t resc[2][256];
    __shared__ int s_ok[2], s_fwc, s_f;
    __shared__ unsigned s_mn[2];
    const int t = threadIdx.x, wb = t >> 6, ln = t & 63;
    if (t < 512) (&fin[0][0])[t] = 0.0;
    if (wb < 2) {
        const unsigned long long *sh = a.sample_hist + (size_t)wb * 256;
        unsigned long long v[4], others = 0, sv = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = 0ull;
            for (int r = 0; r < kSampleReplicas; ++r) v[k] += sh[(size_t)r * kMaxBands * 256 + ln * 4 + k]; // the sampling pass's replicas (kernels.h)
            if (ln * 4 + k) others += v[k];
        }
        for (int r = 0; r < kSampleReplicas; ++r) sv += a.spec->sample_valid_rep[r * kMaxBands + wb];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) others += __shfl_xor(others, m, 64);
        const unsigned long long valid = a.state[wb].stats.valid_count;
        if (blockIdx.x == 0 && ln == 0) a.spec->sample_valid[wb] = sv;
        const unsigned long long invalid = a.total_px - valid;
        // A band without a valid sample is level 0 everywhere (autoscale.rs:466-468 returns the zero raster) and scale_u16_to_u8 of a
        // constant raster has max == min, scale 1.0: the identity (autoscale.rs:356).  Nothing needs a sample to be proven; its
        // estimate is exact (every band-pixel at level 0: `invalid` below is the whole band).
        const bool empty_band = valid == 0ull;
        const bool consistent = sv >= others && sv > 0;
        const unsigned long long s0v = consistent ? sv - others : 0ull; // sampled valid pixels at level 0
        const bool has255 = __builtin_amdgcn_ballot_w64(ln == 63 && v[3] != 0ull) != 0ull;
        const bool has0 = invalid > 0ull || s0v > 0ull;
        // the band's lowest level: 0 is PROVEN by has0 (a level of the sample is a level of the raster); without it the lowest
        // sampled level is a PREDICTION that the fused pass verifies (no byte below it) -- a.allow_rescaled
        unsigned mn = 256u;
#pragma unroll
        for (int k = 3; k >= 0; --k) if (v[k] && (ln * 4 + k)) mn = (unsigned)(ln * 4 + k);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, m, 64));
        if (has0 || empty_band || mn == 256u) mn = 0u;
        if ((a.force & kSpecForceMinMispredict) && mn && mn < 255u) mn += 1u; // (a level the raster undercuts: the verification must refute it)
        bool min_is_exact = false;
        if (a.second) { // the level the first pass found below the prediction IS the band's lowest (it looked at every pixel)
            const unsigned tm = a.spec->true_min[wb]; // (recorded by the pass; cleared by the FIRST launch only: no block of this one writes it)
            if (tm < mn) { mn = tm; min_is_exact = true; }
        }
        const double scale = consistent ? (double)valid / (double)sv : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = ln * 4 + k;
            est[wb][l] = l ? scale * (double)v[k] : (double)invalid + scale * (double)s0v;
        }
        // autoscale.rs:348-364 with (min, max) = (mn, 255), as k_chain_finish computes it from the exact histogram
        const float fmn = (float)mn, fmx = 255.0f;
        const float rs = fmx > fmn ? 255.0f / (fmx - fmn) : 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned x = ln * 4 + k;
            float val = roundf(((float)x - fmn) * rs);
            val = val < 0.0f ? 0.0f : (val > 255.0f ? 255.0f : val);
            resc[wb][x] = (uint8_t)val;
        }
        if (ln == 0) {
            s_mn[wb] = mn;
            s_ok[wb] = (empty_band || (consistent && has255 && (mn == 0u ? (has0 || min_is_exact) : a.allow_rescaled != 0u))) ? 1 : 0;
        }
    }
    __syncthreads();
    // final-level estimates: from mn on the rescale is strictly increasing (its slope is >= 1), so every final level has ONE source
    // per band -- plain stores, no order of additions that could differ between two ranks of a stripe group
    if (t < 512) {
        const int b = t >> 8, x = t & 255;
        if ((unsigned)x >= s_mn[b]) fin[b][resc[b][x]] = est[b][x];
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t total = (uint32_t)(a.total_px + a.total_px);
        const double tc = round((double)total * 0.05);
        const uint32_t target = tc >= 4294967295.0 ? 4294967295u : (uint32_t)tc;
        double cum = 0.0, lt0 = 0.0, lt1 = 0.0;
        int f = kSpecFloorCap;
        for (int i = 0; i < kSpecFloorCap; ++i) {
            cum += fin[0][i] + fin[1][i];
            if (cum >= (double)target) { f = i; break; }
        }
        bool ok = s_ok[0] && s_ok[1];
        if (a.force & kSpecForceNoSpec) ok = false;
        if (a.force & kSpecForceMispredict) f = f < kSpecFloorCap ? f + 1 : kSpecFloorCap - 1;
        if (a.force & kSpecForceMispredict2) f = f + 2 <= kSpecFloorCap ? f + 2 : (f >= 2 ? f - 2 : f); // (two levels off: the second pass's floor is refuted too)
        for (int i = 0; i <= f && i < 64; ++i) { if (i < f) lt0 += fin[0][i] + fin[1][i]; lt1 += fin[0][i] + fin[1][i]; }
        s_fwc = f + 3 < 40 ? f + 3 : 40;
        s_f = f;
        if (blockIdx.x == 0) {
            ChainSpecState *sp = a.spec;
            sp->spec_ok = !ok ? 0u : (s_mn[0] | s_mn[1]) ? kSpecRescaled : kSpecIdentity;
            sp->verdict = 1u; // until the speculative composition has verified the floor
            sp->floor_pred = f;
            sp->done = 0u;
            sp->n_lt[0] = 0ull; sp->n_lt[1] = 0ull; sp->n_below_min = 0ull;
            sp->target = target;
            sp->min_pred[0] = s_mn[0]; sp->min_pred[1] = s_mn[1];
            sp->est_lt[0] = lt0; sp->est_lt[1] = lt1;
            sp->force = a.force;
            sp->pool_overflow = 0u;
            sp->next_item = 0u;
            sp->retry_floor = -1; sp->retry_armed = a.second ? 1u : 0u;
            if (!a.second) { sp->retried = 0u; sp->floor_first = f; sp->retry_min = 0u; }
            if (!a.second) { sp->true_min[0] = 256u; sp->true_min[1] = 256u; }
            for (int i = 0; i < 256; ++i) (&sp->below_hist[0][0])[i] = 0ull;
            if (a.floor_out) *a.floor_out = s_fwc; // stands iff the verdict accepts; k_chain_finish rewrites it otherwise
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && wb < 2) { // per band the lowest level (from mn on) whose final value reaches F, F + 1: what the fused pass counts against
        const unsigned mn = s_mn[wb];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned thr = 256u;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const unsigned x = ln * 4 + k;
                if (x >= mn && (int)resc[wb][x] >= s_f + q) thr = x;
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) thr = min(thr, (unsigned)__shfl_xor((int)thr, m, 64));
            if (ln == 0) a.spec->thr[wb][q] = thr;
        }
    }
    if (blockIdx.x == 0) {
        if (t < 512) { a.exact_hist[t] = 0ull; if (a.resc_out) a.resc_out[t] = resc[t >> 8][t & 255]; }
        if (t < 2 && a.identity_out) a.identity_out[t] = s_mn[t] == 0u ? 1 : 0;
    }
    const int fwc = s_fwc;
    const uint8_t *lut_r = a.supp_rg + (size_t)fwc * 512, *lut_g = lut_r + 256;
    uint8_t *R2 = a.tables, *G2 = a.tables + 256, *B2 = a.tables + 512;
    if (blockIdx.x == 0 && t < 256) {
        const int r1 = resc[0][t], r2 = resc[1][t];
        const uint8_t r = r1 <= fwc ? 0 : lut_r[r1], g = r2 <= fwc ? 0 : lut_g[r2];
        R2[t] = r;
        G2[t] = g;
        if (a.blue_pq && a.blue_by_level) { a.blue_by_level[t] = a.blue_pq[r]; a.blue_by_level[256 + t] = a.blue_pq[256 + g]; }
    }
    for (int i = blockIdx.x * kStatsBlock + t; i < 65536; i += gridDim.x * kStatsBlock) {
        const int r1 = resc[0][i >> 8], r2 = resc[1][i & 255];
        B2[i] = (r1 <= fwc && r2 <= fwc) ? 0 : a.blue_pair_supp[((size_t)lut_r[r1] << 8) | lut_g[r2]];
    }
}

// The second chance of a refuted floor (round 6): the fused pass's own counts say on which side of the prediction the floor lies
// (ChainSpecState::retry_floor, written with the verdict); this kernel rebuilds what k_chain_predict built for the first floor --
// the compose tables, the rescaled form's thresholds -- for that floor, clears the pass's counters and arms the retry kernel.  It
// returns at once on every scene whose first pass stood (and after a refuted lowest level, a pool overflow, SPEC_FORCE = noretry):
// the exact kernels behind it are gated on the verdict as before.
__global__ __launch_bounds__(kStatsBlock) void k_chain_repredict(ChainRepredictArgs a) {
    ChainSpecState *sp = a.spec;
    // (nothing this kernel writes is read by this test: every block decides the same)
    if (!sp->spec_ok || sp->verdict == 0u || sp->pool_overflow || sp->retry_floor < 0 || (sp->force & kSpecForceNoRetry)) {
        // row stripes: the retry's all-reduce of the counts is enqueued on every scene; without a second pass it must sum zeros
        // (the counts are the ranks' sums already), and k_spec_verdict puts them back
        if (a.stripes && blockIdx.x == 0 && threadIdx.x == 0) {
            sp->saved_counts[0] = sp->n_lt[0]; sp->saved_counts[1] = sp->n_lt[1]; sp->saved_counts[2] = sp->n_below_min;
            sp->n_lt[0] = 0ull; sp->n_lt[1] = 0ull; sp->n_below_min = 0ull;
            for (int i = 0; i < 256; ++i) (&sp->below_hist[0][0])[i] = 0ull; // (read by the first verdict only; the second all-reduce sums the whole block)
        }
        return;
    }
    __shared__ uint8_t resc[2][256];
    const int t = threadIdx.x, wb = t >> 6, ln = t & 63;
    if (t < 512) resc[t >> 8][t & 255] = a.resc_in[t];
    __syncthreads();
    const int f = sp->retry_floor, fwc = f + 3 < 40 ? f + 3 : 40;
    if (blockIdx.x == 0 && wb < 2) { // per band the lowest level (from min_pred on) whose final value reaches F, F + 1 (as k_chain_predict)
        const unsigned mn = sp->min_pred[wb];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned thr = 256u;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const unsigned x = ln * 4 + k;
                if (x >= mn && (int)resc[wb][x] >= f + q) thr = x;
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) thr = min(thr, (unsigned)__shfl_xor((int)thr, m, 64));
            if (ln == 0) sp->thr[wb][q] = thr;
        }
    }
    const uint8_t *lut_r = a.supp_rg + (size_t)fwc * 512, *lut_g = lut_r + 256;
    uint8_t *R2 = a.tables, *G2 = a.tables + 256, *B2 = a.tables + 512;
    if (blockIdx.x == 0 && t < 256) {
        const int r1 = resc[0][t], r2 = resc[1][t];
        const uint8_t r = r1 <= fwc ? 0 : lut_r[r1], g = r2 <= fwc ? 0 : lut_g[r2];
        R2[t] = r;
        G2[t] = g;
        if (a.blue_pq && a.blue_by_level) { a.blue_by_level[t] = a.blue_pq[r]; a.blue_by_level[256 + t] = a.blue_pq[256 + g]; }
    }
    for (int i = blockIdx.x * kStatsBlock + t; i < 65536; i += gridDim.x * kStatsBlock) {
        const int r1 = resc[0][i >> 8], r2 = resc[1][i & 255];
        B2[i] = (r1 <= fwc && r2 <= fwc) ? 0 : a.blue_pair_supp[((size_t)lut_r[r1] << 8) | lut_g[r2]];
    }
    if (blockIdx.x == 0 && t == 0) {
        sp->floor_pred = f;
        sp->done = 0u;
        sp->n_lt[0] = 0ull; sp->n_lt[1] = 0ull; sp->n_below_min = 0ull;
        for (int i = 0; i < 256; ++i) (&sp->below_hist[0][0])[i] = 0ull;
        sp->next_item = 0u;
        sp->retry_armed = 1u;
        if (a.floor_out) *a.floor_out = fwc; // stands iff the second verdict accepts; k_chain_finish rewrites it otherwise
    }
}

// In-place u8 remap with the map in device memory (per-band u8 outputs of the chain).
__global__ __launch_bounds__(256) void k_chain_remap(const uint8_t *src, size_t src_pitch, uint8_t *dst,
                                                     size_t dst_pitch, uint32_t rows, uint32_t cols,
                                                     const uint8_t *__restrict__ map, const uint8_t *__restrict__ skip) {
    if (skip && *skip && src == dst) return; // identity map, in place: nothing to do
    __shared__ uint8_t m[256];
    m[threadIdx.x] = map[threadIdx.x];
    __syncthreads();
    const uint64_t total = (uint64_t)rows * cols;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
        const uint32_t r = (uint32_t)(idx / cols), c = (uint32_t)(idx - (uint64_t)r * cols);
        dst[(size_t)r * dst_pitch + c] = m[src[(size_t)r * src_pitch + c]];
    }
}

// ------------------------------------------------------------------------------------
// Partial level histogram of the speculative apply kernel (kernels.hip, PARTIAL_HIST): bins 1..63 are exact, the
// pixels at levels >= 64 were added in bulk to SOME bins >= 64 such that their total and the highest occupied level
// are those of the true histogram.  What k_chain_finish derives from the histogram is unchanged by that:
//   * bin 0 = pixels - sum(other bins): the sum is preserved;
//   * the u8 rescale range (autoscale.rs:348-364) is the lowest and the highest occupied level: the highest is
//     preserved; the lowest is read from the exact bins PROVIDED some level <= kGuardMinLevel (27) is occupied;
//   * the synRGB floor (synthetic_rgb.rs:99-113) is the first final level f whose cumulative count reaches 5 %, and
//     only matters below 37 (floor + 3 is capped at 40).  The rescale maps level x >= 64 to round((x - lo) * 255 /
//     (hi - lo)) >= 64 - lo >= 37 when lo <= 27, so every final level below 37 only collects exact bins: the search
//     either stops at the same f < 37, or runs past 36 in both histograms, where the cap makes the result 40;
//   * "the rescale is the identity on the occupied levels" <=> lo = 0 and hi = 255 (or a single level 0): lo, hi only.
// This guard checks the proviso; when no level <= 27 is occupied (never on natural data: CLAHE output starts at the
// CDF's foot) it clears the histogram and raises the flag that makes k_level_hist_if_flagged recount the level raster.
// ------------------------------------------------------------------------------------
constexpr int kGuardMinLevel = 27;
__global__ __launch_bounds__(256) void k_level_hist_guard(unsigned long long *level_hist, unsigned long long total_px,
                                                          uint32_t *flags) {
    __shared__ unsigned long long part[256];
    __shared__ int low_occupied;
    unsigned long long *lh = level_hist + (size_t)blockIdx.x * 256;
    const int t = threadIdx.x;
    const unsigned long long v = t ? lh[t] : 0ull;
    part[t] = v;
    if (t == 0) low_occupied = 0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) part[t] += part[t + s];
        __syncthreads();
    }
    if (t >= 1 && t <= kGuardMinLevel && v) low_occupied = 1;
    if (t == 0 && total_px > part[0]) low_occupied = 1; // level 0 is occupied
    __syncthreads();
    const bool recount = !low_occupied;
    if (t == 0) flags[blockIdx.x] = recount ? 1u : 0u;
    if (recount) lh[t] = 0ull;
}

__global__ __launch_bounds__(256) void k_level_hist_if_flagged(LevelRecountArgs a) {
    const int band = blockIdx.y;
    if (a.gate ? a.gate->verdict == 0 : !a.flags[band]) return;
    const uint8_t *__restrict__ in = a.levels[band];
    const size_t pitch = a.pitch;
    const uint32_t rows = a.rows, cols = a.cols;
    unsigned long long *__restrict__ hist = a.level_hist + (size_t)band * 256;
    __shared__ uint32_t h[4][256]; // one histogram per wave: a quarter of the collisions on the crowded levels
    for (int i = threadIdx.x; i < 1024; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *hw = h[threadIdx.x >> 6];
    if (pitch % 16 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) { // 16 levels per load, a wave reads 1 KiB of a row
        const uint32_t vpr = (cols + 15) / 16;
        const uint64_t total = (uint64_t)rows * vpr;
        for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
            const uint32_t r = (uint32_t)(idx / vpr), c = (uint32_t)(idx - (uint64_t)r * vpr) * 16;
            const uint4 q = *reinterpret_cast<const uint4 *>(in + (size_t)r * pitch + c);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            const uint32_t nvalid = min(16u, cols - c); // the last vector of a row may reach into the pitch padding
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if ((uint32_t)k < nvalid) atomicAdd(&hw[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu], 1u);
        }
    } else {
        const uint64_t total = (uint64_t)rows * cols;
        for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
            const uint32_t r = (uint32_t)(idx / cols), c = (uint32_t)(idx - (uint64_t)r * cols);
            atomicAdd(&hw[in[(size_t)r * pitch + c]], 1u);
        }
    }
    __syncthreads();
    const uint32_t n = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (n && threadIdx.x) atomicAdd(&hist[threadIdx.x], (unsigned long long)n); // bin 0 stays implied, as the apply kernel leaves it
}

} // namespace

hipError_t launch_chain_stats(const ChainStatsArgs &a, int nbands, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_stats_a, dim3(nbands, kStatsParts), dim3(kPartBlock), 0, s, a);
    hipLaunchKernelGGL(k_chain_stats_b, dim3(nbands, kStatsParts), dim3(kPartBlock), 0, s, a);
    hipLaunchKernelGGL(k_chain_stats_c, dim3(nbands, kStatsParts), dim3(kPartBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_chain_cdfs(const unsigned long long *tile_bins, double *cdfs, uint32_t rows, uint32_t cols, int nbands,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_chain_cdfs, dim3(kTiles * kTiles, nbands), dim3(256), 0, s, tile_bins, cdfs, rows, cols);
    return hipGetLastError();
}
hipError_t launch_chain_finish(const ChainFinishArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_finish, dim3((a.nbands == 2 && a.tables) || a.levels_mode ? 64 : 1), dim3(kStatsBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_chain_repredict(const ChainRepredictArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_repredict, dim3(64), dim3(kStatsBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_chain_predict(const ChainPredictArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_predict, dim3(64), dim3(kStatsBlock), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_chain_remap(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, uint32_t rows,
                              uint32_t cols, const uint8_t *d_map, const uint8_t *d_skip_flag, hipStream_t s) {
    if (!rows || !cols) return hipSuccess;
    const uint64_t want = ((uint64_t)rows * cols + 255) / 256;
    hipLaunchKernelGGL(k_chain_remap, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, src, src_pitch, dst,
                       dst_pitch, rows, cols, d_map, d_skip_flag);
    return hipGetLastError();
}

} // namespace sarpro

namespace sarpro {
hipError_t launch_level_hist_guard(unsigned long long *level_hist, unsigned long long total_px, int nbands, uint32_t *d_flags,
                                   hipStream_t s) {
    hipLaunchKernelGGL(k_level_hist_guard, dim3(nbands), dim3(256), 0, s, level_hist, total_px, d_flags);
    return hipGetLastError();
}
hipError_t launch_level_hist_if_flagged(const LevelRecountArgs &a, int nbands, hipStream_t s) {
    if (!a.rows || !a.cols) return hipSuccess;
    hipLaunchKernelGGL(k_level_hist_if_flagged, dim3(1024, nbands), dim3(256), 0, s, a);
    return hipGetLastError();
}
} // namespace sarpro

// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SAR raster core.
//
// All of these are HBM-bound integer / gather work (no MFMA: the path is pointwise +
// histogram).  Common shape: 64-lane waves read 16 B per lane (8 u16 samples) from
// row-major rasters, so one wave instruction covers 1 KiB of a row; per-pixel
// transcendental math is replaced by host-built tables that live in LDS.
//
// Built with -ffp-contract=off: the CLAHE blend must round like the reference's separate
// f64 multiplies and adds (autoscale.rs:327-329), never as FMAs.
#include "kernels.h"
#include "chain_kernels.h"
#include "device_cdf.h"

#include <algorithm>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

namespace sarpro {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // 4 waves: one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;

template <int VEC> struct U16Vec;
template <> struct U16Vec<8> {
    uint4 v;
    __device__ static U16Vec load(const uint16_t *p) { U16Vec r; r.v = *reinterpret_cast<const uint4 *>(p); return r; }
    __device__ static U16Vec load_stream(const uint16_t *p) { // read-once data: non-temporal
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
        U16Vec r; r.v = make_uint4(t.x, t.y, t.z, t.w); return r;
    }
    __device__ uint32_t get(int j) const {
        uint32_t w = j < 2 ? v.x : (j < 4 ? v.y : (j < 6 ? v.z : v.w));
        return (j & 1) ? (w >> 16) : (w & 0xFFFFu);
    }
};
template <> struct U16Vec<1> {
    uint32_t v;
    __device__ static U16Vec load(const uint16_t *p) { U16Vec r; r.v = *p; return r; }
    __device__ uint32_t get(int) const { return v; }
};

// 16-byte store of data the GPU does not read again (the RGB raster): non-temporal, so that 1.2 GB of output per scene do not
// sit dirty in L2 / Infinity Cache and get written back under the NEXT kernel -- measured on the headline loop: the DN-histogram
// pass that follows the composition of the previous scene 0.36 -> 0.33 ms, the compose pass itself unchanged.
__device__ __forceinline__ void store_stream16(uint4 *p, const uint4 &t) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u tv; tv.x = t.x; tv.y = t.y; tv.z = t.z; tv.w = t.w;
    __builtin_nontemporal_store(tv, reinterpret_cast<v4u *>(p));
}

__device__ __forceinline__ void store_stream8(uint2 *p, const uint2 &t) {
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    v2u tv; tv.x = t.x; tv.y = t.y;
    __builtin_nontemporal_store(tv, reinterpret_cast<v2u *>(p));
}

// Per-lane keep-mask for an 8-sample vector whose columns [col, col+8) may straddle [c0, c1).
struct EdgeMask {
    uint32_t m[4];
    __device__ EdgeMask(int col, int c0, int c1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ca = col + 2 * k, cb = ca + 1;
            m[k] = ((ca >= c0 && ca < c1) ? 0x0000FFFFu : 0u) | ((cb >= c0 && cb < c1) ? 0xFFFF0000u : 0u);
        }
    }
    __device__ void apply(U16Vec<8> &v) const { v.v.x &= m[0]; v.v.y &= m[1]; v.v.z &= m[2]; v.v.w &= m[3]; }
    __device__ bool keep(int j) const { return (m[j >> 1] >> (16 * (j & 1))) & 1u; }
};

// The dynamic LDS block of this kernel starts at LDS address 0 (it holds no static __shared__), so a byte offset
// IS the address: going through the `extern __shared__` symbol makes the compiler add its (zero) link-time
// address to every computed LDS address, one VALU instruction per access.
#define LDS_AT(T, off) (*reinterpret_cast<__attribute__((address_space(3))) T *>((uint32_t)(off)))

__device__ inline int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

// ------------------------------------------------------------------------------------
// 1. Per-tile DN histogram (pass 1).  One work item = a column strip x row range inside
//    one CLAHE tile.  Counts of DN < lds_bins are privatised in LDS (ds_add_u32), the
//    bright tail goes straight to the tile's global histogram, DN = 0 (no-data) is counted
//    in a register so the no-data wedge does not serialise on one LDS word.
//    Algorithmic traffic: 2 B/px read.
// ------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_dn_hist_u16(DnHistArgs a) {
    extern __shared__ uint32_t lds_hist[];
    const Rect rc = a.rects[blockIdx.x];
    const int band = blockIdx.y;
    const uint16_t *__restrict__ in = a.in[band];
    uint32_t *__restrict__ gh = a.tile_hist[band] + (size_t)rc.id[0] * 65536u;
    const uint32_t W = a.lds_bins;

    for (uint32_t i = threadIdx.x; i < W; i += kBlock) lds_hist[i] = 0;
    __syncthreads();

    const int col = rc.cstart + lane_id() * VEC;
    const bool lane_on = col < rc.c1 && col + VEC > rc.c0;
    uint32_t zeros = 0;

    auto consume = [&](const U16Vec<VEC> &v) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = col + j;
            if (c >= rc.c0 && c < rc.c1) {
                const uint32_t d = v.get(j);
                if (d == 0) ++zeros;
                else if (d < W) atomicAdd(&lds_hist[d], 1u);
                else atomicAdd(&gh[d], 1u);
            }
        }
    };

    if (lane_on) {
        const int step = kWavesPerBlock;
        int r = rc.r0 + wave_id();
        // two rows in flight per wave: the loads of row r+step issue before row r is consumed
        for (; r + step < rc.r1; r += 2 * step) {
            const U16Vec<VEC> v0 = U16Vec<VEC>::load(in + (size_t)r * a.pitch + col);
            const U16Vec<VEC> v1 = U16Vec<VEC>::load(in + (size_t)(r + step) * a.pitch + col);
            consume(v0);
            consume(v1);
        }
        if (r < rc.r1) consume(U16Vec<VEC>::load(in + (size_t)r * a.pitch + col));
    }
    if (zeros) atomicAdd(&lds_hist[0], zeros); // slot 0 is otherwise unused (d == 0 never lands in LDS)
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < W; i += kBlock) {
        const uint32_t n = lds_hist[i];
        if (n) atomicAdd(&gh[i], n);
    }
}

// ------------------------------------------------------------------------------------
// 1b. The same histogram for 16-byte-vector rasters, the form the 400 MP scene runs.  No per-pixel branches: each pixel does ONE
//     unconditional ds_add_u32 -- DN in [1, lds_bins) to its bin, everything else (DN = 0 and the
//     bright tail) to a per-lane dummy word, so the no-data wedge neither serialises on one LDS
//     word nor costs an exec-mask round trip.  Bright-tail pixels are flagged in a bitmask and
//     added to the tile's global histogram by a rarely taken branch per row.  The DN = 0 count is
//     not accumulated at all: it is (pixels of the tile) - (sum of the other bins), restored on
//     the host.  Loads are software-pipelined: the next two rows are in flight while two rows are
//     consumed (4 KiB per wave).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dn_hist_u16_interior(DnHistArgs a) {
    constexpr int VEC = 8;
    extern __shared__ uint32_t lds_hist[];
    const Rect rc = a.rects[blockIdx.x];
    const int band = blockIdx.y;
    const uint16_t *__restrict__ in = a.in[band];
    uint32_t *__restrict__ gh = a.tile_hist[band] + (size_t)rc.id[0] * 65536u;
    const uint32_t W = a.lds_bins;

    for (uint32_t i = threadIdx.x; i < W + kWave; i += kBlock) lds_hist[i] = 0;
    __syncthreads();

    const int col = rc.cstart + lane_id() * VEC;
    const uint32_t dummy = (W + (uint32_t)lane_id()) * 4u; // byte offset of this lane's dummy word
    unsigned char *lds_bytes = reinterpret_cast<unsigned char *>(lds_hist);
    // lanes that straddle the item's edge zero their out-of-range samples at load (4 ANDs per row):
    // DN = 0 lands in the dummy word and is never counted
    const EdgeMask em(col, rc.c0, rc.c1);

    auto consume = [&](const U16Vec<VEC> &v) {
        uint32_t big = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint32_t d = v.get(j);
            const bool in_lds = d - 1u < W - 1u; // 1 <= d < W
            big |= (d >= W ? 1u : 0u) << j;
            const uint32_t off = in_lds ? d * 4u : dummy;
            atomicAdd(reinterpret_cast<uint32_t *>(lds_bytes + off), 1u);
        }
        if (big) { // bright tail: rare
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if ((big >> j) & 1u) atomicAdd(&gh[v.get(j)], 1u);
        }
    };

    if (col < rc.c1 && col + VEC > rc.c0) {
        const int step = kWavesPerBlock;
        const uint16_t *p = in + col;
        int r = rc.r0 + wave_id();
        // loads are unconditional (row index clamped to the item's last row) so the compiler can keep two
        // rows in flight behind a counted vmcnt; rows past the end are loaded redundantly but not consumed
        if (r < rc.r1) {
            const int last = rc.r1 - 1;
            U16Vec<VEC> a0 = U16Vec<VEC>::load(p + (size_t)r * a.pitch);
            U16Vec<VEC> a1 = U16Vec<VEC>::load(p + (size_t)min(r + step, last) * a.pitch);
            for (; r < rc.r1; r += 2 * step) {
                const U16Vec<VEC> b0 = U16Vec<VEC>::load(p + (size_t)min(r + 2 * step, last) * a.pitch);
                const U16Vec<VEC> b1 = U16Vec<VEC>::load(p + (size_t)min(r + 3 * step, last) * a.pitch);
                em.apply(a0);
                em.apply(a1);
                consume(a0);
                if (r + step < rc.r1) consume(a1);
                a0 = b0;
                a1 = b1;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x + 1; i < W; i += kBlock) { // bin 0 is restored on the host
        const uint32_t n = lds_hist[i];
        if (n) atomicAdd(&gh[i], n);
    }
}

// ------------------------------------------------------------------------------------
// 1c. The whole-raster (untiled) DN histogram of the percentile strategies as a LINEAR sweep: persistent workgroups,
//     each wave takes 1024 consecutive pixels of a row (2 KiB: two 1-KiB load instructions) at a time in launch order,
//     the next chunk in flight while this one is counted.  Same counting as 1b (one unconditional ds_add_u32 per
//     pixel, DN = 0 and the bright tail to a per-lane dummy word, tail pixels to the global histogram).  The strip
//     walk of 1b exists for the CLAHE tiles; without tiles the in-order sweep reads faster.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dn_hist_u16_linear(DnHistArgs a, uint32_t rows, uint32_t cols) {
    extern __shared__ uint32_t lds_hist[];
    const int band = blockIdx.y;
    const uint16_t *__restrict__ in = a.in[band];
    uint32_t *__restrict__ gh = a.tile_hist[band];
    const uint32_t W = a.lds_bins;
    for (uint32_t i = threadIdx.x; i < W + kWave; i += kBlock) lds_hist[i] = 0;
    __syncthreads();
    const int lane = lane_id();
    const uint32_t dummy = (W + (uint32_t)lane) * 4u;
    unsigned char *lds_bytes = reinterpret_cast<unsigned char *>(lds_hist);
    auto consume = [&](const uint4 &q) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        uint32_t big = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t d = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu);
            const bool in_lds = d - 1u < W - 1u; // 1 <= d < W
            big |= (d >= W ? 1u : 0u) << j;
            const uint32_t off = in_lds ? d * 4u : dummy;
            atomicAdd(reinterpret_cast<uint32_t *>(lds_bytes + off), 1u);
        }
        if (big) { // bright tail: rare
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((big >> j) & 1u) atomicAdd(&gh[(j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu)], 1u);
        }
    };
    // chunk c of a row: pixels [c * 1024, c * 1024 + 1024); lane l owns pixels l*8 .. l*8+7 of each 512-pixel half
    const uint32_t cpr = (cols + 1023) / 1024;
    const uint64_t chunks = (uint64_t)rows * cpr, nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
    const uint32_t last_col = a.pitch >= 8 ? (uint32_t)a.pitch - 8 : 0u; // clamp: masked lanes load inside the row's pitch
    auto load = [&](uint64_t ch, int half) {
        const uint32_t r = (uint32_t)(ch / cpr), col = (uint32_t)(ch - (uint64_t)r * cpr) * 1024 + half * 512 + lane * 8;
        return *reinterpret_cast<const uint4 *>(in + (size_t)r * a.pitch + min(col, last_col));
    };
    uint64_t ch = (uint64_t)blockIdx.x * kWavesPerBlock + wave_id();
    if (ch < chunks) {
        uint4 n0 = load(ch, 0), n1 = load(ch, 1);
        for (; ch < chunks; ch += nwaves) {
            uint4 q0 = n0, q1 = n1;
            const uint64_t nx = ch + nwaves < chunks ? ch + nwaves : ch;
            n0 = load(nx, 0); n1 = load(nx, 1);
            const uint32_t r = (uint32_t)(ch / cpr), c0 = (uint32_t)(ch - (uint64_t)r * cpr) * 1024 + lane * 8;
            if (c0 + 520 > cols) { // the row's last chunk: samples past the last column become DN = 0 (never counted)
                const EdgeMask m0((int)c0, 0, (int)cols), m1((int)c0 + 512, 0, (int)cols);
                q0.x &= m0.m[0]; q0.y &= m0.m[1]; q0.z &= m0.m[2]; q0.w &= m0.m[3];
                q1.x &= m1.m[0]; q1.y &= m1.m[1]; q1.z &= m1.m[2]; q1.w &= m1.m[3];
            }
            consume(q0);
            consume(q1);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x + 1; i < W; i += kBlock) { // bin 0 is restored by the consumer
        const uint32_t n = lds_hist[i];
        if (n) atomicAdd(&gh[i], n);
    }
}

// ------------------------------------------------------------------------------------
// 2. Sum the per-tile histograms into the band's global 65536-bin histogram (u64).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sum_tile_hists(SumTileHistArgs a, int ntiles) {
    uint32_t *__restrict__ th = a.tile_hist[blockIdx.y];
    const uint32_t dn = (blockIdx.x * kBlock + threadIdx.x) * 4; // 4 consecutive DNs per thread: 16-byte loads
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 16
    for (int t = 0; t < ntiles; ++t) { // independent loads: many in flight
        const uint4 v = *reinterpret_cast<const uint4 *>(th + (size_t)t * 65536u + dn);
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        if (a.clear && (v.x | v.y | v.z | v.w)) *reinterpret_cast<uint4 *>(th + (size_t)t * 65536u + dn) = make_uint4(0u, 0u, 0u, 0u);
    }
    unsigned long long *o = a.out[blockIdx.y] + dn;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
}

// ------------------------------------------------------------------------------------
// 3. Per-tile 256-bin CLAHE histogram from the per-tile DN histogram and the DN -> bin table
//    (autoscale.rs:259-268 without touching the pixels again).  One block per tile.
// ------------------------------------------------------------------------------------
constexpr int kTileBinBlock = 1024; // one workgroup sweeps a tile's 65536 counters: the sweep is latency-bound, so a wide one
__global__ __launch_bounds__(kTileBinBlock) void k_tile_bin_hist(TileBinHistArgs a) {
    __shared__ unsigned long long h[256];
    const int band = blockIdx.y;
    uint32_t *__restrict__ t = a.tile_hist[band] + (size_t)blockIdx.x * 65536u;
    const uint8_t *__restrict__ binlut = a.binlut[band];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = 0; base < 65536u; base += 16 * kTileBinBlock) { // 32 independent loads in flight per thread
        uint32_t n[16], bin[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t dn = base + k * kTileBinBlock + threadIdx.x; n[k] = t[dn]; bin[k] = binlut[dn]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t dn = base + k * kTileBinBlock + threadIdx.x;
            if (n[k] && dn) atomicAdd(&h[bin[k]], (unsigned long long)n[k]); // DN = 0 is invalid: not counted
            if (a.clear && n[k]) t[dn] = 0u; // only the occupied bins are written: a few thousand of the 65536 per tile
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) a.out[band][(size_t)blockIdx.x * 256u + threadIdx.x] = h[threadIdx.x];
    if (a.cdfs_out[band]) { // the tile's CDF in the same launch (no reduction over ranks in between: one kernel boundary less on the chain)
        __shared__ double scr[256];
        __shared__ unsigned long long cum[256];
        const bool active = threadIdx.x < 256;
        const int b = active ? (int)threadIdx.x : 0;
        const double c = clahe_tile_cdf_entry(active ? h[b] : 0ull, (int)blockIdx.x, b, active, a.rows, a.cols, scr, cum);
        if (active) a.cdfs_out[band][(size_t)blockIdx.x * 256u + b] = c;
    }
}

// ------------------------------------------------------------------------------------
// 4. CLAHE apply (pass 2): DN -> bin (windowed table in LDS) -> bilinear blend of the four
//    neighbouring tile CDFs in exact IEEE f64 (autoscale.rs:307-330) -> level
//    (autoscale.rs:600-606), plus the 256-bin histogram of the u8 levels that the u8 rescale
//    (autoscale.rs:348-364) and the suppressed-synRGB floor (synthetic_rgb.rs:92-113) need.
//    One work item = a 64*VEC-column strip x row range inside one interpolation cell, so the
//    four CDFs are fixed per block and the per-column weights (dx, 1-dx) stay in registers;
//    the per-row weights (dy, 1-dy) are wave-uniform.
//    Algorithmic traffic: 2 B/px read + 1 (u8) or 2 (u16) B/px written.
// ------------------------------------------------------------------------------------
#ifndef SARPRO_U16_CDF_COPIES
#define SARPRO_U16_CDF_COPIES 2
#endif
template <int VEC, bool OUT16>
__global__ __launch_bounds__(kBlock) void k_clahe_apply_u16(ClaheApplyArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // the four CDFs of a bin as TWO 16-byte halves in two arrays, (c00, c01) at [bin] and (c10, c11) at [256 + bin]: with one 32-byte
    // entry per bin each of the two 16-byte reads of a sample could only ever land on half of the LDS banks (entry stride 32 B: the
    // first halves on bank quads 0, 2, 4, 6, the second halves on the odd ones) -- 70 % of the pass's LDS cycles were bank conflicts
    // ... and kCopies copies of both arrays, chosen by lane: the eight lanes a 16-byte LDS read serves per cycle pick among eight bank
    // quads at random -- half of them on each copy collide less
    constexpr int kCopies = SARPRO_U16_CDF_COPIES;
    double2 *cdf2 = reinterpret_cast<double2 *>(lds_raw) + (kCopies > 1 ? ((threadIdx.x >> 2) & (kCopies - 1)) * 512 : 0); // [copy][2][256]
    uint32_t *lds_hist = reinterpret_cast<uint32_t *>(lds_raw + kCopies * 256 * 4 * 8);  // [256]
    uint8_t *lds_lut = lds_raw + kCopies * 256 * 4 * 8 + 256 * 4;                        // window

    const Rect rc = a.rects[blockIdx.x];
    const int band = blockIdx.y;
    const uint16_t *__restrict__ in = a.in[band];
    const double *__restrict__ cdfs = a.cdfs[band];
    const uint8_t *__restrict__ glut = a.binlut[band];
    // chain mode: the window is only known on the device (k_chain_stats); the table is constant from win_hi on
    const uint32_t win_lo = a.dev_state ? 0u : a.win_lo[band], win_hi = a.dev_state ? a.dev_state[band].win_hi : a.win_hi[band];
    const bool lut_lds = a.dev_state ? win_hi < a.lut_cap : a.lut_in_lds != 0;
    unsigned long long *ghist = a.level_hist[band];

    {   // stage the four CDFs of every bin
        const int b = threadIdx.x; // kBlock == 256 bins
        const double2 ct = make_double2(cdfs[(size_t)rc.id[0] * 256 + b], cdfs[(size_t)rc.id[1] * 256 + b]);
        const double2 cb = make_double2(cdfs[(size_t)rc.id[2] * 256 + b], cdfs[(size_t)rc.id[3] * 256 + b]);
#pragma unroll
        for (int k = 0; k < kCopies; ++k) {
            reinterpret_cast<double2 *>(lds_raw)[k * 512 + b] = ct;
            reinterpret_cast<double2 *>(lds_raw)[k * 512 + 256 + b] = cb;
        }
        lds_hist[b] = 0;
        if (lut_lds)
            for (uint32_t i = threadIdx.x; i <= win_hi - win_lo; i += kBlock) lds_lut[i] = glut[win_lo + i];
    }
    __syncthreads();

    const int col = rc.cstart + lane_id() * VEC;
    const bool lane_on = col < rc.c1 && col + VEC > rc.c0;
    const bool full = col >= rc.c0 && col + VEC <= rc.c1;
    uint32_t zeros = 0;

    double dx[VEC], omdx[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = col + j;
        const bool in_rng = c >= rc.c0 && c < rc.c1;
        const RowWeight w = a.col_w[in_rng ? c : rc.c0];
        dx[j] = w.d;
        omdx[j] = w.omd;
    }

    auto process_row = [&](int r, const U16Vec<VEC> &v) {
        const RowWeight rw = a.row_w[a.row_off + r]; // wave-uniform
        const double dy = rw.d, omdy = rw.omd;
        uint32_t lv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint32_t d = v.get(j);
            const uint32_t dc = min(max(d, win_lo), win_hi);
            const uint32_t bin = lut_lds ? (uint32_t)lds_lut[dc - win_lo] : (uint32_t)glut[dc];
            const double2 ct = cdf2[bin], cb = cdf2[256 + bin];
            const double top = ct.x * omdx[j] + ct.y * dx[j];
            const double bottom = cb.x * omdx[j] + cb.y * dx[j];
            double o = top * omdy + bottom * dy;
            o = fmin(fmax(o, 0.0), 1.0);
            const uint32_t level = (uint32_t)(o * a.max_val); // truncation, o*max_val in [0, max_val]
            lv[j] = d ? level : 0u;                           // invalid (DN = 0) -> 0
        }
        if (!OUT16 && ghist) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = col + j;
                if (c >= rc.c0 && c < rc.c1) {
                    if (lv[j] == 0) ++zeros;
                    else atomicAdd(&lds_hist[lv[j]], 1u);
                }
            }
        }
        if (OUT16) {
            uint16_t *o16 = reinterpret_cast<uint16_t *>(a.out[band]) + (size_t)r * a.out_pitch + col;
            if (VEC == 8 && full) {
                uint4 pk;
                pk.x = lv[0] | (lv[1] << 16); pk.y = lv[2] | (lv[3] << 16);
                pk.z = lv[4] | (lv[5] << 16); pk.w = lv[6] | (lv[7] << 16);
                *reinterpret_cast<uint4 *>(o16) = pk;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (col + j >= rc.c0 && col + j < rc.c1) o16[j] = (uint16_t)lv[j];
            }
        } else {
            uint8_t *o8 = reinterpret_cast<uint8_t *>(a.out[band]) + (size_t)r * a.out_pitch + col;
            if (VEC == 8 && full) {
                uint2 pk;
                pk.x = lv[0] | (lv[1] << 8) | (lv[2] << 16) | (lv[3] << 24);
                pk.y = lv[4] | (lv[5] << 8) | (lv[6] << 16) | (lv[7] << 24);
                *reinterpret_cast<uint2 *>(o8) = pk;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (col + j >= rc.c0 && col + j < rc.c1) o8[j] = (uint8_t)lv[j];
            }
        }
    };

    if (lane_on) {
        const int step = kWavesPerBlock;
        int r = rc.r0 + wave_id();
        for (; r + step < rc.r1; r += 2 * step) {
            const U16Vec<VEC> v0 = U16Vec<VEC>::load(in + (size_t)r * a.in_pitch + col);
            const U16Vec<VEC> v1 = U16Vec<VEC>::load(in + (size_t)(r + step) * a.in_pitch + col);
            process_row(r, v0);
            process_row(r + step, v1);
        }
        if (r < rc.r1) process_row(r, U16Vec<VEC>::load(in + (size_t)r * a.in_pitch + col));
    }

    if (!OUT16 && ghist) {
        if (zeros) atomicAdd(&lds_hist[0], zeros);
        __syncthreads();
        const uint32_t n = lds_hist[threadIdx.x];
        if (n) atomicAdd(&ghist[threadIdx.x], (unsigned long long)n);
    }
}

// ------------------------------------------------------------------------------------
// 4a. CLAHE apply, exact, u16 levels out: the conflict-free form (config 3).
//     Kernel 4 spends two thirds of its LDS cycles in bank conflicts: a ds_read_b128 is served in four groups of 16 lanes, each
//     group one cycle when its 16 addresses fall into 16 different 16-byte slots of the 256-byte bank row, and a gather of random
//     bins puts three lanes on the busiest slot on average.  Here every bin holds SIXTEEN copies of its (c00, c01) pair side by
//     side -- one whole bank row -- and of its (c10, c11) pair in the row after it; a lane reads copy (lane & 15), and the lanes of
//     each of the instruction's four groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32) have sixteen different
//     values of lane & 15: every gather is conflict-free whatever the bins are.  128 KiB of tables: one persistent 1024-thread
//     workgroup per CU walks its share of the cell-major item list (equal rows per workgroup; vertical neighbours, strip after strip
//     of a cell: the tables are rebuilt, and the waves meet, only where the cell changes), items of up to 1024 rows (16 waves x 64
//     rows: lane l of a wave holds dy of the wave's l-th row), the
//     DN -> bin bytes of the band's window stay in LDS for the whole launch, rows are loaded and stored through buffer
//     descriptors (rows past the item's end and lanes outside it pass an out-of-range offset: no branch around a load or a
//     store, so the wait for the rows prefetched one step ahead does not wait for the stores issued after them).
//     Interior cells skip the clamp of the blend to [0, 1]: with weights in [0, 1] that sum to exactly 1 (dx is a multiple of
//     2^-53 below 1, so 1 - dx is exact) and CDFs in [0, 1] every rounded product is <= its weight and the rounded sums <= 1.
//     Algorithmic traffic: 2 B/px read + 2 B/px written.
// ------------------------------------------------------------------------------------
#ifdef SARPRO_RGB_WG_TIMES // instrumented build (tools/rgb_wg_times.py): when each persistent workgroup of the fused CLAHE -> RGB pass (or, KERNEL=clahe_apply_u16, of the exact u16 kernel) started and ended (100 MHz clock)
__device__ unsigned long long g_rgb_wg_times[1024][8]; // start, end, then thread 0's sums: wait at the item barrier, prologue, rows, items (u16 kernel: rows in extrapolating cells, rows of items with a straddling lane, rows, items, table builds)
#endif
constexpr int kCfBlock = 1024, kCfWaves = 16;
#ifndef SARPRO_CF_ROWS_AHEAD
#define SARPRO_CF_ROWS_AHEAD 2
#endif
struct CfLds {
    static constexpr uint32_t lut = 0;                  // DN -> bin bytes of the window [0, win_hi]: the clamped DN IS the byte's address
    static constexpr uint32_t lut_cap = 32768;
    static constexpr uint32_t cdf = lut_cap;            // [256 bins][2][16 copies] double2: 512 B per bin
    static constexpr uint32_t total = cdf + 256 * 512;  // 160 KiB
};
__global__ __launch_bounds__(kCfBlock) void k_clahe_apply_u16_cf(ClaheApplyArgs a, const int32_t *__restrict__ first, int nbands) {
    extern __shared__ __align__(16) unsigned char lds[];
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    constexpr int VEC = 8, NR = SARPRO_CF_ROWS_AHEAD;
    const int band = (int)blockIdx.x % nbands, wg = (int)blockIdx.x / nbands;
    const int lane = lane_id(), wave = wave_id();
    const uint16_t *__restrict__ in = a.in[band];
    uint16_t *__restrict__ out = reinterpret_cast<uint16_t *>(a.out[band]);
    const double *__restrict__ cdfs = a.cdfs[band];
    const uint8_t *__restrict__ glut = a.binlut[band];
    const uint32_t win_hi = a.dev_state[band].win_hi; // (chain mode: the window is [0, win_hi], known on the device only)
    const bool lut_lds = win_hi < CfLds::lut_cap;      // (else: the bins are gathered from the global table)
    if (lut_lds)
        for (uint32_t i = threadIdx.x; i <= win_hi; i += kCfBlock) lds[CfLds::lut + i] = glut[i];
    const uint32_t k16 = CfLds::cdf + (uint32_t)(lane & 15) * 16u;
    const double max_val = a.max_val;
    const uint32_t in_row_bytes = (uint32_t)a.in_pitch * 2u, out_row_bytes = (uint32_t)a.out_pitch * 2u;

#ifdef SARPRO_RGB_WG_TIMES
    if (threadIdx.x == 0) g_rgb_wg_times[blockIdx.x & 1023][0] = wall_clock64();
    unsigned long long cf_n[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    int have[4] = {-1, -1, -1, -1}; // the tiles whose CDFs the tables hold
    for (int item = first[wg]; item < first[wg + 1]; ++item) {
        const Rect rc = a.rects[item];
#ifdef SARPRO_RGB_WG_TIMES
        {
            const unsigned long long nr = (unsigned long long)(rc.r1 - rc.r0);
            if (rc.pad[0] & 1) cf_n[0] += nr;
            if ((rc.c0 - rc.cstart) % 8 != 0 || ((rc.c1 - rc.cstart) % 8 != 0 && rc.c1 - rc.cstart < 512)) cf_n[1] += nr;
            cf_n[2] += nr; ++cf_n[3];
            if (rc.id[0] != have[0] || rc.id[1] != have[1] || rc.id[2] != have[2] || rc.id[3] != have[3]) ++cf_n[4];
        }
#endif
        // a workgroup's items are vertical neighbours, strip after strip of a cell: the tables change three or four times per launch,
        // and only then do the waves meet -- otherwise each runs on into the next item on its own
        if (rc.id[0] != have[0] || rc.id[1] != have[1] || rc.id[2] != have[2] || rc.id[3] != have[3]) {
        have[0] = rc.id[0]; have[1] = rc.id[1]; have[2] = rc.id[2]; have[3] = rc.id[3];
        __syncthreads(); // the previous item's rows are done (first item: the byte table has landed once the next barrier is passed)
        {   // sixteen copies of both halves of every bin: thread = (bin mod 64, copy), four bins each; a wave's stores are contiguous
            const uint32_t copy = threadIdx.x & 15u, b0 = threadIdx.x >> 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t b = b0 + 64u * (uint32_t)i;
                v2d ct, cb;
                ct.x = cdfs[(size_t)rc.id[0] * 256 + b]; ct.y = cdfs[(size_t)rc.id[1] * 256 + b];
                cb.x = cdfs[(size_t)rc.id[2] * 256 + b]; cb.y = cdfs[(size_t)rc.id[3] * 256 + b];
                LDS_AT(v2d, CfLds::cdf + b * 512u + copy * 16u) = ct;
                LDS_AT(v2d, CfLds::cdf + b * 512u + 256u + copy * 16u) = cb;
            }
        }
        __syncthreads();
        }

        const int col = rc.cstart + lane * VEC;
        const bool lane_on = col < rc.c1 && col + VEC > rc.c0;
        const bool full = col >= rc.c0 && col + VEC <= rc.c1;
        double dx[VEC], omdx[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = col + j;
            const RowWeight w = a.col_w[(c >= rc.c0 && c < rc.c1) ? c : rc.c0];
            dx[j] = w.d;
            omdx[j] = w.omd;
        }
        // lane l: dy of this wave's l-th row of the item (rows r0 + wave + 16 l)
        const int nrows_w = (rc.r1 - rc.r0 - wave + kCfWaves - 1) / kCfWaves; // <= 64
        const double dyl = a.row_w[a.row_off + min(rc.r0 + wave + kCfWaves * lane, rc.r1 - 1)].d;
        const uint32_t dyl_lo = (uint32_t)__double_as_longlong(dyl), dyl_hi = (uint32_t)(__double_as_longlong(dyl) >> 32);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(in) + (size_t)rc.r0 * a.in_pitch + rc.cstart, 0,
                                                                               (int)((uint32_t)(rc.r1 - rc.r0) * in_row_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)rc.r0 * a.out_pitch + rc.cstart, 0,
                                                                                (int)((uint32_t)(rc.r1 - rc.r0) * out_row_bytes), 0x00020000);
        const uint32_t voff_ld = lane_on ? (uint32_t)lane * 16u : 0xFFFFFFFFu, voff_st = full ? (uint32_t)lane * 16u : 0xFFFFFFFFu;
        // The samples of a lane that straddles the item's edge (at most one lane at each edge) leave as ONE 2-byte store per row that every
        // lane issues: lanes 0-7 take the eight samples of the left straddling lane, lanes 8-15 those of the right one (v_readlane of its
        // packed row), every other lane -- and every sample outside the item -- passes an out-of-range offset.  (Written from the
        // straddling lane itself, inside a divergent branch, the stores were on one path only and the wait for the prefetched rows at the
        // loop top covered the turn's own stores in every item with such a lane: the fused pass's edge bytes)
        const int edge_l = (rc.c0 - rc.cstart) % VEC != 0 ? (rc.c0 - rc.cstart) / VEC : -1;                                       // (wave-uniform)
        const int edge_r = ((rc.c1 - rc.cstart) % VEC != 0 && rc.c1 - rc.cstart < 64 * VEC) ? (rc.c1 - rc.cstart) / VEC : -1;
        uint32_t voff_edge = 0xFFFFFFFFu;
        {
            const int src = lane < 8 ? edge_l : (lane < 16 && edge_r != edge_l) ? edge_r : -1, c = rc.cstart + src * VEC + (lane & 7);
            if (src >= 0 && c >= rc.c0 && c < rc.c1) voff_edge = (uint32_t)(src * VEC + (lane & 7)) * 2u;
        }

        auto item_rows = [&](auto edge_tag, auto lds_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value, LUTLDS = decltype(lds_tag)::value;
            auto load_rows = [&](v4u (&dst)[NR], int k) { // rows k .. k + NR - 1 of this wave (k counts the wave's rows)
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const bool on = k + i < nrows_w; // wave-uniform
                    dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, on ? voff_ld : 0xFFFFFFFFu, on ? (uint32_t)(wave + kCfWaves * (k + i)) * in_row_bytes : 0u, 0);
                }
            };
            v4u cur[NR], nxt[NR];
            load_rows(cur, 0);
            // NR dropped stores behind the first rows' loads: the loop is entered in its steady state (NR loads, then NR stores in
            // flight -- two per row with the edge samples' one --), so the wait for a turn's rows at the loop top is vmcnt(2 NR) on both edges -- without them the entry edge asks
            // for vmcnt(NR - 1), which in every later turn also waits for the write acknowledgement of the turn's first store
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                v4u z; z.x = z.y = z.z = z.w = 0u;
                __builtin_amdgcn_raw_buffer_store_b128(z, rs_out, 0xFFFFFFFFu - 16u * (uint32_t)i, 0u, 2);
                __builtin_amdgcn_raw_buffer_store_b16((uint16_t)0, rs_out, 0xFFFFFFFFu - 2u * (uint32_t)i, 0u, 0);
            }
            // (everything requested so far is waited for HERE: a first row that the allocator parks in scratch and reloads on the loop's doorstep
            // is otherwise the newest operation in flight at the entry edge, and the wait at the loop top becomes vmcnt(0) for every turn)
#pragma unroll
            for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(cur[i])); // (the rows are USED here: a reload, and the wait for it, come before this line)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            for (int k = 0; k < nrows_w; k += NR) {
                load_rows(nxt, k + NR);
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int kr = min(k + i, nrows_w - 1);
                    const bool on = k + i < nrows_w;
                    const double dy = __longlong_as_double(((long long)(uint32_t)__builtin_amdgcn_readlane((int)dyl_hi, kr) << 32) |
                                                           (long long)(uint32_t)__builtin_amdgcn_readlane((int)dyl_lo, kr));
                    const double omdy = 1.0 - dy;
                    const uint32_t w[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
                    uint32_t lv[VEC];
                    // Three phases with nothing moved across: the row's eight bin bytes, its sixteen gathers, the blends.  (Left to the
                    // scheduler every sample's byte read -> gather -> blend chain ran behind the previous sample's: two LDS round
                    // trips per sample, sixteen per row, and four waves per SIMD do not hide that.)
                    uint32_t bin[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const uint32_t d = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu);
                        const uint32_t dc = min(d, win_hi);
                        bin[j] = LUTLDS ? (uint32_t)LDS_AT(uint8_t, CfLds::lut + dc) : (uint32_t)glut[dc];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    v2d ct[VEC], cb[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const uint32_t ea = (bin[j] << 9) + k16;
                        ct[j] = LDS_AT(v2d, ea);
                        cb[j] = LDS_AT(v2d, 256u + ea);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const double top = ct[j].x * omdx[j] + ct[j].y * dx[j];
                        const double bottom = cb[j].x * omdx[j] + cb[j].y * dx[j];
                        double o = top * omdy + bottom * dy;
                        if (EDGE) o = fmin(fmax(o, 0.0), 1.0);
                        lv[j] = (uint32_t)(o * max_val); // truncation, o * max_val in [0, max_val]
                    }
                    // invalid samples (DN = 0) -> level 0: the packed pair times min(DN, 1) per half (two packed 16-bit operations per pair)
                    uint32_t pw[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t nz; // (written out: the compiler turns the vector form into compares and selects per half)
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(w[q]), "s"(0x00010001u));
                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(pw[q]) : "v"(lv[2 * q] | (lv[2 * q + 1] << 16)), "v"(nz));
                        lv[2 * q] = pw[q] & 0xFFFFu; lv[2 * q + 1] = pw[q] >> 16; // (only the straddling lanes below read these)
                    }
                    v4u pk;
                    pk.x = pw[0]; pk.y = pw[1]; pk.z = pw[2]; pk.w = pw[3];
                    const uint32_t soff = on ? (uint32_t)(wave + kCfWaves * (k + i)) * out_row_bytes : 0u;
                    // (gfx950: the data registers of a 128-bit buffer store with an SGPR offset must not be written in the next issue
                    // slots -- see the fused CLAHE -> RGB pass)
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, on ? voff_st : 0xFFFFFFFFu, soff, 2 /* nt */);
                    asm volatile("s_nop 1" : : "v"(pk) : "memory");
                    uint32_t e16 = 0u;
                    if (edge_l >= 0 || edge_r >= 0) { // (wave-uniform; nothing but register moves inside)
                        const int el = max(edge_l, 0), er = max(edge_r, 0);
                        uint32_t lid; // (the lane number made HERE: kept across the row loop, the shift below was spilled -- and a reload waits with vmcnt(0))
                        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lid));
                        const bool lid8 = lid < 8u;
                        const uint32_t j = lid & 7u;
                        uint32_t xs[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)pw[q], el), xr = (uint32_t)__builtin_amdgcn_readlane((int)pw[q], er);
                            xs[q] = lid8 ? xl : xr;
                        }
                        const uint32_t w2 = (j & 4u) ? ((j & 2u) ? xs[3] : xs[2]) : ((j & 2u) ? xs[1] : xs[0]);
                        e16 = (j & 1u) ? w2 >> 16 : w2 & 0xFFFFu;
                    }
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)e16, rs_out, on ? voff_edge : 0xFFFFFFFFu, soff, 0);
                }
#pragma unroll
                for (int i = 0; i < NR; ++i) cur[i] = nxt[i];
            }
        };
        if (lut_lds) {
            if (rc.pad[0] & 1) item_rows(std::true_type{}, std::true_type{});
            else item_rows(std::false_type{}, std::true_type{});
        } else {
            if (rc.pad[0] & 1) item_rows(std::true_type{}, std::false_type{});
            else item_rows(std::false_type{}, std::false_type{});
        }
    }
#ifdef SARPRO_RGB_WG_TIMES
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long *o = g_rgb_wg_times[blockIdx.x & 1023]; o[1] = wall_clock64(); for (int k = 0; k < 5; ++k) o[2 + k] = cf_n[k]; }
#endif
}

// ------------------------------------------------------------------------------------
// 4b. CLAHE apply, u8 output, speculative form (the headline kernel).
//     PMC profiling of kernel 4 showed it bound by VALU issue and LDS latency (38 VALU + 13 SALU
//     instructions and 32 B of conflicting LDS gathers per pixel, waves waiting 50 % of the time),
//     not by HBM.  This form does per pixel:
//       DN -> min(DN, win_hi) -> u16 table in LDS that holds the BYTE OFFSET of the pixel's CDF
//       entry (entry 256 is all-zero and is where DN = 0 points, so invalid pixels need no
//       branch) -> one 16-B gather of the four CDFs as f32 -> 6 f32 FMAs -> floor.
//     The f32 result decides the level unless o*255 lies within kSpecDelta of an integer; those
//     pixels (~0.1 % in interior cells) are recomputed with the reference's exact f64 sequence (CDFs and column
//     weights as f64 in LDS), so the raster is bit-identical to kernel 4's.
//     Error bound of the f32 value y32 against the reference's y = o*255 (u = 2^-24):
//       inputs are the f64 CDFs / weights rounded once to f32 (1+e, |e| <= u); with dx in [-0.5, 1) (the first
//       half tile extrapolates), top32 = fma(fl(c01 - c00), dx, c00) carries the input roundings
//       u (|c00| |1-dx| + |c01| |dx|) <= 2u, u |c01 - c00| |dx| each for the subtraction and for dx, and u |top| <= 2u
//       for the fma:  |dtop| <= 6u <= 8u,
//       the same for bottom, and |top|, |bottom| <= 2;  the row weights hold the 255 factor
//       (2 roundings), p = (top, bottom) * wy adds one more, the final add one more:
//       |y32 - y| <= 255 (|1-dy| + |dy|) (8u + 3u*2) + u |y|  <=  255*2*14u + 1020u  =  4.9e-4.
//     In interior cells (every weight in [0,1], weight pairs sum to 1, |top| <= 1) the same four terms are <= u each:
//       |dtop| <= 4u, |y32 - y| <= 255 (4u + 3u) + 255u = 255*8u = 1.2e-4.
//     The margin is therefore chosen per work item: kSpecDeltaEdge = 1/1024 = 9.8e-4 where the cell
//     extrapolates (first half tile row / column), kSpecDeltaInner = 1/4096 = 2.4e-4 elsewhere --
//     2x the respective worst case (round 1 ran with 4x: twice as many pixels on the exact path).  Outside the margin floor(y32) = floor(y), inside it the exact
//     path decides.  The reference's own f64 rounding (1e-13) is far below it.
//     (The test compares the bytes of ya = y32 - 0.5 - delta and yb = fl(ya + 2 delta): yb's own rounding, <= u |yb| = 1.5e-5
//     (3.6e-5 where the cell extrapolates), eats into the upper side, so what must hold is delta > err + u |yb|: 1.37e-4 /
//     5.3e-4 -- the margins are 1.8x that.  Measured over 3.2e9 samples (tools/spec_margin.py, an instrumented build that
//     evaluates the reference's y for every sample): |y32 - y| <= 4.4e-5 interior, 7.8e-5 extrapolating.)
//     All-zero and all-one CDF entries, whose exact results are known, get biased f32 entries that land
//     mid-interval (see the staging code), so they never reach the exact path.
//     Level 0 goes to a per-lane dummy histogram word (bin 0 = pixels - other bins): a shared word would
//     serialise the no-data wedge (measured: +7 % on the whole kernel).
// ------------------------------------------------------------------------------------
//     Round 4: the margins went from 1.8x to 1.17x what the bound requires (6.2e-4 / 1.6e-4 against 5.3e-4 / 1.37e-4; the largest
//     error ever measured is 0.13 / 0.28 of them; bias = fl(-0.5 - delta) and fl(2 delta) add < 1e-7): a third fewer samples on the
//     exact path, whose block sits in the row loop of every wave-row that holds one.
//     Cells that extrapolate along ONE axis only (every cell of the first half tile row or column but the corner) get their own
//     margins.  dy in [-0.5, 0), x interior: |top|, |bottom| <= 1 with |dtop| <= 4u as in interior cells; 255 (1.5 (4u + 2u) +
//     0.5 (4u + 2u)) for the two products, u |inner| <= 383u and u |y| <= 510u for the two fma roundings: 3953u = 2.36e-4, and yb's own
//     rounding 510u: delta > 2.66e-4.  dx in [-0.5, 0), y interior: |top| <= 2, |dtop| <= 5u: 255 (5u + 2 * 2u) + 511u + 510u = 3316u =
//     1.98e-4, + 510u: delta > 2.28e-4.  (Rect::pad[0] bit 1: dy < 0, bit 2: dx < 0.)
constexpr float kSpecDeltaEdgeY = 3.0e-4f, kSpecDeltaEdgeX = 2.6e-4f;
constexpr float kSpecDeltaEdge = 6.2e-4f, kSpecDeltaInner = 1.6e-4f;
__device__ __forceinline__ float spec_delta(int pad0) { // the f32 blend's margin for a work item of this kind of cell
    return !(pad0 & 1) ? kSpecDeltaInner : (pad0 & 6) == 2 ? kSpecDeltaEdgeY : (pad0 & 6) == 4 ? kSpecDeltaEdgeX : kSpecDeltaEdge;
}
#ifdef SARPRO_SPEC_MEASURE // instrumented build (tools/spec_margin.py): the largest |y32 - y| the speculative blend produced, per margin class
__device__ uint32_t g_spec_max_err[4]; // float bits, one per margin of spec_delta: [0] interior cells, [1] dy < 0 only, [2] dx < 0 only, [3] the corner (both)
__device__ __forceinline__ int spec_class(int pad0) { return !(pad0 & 1) ? 0 : (pad0 & 6) == 2 ? 1 : (pad0 & 6) == 4 ? 2 : 3; }
#endif
constexpr uint32_t kPartialHistLevels = 64; // partial level histogram: levels below this are counted one by one
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr uint32_t kSpecLutMaxEntries = 16384; // u16 offsets: 32 KiB of LDS at most

// The f32 CDF table is stored kCdfCopies times, interleaved so that copy c only occupies the 16-B slot
// classes {c, c+R, ...} of each 256-B LDS row; lane l reads copy l % R.  A ds_read_b128 services 16 lanes
// per cycle group and every such group holds exactly 16/R lanes of each residue, so lanes of different
// copies can never collide and the random per-pixel bins only conflict among 16/R lanes instead of 16.
// Measured on MI355X (400 MP scene): R = 2 costs 4 KiB of LDS per workgroup, which drops residency from
// 6 to 5 workgroups per CU, and the kernel gets SLOWER (0.78 -> 0.82 ms): it is more sensitive to resident
// waves than to gather conflicts.  Kept as a tunable; 1 = a single copy.
#ifndef SARPRO_CDF_COPIES
#define SARPRO_CDF_COPIES 1
#endif
constexpr uint32_t kCdfCopies = SARPRO_CDF_COPIES;
constexpr uint32_t kCdfPerRow = 16 / kCdfCopies;                         // entries per 256-B row
constexpr uint32_t kCdf32Bytes = ((257 + kCdfPerRow - 1) / kCdfPerRow) * 256;
__host__ __device__ constexpr uint32_t cdf32_offset(uint32_t entry) {    // byte offset of copy 0 of `entry`
    return (entry / kCdfPerRow) * 256u + (entry % kCdfPerRow) * kCdfCopies * 16u;
}
__host__ __device__ constexpr uint32_t cdf32_entry(uint32_t offset) {    // inverse of cdf32_offset
    return (offset / 256u) * kCdfPerRow + (offset % 256u) / (kCdfCopies * 16u);
}

struct SpecLds { // byte offsets into dynamic LDS
    static constexpr uint32_t cdf64 = 0;                      // [257][4] double
    static constexpr uint32_t cdf32 = 257 * 32 + 224;         // kCdfCopies interleaved copies of [257] float4 (256-B aligned)
    static constexpr uint32_t colw = cdf32 + kCdf32Bytes;     // [512] double: exact dx of the strip's columns
    static constexpr uint32_t hist = colw + 512 * 8;          // [256 + 64 + 4] u32: level bins | per-lane dummy words | [320] valid samples counted
    static constexpr uint32_t lut = hist + (256 + 64 + 4) * 4; // [lut_cap] u16
};
// The row weights stay in global memory: one wave-uniform load per row, prefetched with the row.
// (1 - dx) and (1 - dy) are recomputed as 1.0 - d: the reference's own expression (autoscale.rs:327-329)


// wave-uniform values moved to SGPRs (the row weights: they would otherwise hold VGPRs across the whole row)
__device__ __forceinline__ float to_sgpr(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
// (b3 << 24) | (b2 << 16) | (b1 << 8) | b0 of four values < 256 in three v_lshl_or_b32 (as plain shifts and ors the compiler
// re-associates them into five instructions per dword, masks of already zero-extended bytes included)
__device__ __forceinline__ uint32_t lshl_or(uint32_t hi, uint32_t sh, uint32_t lo) {
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(hi), "n"(sh), "v"(lo));
    return d;
}
#define pack4(b0, b1, b2, b3) lshl_or(lshl_or((b3), 8, (b2)), 16, lshl_or((b1), 8, (b0)))
__device__ __forceinline__ uint32_t to_sgpr_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ double to_sgpr(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// HIST: 0 = every level of every pixel counted, 1 = partial (levels < kPartialHistLevels one by one, the rest in bulk), 2 = the
// partial form on SAMPLED rows only (row % sample_stride == sample_phase): the histogram is then an estimate that the chain
// uses to PREDICT the synRGB floor, which the compose pass verifies exactly (chain_kernels.hip, k_chain_predict); together
// with the levels the sampled rows count their valid (DN != 0) pixels, the stratum the estimate is scaled by.
#define SPEC_LOAD(p) U16Vec<VEC>::load(p)
template <bool LUT_LDS, int HIST>
__device__ __forceinline__ void clahe_spec_rows(const ClaheApplyArgs &a, const Rect &rc, int band, unsigned char *lds,
                                                uint32_t win_hi) {
    constexpr int VEC = 8;
    const uint16_t *__restrict__ in = a.in[band];
    const uint8_t *__restrict__ glut = a.binlut[band];
    const RowWeight *__restrict__ row_w = a.row_w + a.row_off;
    const bool count_levels = a.level_hist[band] != nullptr;
    const int col = rc.cstart + lane_id() * VEC;
    const bool full = col >= rc.c0 && col + VEC <= rc.c1;
    // edge lanes: samples outside the item are computed like the others (whatever DN the row holds there), then their
    // level bytes are masked to 0 -- uncounted -- and never stored; masking the 8 levels costs 2 operations per row,
    // masking the 8 input samples cost 8
    uint32_t keep[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (col + j >= rc.c0 && col + j < rc.c1) keep[j >> 2] |= 0xFFu << (8 * (j & 3));
    uint8_t *dump = a.dump + ((((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave_id()) % (kSpecDumpBytes / 512)) * 512 +
                    lane_id() * 8; // one 512-B line per wave, shared round-robin

    float dxf[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) dxf[j] = (float)*reinterpret_cast<const double *>(lds + SpecLds::colw + (lane_id() * VEC + j) * 8);
    const uint32_t copy_off = ((uint32_t)lane_id() & (kCdfCopies - 1)) * 16u; // this lane's copy of the f32 CDF table
    const float near_delta = spec_delta(rc.pad[0]); // pad[0] bit 0: extrapolating cell (bit 1: in y, bit 2: in x)
    const float bias = -0.5f - near_delta, two_delta = 2.0f * near_delta;
    const uint32_t win_hi2 = win_hi | (win_hi << 16);
    constexpr bool PARTIAL_HIST = HIST != 0;
    uint32_t lane_max = 0u, lane_high = 0u; // PARTIAL_HIST: this lane's highest level and its count of levels >= kPartialHistLevels
    uint32_t lane_valid = 0u;               // HIST == 2: this lane's kept samples with DN != 0 on the sampled rows
#ifdef SARPRO_SPEC_MEASURE
    float spec_err = 0.0f;
#endif

    auto process_row = [&](int r, const U16Vec<VEC> &v, const double dy_v, const bool sampled) { // dy, sampled: wave-uniform
        // (counted first: the row's samples are then dead after the offset lookups, not held across the blend)
        if (HIST >= 2 && count_levels && sampled) { // valid samples of this row: non-zero halfwords, kept ones only
            const uint32_t w[4] = {v.v.x, v.v.y, v.v.z, v.v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t nz = (((w[k] & 0x7FFF7FFFu) + 0x7FFF7FFFu) | w[k]) & 0x80008000u; // bit 15 / 31: halfword != 0
                const uint32_t kb = keep[k >> 1] >> (16 * (k & 1));                               // the two keep bytes of these samples
                lane_valid += (uint32_t)__builtin_popcount(nz & (((kb & 0x80u) << 8) | ((kb & 0x8000u) << 16)));
            }
        }
        const double dy = to_sgpr(dy_v), omdy = to_sgpr(1.0 - dy);
        const float wy1 = to_sgpr((float)omdy * 255.0f), wy2 = to_sgpr((float)dy * 255.0f);
        uint32_t off[VEC];
        uint32_t pk[2] = {0u, 0u}, pb[2] = {0u, 0u}; // the 8 levels, packed as they will be stored
        if (LUT_LDS) {
            // win_hi < 16384 here: both samples of a dword are clamped at once (v_pk_min_u16), then one SDWA shift per sample
            // selects its half and doubles it into the byte offset of its u16 table entry
            const uint32_t w[4] = {v.v.x, v.v.y, v.v.z, v.v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                typedef unsigned short v2us __attribute__((ext_vector_type(2)));
                const v2us c = __builtin_elementwise_min(__builtin_bit_cast(v2us, w[k]), __builtin_bit_cast(v2us, win_hi2));
                const uint32_t cw = __builtin_bit_cast(uint32_t, c);
                uint32_t a0, a1;
                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a0) : "v"(1u), "v"(cw));
                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a1) : "v"(1u), "v"(cw));
                off[2 * k] = (uint32_t)LDS_AT(uint16_t, SpecLds::lut + a0);
                off[2 * k + 1] = (uint32_t)LDS_AT(uint16_t, SpecLds::lut + a1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const uint32_t i = min(v.get(j), win_hi);
                off[j] = SpecLds::cdf32 + cdf32_offset(i ? (uint32_t)glut[i] : 256u);
            }
        }
        constexpr int kAhead = 2; // CDF gathers issued ahead of their use (3 in flight; measured: 1 -> 2 -1.4 %, 4 no better)
        v4f cq[VEC];
#pragma unroll
        for (int j = 0; j < kAhead; ++j) cq[j] = LDS_AT(v4f, off[j] + copy_off);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (j + kAhead < VEC) cq[j + kAhead] = LDS_AT(v4f, off[j + kAhead] + copy_off);
            const v4f c4 = cq[j]; // (c00, c10, c01 - c00, c11 - c10): the differences are taken once, when the table is staged
            const float top = fmaf(c4.z, dxf[j], c4.x); // c00 + dx (c01 - c00): no (1 - dx) to keep in registers
            const float bottom = fmaf(c4.w, dxf[j], c4.y);
            // ya = y - 0.5 - delta, yb = y - 0.5 + delta: v_cvt_pk_u8_f32 rounds to nearest-even and saturates, so both
            // give the same byte n only if y lies in (n + (delta - err), n + 1 - (delta - err)): then n = floor(y)
            // clamped to 0..255.  Where the bytes differ the pixel goes to the exact path.
            const float ya = fmaf(bottom, wy2, fmaf(top, wy1, bias));
            const float yb = ya + two_delta;
            pk[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(ya, j & 3, pk[j >> 2]);
            pb[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(yb, j & 3, pb[j >> 2]);
#ifdef SARPRO_SPEC_MEASURE
            {   // the reference's y for EVERY sample (unclamped), against the f32 value; biased entries (all-zero / all-one bins) excluded
                const double4 e4 = *reinterpret_cast<const double4 *>(lds + SpecLds::cdf64 + cdf32_entry(off[j] - SpecLds::cdf32) * 32u);
                const bool biased = (e4.x == 0.0 && e4.y == 0.0 && e4.z == 0.0 && e4.w == 0.0) ||
                                    (e4.x == 1.0 && e4.y == 1.0 && e4.z == 1.0 && e4.w == 1.0 && !(rc.pad[0] & 1));
                const double dxe = *reinterpret_cast<const double *>(lds + SpecLds::colw + (lane_id() * VEC + j) * 8);
                const double te = e4.x * (1.0 - dxe) + e4.y * dxe, be = e4.z * (1.0 - dxe) + e4.w * dxe;
                const double ye = (te * omdy + be * dy) * 255.0;
                const float err = (float)fabs(((double)ya - (double)bias) - ye);
                if (!biased && ((keep[j >> 2] >> (8 * (j & 3))) & 0xFFu)) spec_err = fmaxf(spec_err, err);
            }
#endif
        }
        const uint32_t d0 = pk[0] ^ pb[0], d1 = pk[1] ^ pb[1];
        if (d0 | d1) { // rare: some pixel of this lane lies within the margin of an integer -> exact path
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (((j < 4 ? d0 : d1) >> (8 * (j & 3))) & 0xFFu) { // reference op order (autoscale.rs:327-329, 602)
                    const double4 c4 = *reinterpret_cast<const double4 *>(lds + SpecLds::cdf64 + cdf32_entry(off[j] - SpecLds::cdf32) * 32u);
                    const double dx = *reinterpret_cast<const double *>(lds + SpecLds::colw + (lane_id() * VEC + j) * 8);
                    const double top = c4.x * (1.0 - dx) + c4.y * dx;
                    const double bottom = c4.z * (1.0 - dx) + c4.w * dx;
                    double o = top * omdy + bottom * dy;
                    o = fmin(fmax(o, 0.0), 1.0);
                    const uint32_t level = (uint32_t)(o * 255.0);
                    const uint32_t sh = 8 * (j & 3);
                    pk[j >> 2] = (pk[j >> 2] & ~(0xFFu << sh)) | (level << sh);
                }
            }
        }
        pk[0] &= keep[0];
        pk[1] &= keep[1];
        if (count_levels && (HIST < 2 || sampled)) { // level 0 (incl. masked edge samples) is not counted: bin 0 = pixels - others
            // one predicated ds_add_u32 per pixel, EXEC narrowed to the lanes that count (a shared bin 0 would serialise
            // the no-data wedge); written out because the compiler wraps each predicated atomic in a branch
            const uint32_t one = 1u, two = 2u, zero = 0u, lim = kPartialHistLevels;
            unsigned long long sv, sw;
            uint32_t haddr;
            if (!PARTIAL_HIST) {
#define SPEC_HIST_BYTE(P, K)                                                                                                   \
    asm volatile("v_cmp_ne_u32_sdwa vcc, %2, %5 src0_sel:BYTE_" #K " src1_sel:DWORD\n\t"                                    \
                 "v_lshlrev_b32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K "\n\t" \
                 "s_and_saveexec_b64 %1, vcc\n\t"                                                                             \
                 "ds_add_u32 %0, %3 offset:%6\n\t"                                                                            \
                 "s_mov_b64 exec, %1"                                                                                         \
                 : "=&v"(haddr), "=&s"(sv) : "v"(P), "v"(one), "v"(two), "v"(zero), "n"(SpecLds::hist) : "vcc", "memory")
                SPEC_HIST_BYTE(pk[0], 0); SPEC_HIST_BYTE(pk[0], 1); SPEC_HIST_BYTE(pk[0], 2); SPEC_HIST_BYTE(pk[0], 3);
                SPEC_HIST_BYTE(pk[1], 0); SPEC_HIST_BYTE(pk[1], 1); SPEC_HIST_BYTE(pk[1], 2); SPEC_HIST_BYTE(pk[1], 3);
#undef SPEC_HIST_BYTE
            } else {
                // Only levels 1 .. kPartialHistLevels-1 are counted one by one (a quarter of the lanes on equalised data:
                // the LDS atomics, not the arithmetic, are what the histogram costs).  Of the levels above the lane keeps
                // their number and its highest level; they are added to that level's bin after the rows.  The consumers
                // (rescale range, synRGB floor) only need the exact low bins, the total and the highest level present --
                // see chain_kernels.hip k_level_hist_guard for the proof and the fallback when the low bins are empty.
#define SPEC_HIST_BYTE(P, K)                                                                                                   \
    asm volatile("v_cmp_ne_u32_sdwa vcc, %3, %6 src0_sel:BYTE_" #K " src1_sel:DWORD\n\t"                                    \
                 "v_cmp_lt_u32_sdwa %2, %3, %7 src0_sel:BYTE_" #K " src1_sel:DWORD\n\t"                                     \
                 "v_lshlrev_b32_sdwa %0, %5, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #K "\n\t" \
                 "s_and_b64 vcc, vcc, %2\n\t"                                                                                 \
                 "s_and_saveexec_b64 %1, vcc\n\t"                                                                             \
                 "ds_add_u32 %0, %4 offset:%8\n\t"                                                                            \
                 "s_mov_b64 exec, %1"                                                                                         \
                 : "=&v"(haddr), "=&s"(sv), "=&s"(sw) : "v"(P), "v"(one), "v"(two), "v"(zero), "v"(lim), "n"(SpecLds::hist) : "vcc", "memory")
                SPEC_HIST_BYTE(pk[0], 0); SPEC_HIST_BYTE(pk[0], 1); SPEC_HIST_BYTE(pk[0], 2); SPEC_HIST_BYTE(pk[0], 3);
                SPEC_HIST_BYTE(pk[1], 0); SPEC_HIST_BYTE(pk[1], 1); SPEC_HIST_BYTE(pk[1], 2); SPEC_HIST_BYTE(pk[1], 3);
#undef SPEC_HIST_BYTE
                static_assert(kPartialHistLevels == 64, "the high-level count below tests the top two bits of a level");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    lane_high += (uint32_t)__builtin_popcount((pk[h] | (pk[h] << 1)) & 0x80808080u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) lane_max = max(lane_max, (pk[h] >> (8 * k)) & 0xFFu);
                }
            }
        }
        if (HIST == 3) return; // the sample-only pass writes no levels
        uint8_t *o8 = reinterpret_cast<uint8_t *>(a.out[band]) + (size_t)r * a.out_pitch + col;
        // Every lane issues the 8-byte store, edge lanes into the scratch line: a store inside a divergent branch sits
        // behind an `execz` skip, the waitcnt pass then sees a path through the row without a store and makes the next
        // row wait for vmcnt(0) -- i.e. for this row's store to be acknowledged -- before it may use its prefetched data.
        *reinterpret_cast<uint2 *>(full ? o8 : dump) = make_uint2(pk[0], pk[1]);
        if (!full) {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if ((keep[j >> 2] >> (8 * (j & 3))) & 1u) o8[j] = (uint8_t)(pk[j >> 2] >> (8 * (j & 3)));
        }
    };

    if (HIST == 3) { // sample-only pass (the fused CLAHE -> RGB chain's floor prediction): the item's sampled rows, nothing else
        if (col < rc.c1 && col + VEC > rc.c0) {
            const int stride = (int)a.sample_stride;
            const int first = rc.r0 + ((int)a.sample_phase - (a.row_off + rc.r0) % stride + stride) % stride;
            const int step = kWavesPerBlock * stride;
            const uint16_t *p = in + col;
            int r = __builtin_amdgcn_readfirstlane(first + wave_id() * stride);
            if (r < rc.r1) { // (three rows in flight instead of one measured SLOWER here: 0.069 against 0.057 ms)
                U16Vec<VEC> cur = SPEC_LOAD(p + (size_t)r * a.in_pitch);
                double dy = row_w[r].d;
                for (; r < rc.r1; r += step) {
                    const int rn = min(r + step, rc.r1 - 1); // (a redundant load at the end, so that the prefetch is unconditional)
                    const U16Vec<VEC> nxt = SPEC_LOAD(p + (size_t)rn * a.in_pitch);
                    const double dyn = row_w[rn].d;
                    process_row(r, cur, dy, true);
                    cur = nxt;
                    dy = dyn;
                }
            }
        }
    } else if (col < rc.c1 && col + VEC > rc.c0) {
        const int step = kWavesPerBlock;
        const uint16_t *p = in + col;
        int r = __builtin_amdgcn_readfirstlane(rc.r0 + wave_id());
        // HIST == 2: row r is sampled iff (row_off + r) % stride == phase; kept as a scalar residue that advances with r
        const int stride = (int)a.sample_stride, phase = (int)a.sample_phase;
        int smod = HIST == 2 ? __builtin_amdgcn_readfirstlane((a.row_off + r) % stride) : 0;
        // the next row is always loaded (clamped to the item's last row: at worst one redundant load), so the
        // load is unconditional and the compiler can wait with vmcnt(1) -- a conditional prefetch made it wait
        // vmcnt(0) right after issuing it, i.e. no overlap at all inside a wave
        if (r < rc.r1) {
            U16Vec<VEC> cur = SPEC_LOAD(p + (size_t)r * a.in_pitch);
            double dy = row_w[r].d; // exact dy of the row, wave-uniform, prefetched like the row itself
            { // first row peeled: the loop is then entered with the same operations in flight as on its back edge
                const int rn = min(r + step, rc.r1 - 1);
                const U16Vec<VEC> nxt = SPEC_LOAD(p + (size_t)rn * a.in_pitch);
                const double dyn = row_w[rn].d;
                process_row(r, cur, dy, smod == phase);
                cur = nxt;
                dy = dyn;
                r += step;
                if (HIST == 2) { smod += step; if (smod >= stride) smod -= stride; }
            }
            for (; r < rc.r1; r += step) {
                const int rn = min(r + step, rc.r1 - 1);
                const U16Vec<VEC> nxt = SPEC_LOAD(p + (size_t)rn * a.in_pitch);
                const double dyn = row_w[rn].d;
                process_row(r, cur, dy, smod == phase);
                cur = nxt;
                dy = dyn;
                if (HIST == 2) { smod += step; if (smod >= stride) smod -= stride; }
            }
        }
    }
    if (PARTIAL_HIST && count_levels && lane_high) // lane_high > 0 implies lane_max >= kPartialHistLevels
        atomicAdd(reinterpret_cast<uint32_t *>(lds + SpecLds::hist) + lane_max, lane_high);
    if (HIST >= 2 && count_levels) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) lane_valid += __shfl_xor(lane_valid, m, 64);
        if (lane_id() == 0 && lane_valid) atomicAdd(reinterpret_cast<uint32_t *>(lds + SpecLds::hist) + 320, lane_valid);
    }
#ifdef SARPRO_SPEC_MEASURE
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) spec_err = fmaxf(spec_err, __shfl_xor(spec_err, m, 64));
    if (lane_id() == 0 && spec_err > 0.0f) atomicMax(&g_spec_max_err[spec_class(rc.pad[0])], __float_as_uint(spec_err));
#endif
}

// one kernel per histogram mode (known on the host): each gets the registers ITS row loop needs, not the maximum of all three
template <int HIST>
__global__ __launch_bounds__(kBlock) void k_clahe_apply_u8_spec(ClaheApplyArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    if (a.gate && a.gate->verdict == 0) return; // fallback launch of the fused chain: the fused pass's RGB stands
    const Rect rc = a.rects[blockIdx.x];
    const int band = blockIdx.y;
    const double *__restrict__ cdfs = a.cdfs[band];
    const uint8_t *__restrict__ glut = a.binlut[band];
    // the table is constant from win_hi on; beyond the LDS capacity it is gathered from global memory
    const uint32_t win_hi = a.dev_state ? a.dev_state[band].win_hi : a.win_hi[band];
    const bool lut_lds = win_hi < a.lut_cap;
    unsigned long long *ghist = a.level_hist[band];
    {
        const int b = threadIdx.x;
        double c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = cdfs[(size_t)rc.id[k] * 256 + b];
        *reinterpret_cast<double4 *>(lds + SpecLds::cdf64 + b * 32) = make_double4(c[0], c[1], c[2], c[3]);
        // f32 copy laid out (c00, c10 | c01 - c00, c11 - c10): (top, bottom) = (x,y) + dx*(z,w), one FMA each.
        // Saturated bins (all four CDFs exactly 1.0: at least the top bin, i.e. every pixel above p99) blend
        // to y = 255 +- rounding, which the margin test would send to the exact path for ~1 % of all pixels.
        // In an interior cell their exact result is known: with dx in [0,1), fl(fl(1-dx) + dx) = 1.0 exactly
        // (the error of fl(1-dx) is at most half an ulp of the result and ties round to even), likewise
        // for dy, so o = 1.0 and the level is 255.  Their f32 entry is therefore biased to 1.001: y32 = 255.25
        // floors (and clamps) to 255 and is never "near".  Extrapolating cells keep the exact path.
        const bool saturated = c[0] == 1.0 && c[1] == 1.0 && c[2] == 1.0 && c[3] == 1.0 && !(rc.pad[0] & 1);
        // Bins whose four CDFs are all exactly 0 (and entry 256, where invalid pixels point) have the exact
        // result 0 whatever the weights; their f32 entry is k = 0.5/255 so that y32 = 0.5 (weights sum to 1):
        // level 0, never "near" -- no separate zero test per pixel.
        const bool zero = c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0 && c[3] == 0.0;
        const float kz = 0.5f / 255.0f;
        const float c00 = (float)c[0], c01 = (float)c[1], c10 = (float)c[2], c11 = (float)c[3];
        const float4 e32 = saturated ? make_float4(1.001f, 1.001f, 0.0f, 0.0f)
                           : zero    ? make_float4(kz, kz, 0.0f, 0.0f)
                                     : make_float4(c00, c10, c01 - c00, c11 - c10);
#pragma unroll
        for (uint32_t cc = 0; cc < kCdfCopies; ++cc)
            *reinterpret_cast<float4 *>(lds + SpecLds::cdf32 + cdf32_offset(b) + cc * 16) = e32;
        if (b == 0) {
            *reinterpret_cast<double4 *>(lds + SpecLds::cdf64 + 256 * 32) = make_double4(0.0, 0.0, 0.0, 0.0);
#pragma unroll
            for (uint32_t cc = 0; cc < kCdfCopies; ++cc)
                *reinterpret_cast<float4 *>(lds + SpecLds::cdf32 + cdf32_offset(256) + cc * 16) = make_float4(kz, kz, 0.0f, 0.0f);
        }
        for (int i = b; i < 256 + 64 + 4; i += kBlock) reinterpret_cast<uint32_t *>(lds + SpecLds::hist)[i] = 0;
        for (int i = b; i < 512; i += kBlock) { // exact column weights of this strip, for the f64 path
            const int c2 = rc.cstart + i;
            *reinterpret_cast<double *>(lds + SpecLds::colw + i * 8) = a.col_w[(c2 >= rc.c0 && c2 < rc.c1) ? c2 : rc.c0].d;
        }
        // DN -> offset table: 16 DNs per thread and load (one round trip to L2 instead of one per 256 entries: the per-item staging
        // of this kernel was ~25 us, visible wherever items are short -- the sample-only pass, small scenes)
        if (lut_lds)
            for (uint32_t g = b; g * 16u <= win_hi; g += kBlock) {
                const uint4 q = *reinterpret_cast<const uint4 *>(glut + g * 16u);
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                uint32_t e[8];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t dn = g * 16u + k, bin = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    const uint32_t o16 = SpecLds::cdf32 + cdf32_offset(dn ? bin : 256u);
                    e[k >> 1] = (k & 1) ? (e[k >> 1] | (o16 << 16)) : o16;
                }
                uint4 *dst = reinterpret_cast<uint4 *>(lds + SpecLds::lut + g * 32u);
                dst[0] = make_uint4(e[0], e[1], e[2], e[3]);
                dst[1] = make_uint4(e[4], e[5], e[6], e[7]);
            }
    }
    __syncthreads();
    if (lut_lds) clahe_spec_rows<true, HIST>(a, rc, band, lds, win_hi);
    else clahe_spec_rows<false, HIST>(a, rc, band, lds, win_hi);
    if (ghist) {
        // The rows' ds_add_u32 are inline assembly: the compiler's wait-count pass does not see them and emitted a bare s_barrier here, so a
        // wave could pass the barrier with its last adds still in flight and the flush below missed them (round 5: 162 of 36 M band-pixels at
        // 36 MP -- enough to leave bin 0 = pixels - others above zero on a raster WITHOUT level 0, where it decides the u8 rescale).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t n = reinterpret_cast<const uint32_t *>(lds + SpecLds::hist)[threadIdx.x];
        // HIST == 2: the item's sampled rows stand for all its rows: counts are added with the weight rows / sampled rows (fixed
        // point, kSampleWeightOne = 1.0), so that row blocks whose height is not a multiple of the stride are neither over- nor
        // under-represented (post-stratification by work item; an unweighted sample was off by up to 0.2 % on the headline scene,
        // whose texture classes change every 1250 rows)
        unsigned long long w = 1ull;
        if (HIST >= 2) {
            const int stride = (int)a.sample_stride, m = (a.row_off + rc.r0) % stride;
            const int first = rc.r0 + ((int)a.sample_phase - m + stride) % stride;
            const int cnt = first < rc.r1 ? (rc.r1 - 1 - first) / stride + 1 : 0;
            w = cnt ? ((unsigned long long)(rc.r1 - rc.r0) * kSampleWeightOne + (unsigned)cnt / 2) / (unsigned)cnt : 0ull;
        }
        const uint32_t replica = HIST >= 2 ? (blockIdx.x % (uint32_t)kSampleReplicas) : 0u; // (kernels.h: kSampleReplicas)
        if (n && threadIdx.x) atomicAdd(&ghist[(size_t)replica * 256 * kMaxBands + threadIdx.x], (unsigned long long)n * w); // bin 0 restored by the consumer
        if (HIST >= 2 && threadIdx.x == 0) {
            const uint32_t nv = reinterpret_cast<const uint32_t *>(lds + SpecLds::hist)[320];
            if (nv) atomicAdd(&a.sample_valid[replica * kMaxBands + band], (unsigned long long)nv * w);
        }
    }
}

// ------------------------------------------------------------------------------------
// 5. Table apply for the percentile strategies: out = LUT[DN] (the table already folds
//    dB, clip, gamma, quantisation and -- for u8 -- the global rescale).  Flat over
//    (row, vector) pairs.  Algorithmic traffic: 2 B/px read + 1 or 2 B/px written.
// ------------------------------------------------------------------------------------
template <int VEC, bool OUT16>
__global__ __launch_bounds__(kBlock) void k_lut_apply_u16(LutApplyArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using Entry = typename std::conditional<OUT16, uint16_t, uint8_t>::type;
    Entry *lds_lut = reinterpret_cast<Entry *>(lds_raw);
    const Entry *__restrict__ glut = reinterpret_cast<const Entry *>(a.lut);
    const uint32_t win_lo = a.dev_state ? 0u : a.win_lo, win_hi = a.dev_state ? a.dev_state[a.band].win_hi : a.win_hi;
    const bool lut_lds = a.dev_state ? win_hi < a.lut_cap : a.lut_in_lds != 0;
    if (lut_lds) {
        for (uint32_t i = threadIdx.x; i <= win_hi - win_lo; i += kBlock) lds_lut[i] = glut[win_lo + i];
        __syncthreads();
    }
    const uint32_t vpr = (a.cols + VEC - 1) / VEC;
    const uint64_t total = (uint64_t)a.rows * vpr;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(idx / vpr);
        const uint32_t col = (uint32_t)(idx - (uint64_t)r * vpr) * VEC;
        const U16Vec<VEC> v = U16Vec<VEC>::load(a.in + (size_t)r * a.in_pitch + col);
        uint32_t lv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const uint32_t d = v.get(j);
            const uint32_t dc = min(max(d, win_lo), win_hi);
            const uint32_t l = lut_lds ? (uint32_t)lds_lut[dc - win_lo] : (uint32_t)glut[dc];
            lv[j] = d ? l : 0u;
        }
        const bool full = col + VEC <= a.cols;
        if (OUT16) {
            uint16_t *o = reinterpret_cast<uint16_t *>(a.out) + (size_t)r * a.out_pitch + col;
            if (VEC == 8 && full) {
                uint4 pk;
                pk.x = lv[0] | (lv[1] << 16); pk.y = lv[2] | (lv[3] << 16);
                pk.z = lv[4] | (lv[5] << 16); pk.w = lv[6] | (lv[7] << 16);
                *reinterpret_cast<uint4 *>(o) = pk;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) if (col + j < a.cols) o[j] = (uint16_t)lv[j];
            }
        } else {
            uint8_t *o = reinterpret_cast<uint8_t *>(a.out) + (size_t)r * a.out_pitch + col;
            if (VEC == 8 && full) {
                uint2 pk;
                pk.x = lv[0] | (lv[1] << 8) | (lv[2] << 16) | (lv[3] << 24);
                pk.y = lv[4] | (lv[5] << 8) | (lv[6] << 16) | (lv[7] << 24);
                *reinterpret_cast<uint2 *>(o) = pk;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) if (col + j < a.cols) o[j] = (uint8_t)lv[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// 6. Synthetic-RGB compose (synthetic_rgb.rs:55-64 / 158-175) from two u8 level rasters.
//    Tables (with each band's u8 rescale and the water short-circuit folded in) live in LDS:
//    R2[256] | G2[256] | B2[65536].  16 px per lane: 2 x 16 B loads, 3 x 16 B stores.
//    Algorithmic traffic: 2 B/px read + 3 B/px written.
// ------------------------------------------------------------------------------------
constexpr int kComposeBlock = 1024; // 66 KiB of tables per block -> 2 blocks (32 waves) per CU
constexpr int kComposeTableBytes = 512 + 65536;
constexpr int kComposeStageBytes = (kComposeBlock / kWave) * 3072; // per-wave 3 KiB transpose stage (VEC = 16)

// SPEC (the CLAHE chain's speculative composition, chain_kernels.hip k_chain_predict): the tables were built for a PREDICTED
// floor F; while composing, the pass counts the band-pixels (both level rasters) with level >= F and >= F + 1 on the packed
// level bytes it has loaded anyway, and the workgroup that finishes last turns the counts into the verdict:
// cum(F-1) < target <= cum(F)  <=>  F is the floor of synthetic_rgb.rs:99-113.  Counting without a compare per byte:
// with A_T = sum |x - T| over the bytes (one v_sad_u8 per dword and threshold, accumulating),
//   #{x >= T + 1} = sum min(x, T + 1) - sum min(x, T)   and   sum min(x, T) = (sum x + N T - A_T) / 2
//   =>  #{x >= T + 1} = (N + A_T - A_(T+1)) / 2,
// so three accumulators (T = F - 1, F, F + 1) give both counts for 3 instead of 8 VALU operations per dword (a SWAR
// compare-and-popcount form cost the pass 0.025 ms of its 0.38).  Gating: SPEC runs iff spec_ok; the plain form with a.spec
// set runs iff the verdict refuted the speculative RGB (or it never ran).

template <int VEC, bool SPEC>
__global__ __launch_bounds__(kComposeBlock) void k_compose_u8(ComposeArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if (SPEC) { if (a.spec->spec_ok != kSpecIdentity) return; } // (a predicted rescale is verified by the fused pass only)
    else if (a.spec && a.spec->verdict == 0) return; // fallback composition: the speculative RGB stands
    // SPEC: thresholds F - 1 (F when F = 0: unused), F, F + 1 in every byte; this lane's |x - T| sums over its vectors, its
    // pixel count, and the direct counts of the ragged row tails
    uint32_t t4[3] = {0u, 0u, 0u}, sad[3] = {0u, 0u, 0u}, n_ge[2] = {0u, 0u}, n_px = 0u, n_vec_px = 0u;
    uint32_t fpred = 0u;
    if (SPEC) {
        fpred = (uint32_t)a.spec->floor_pred;
        t4[0] = (fpred ? fpred - 1u : 0u) * 0x01010101u; t4[1] = fpred * 0x01010101u; t4[2] = (fpred + 1u) * 0x01010101u;
    }
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.tables);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < kComposeTableBytes / 16; i += kComposeBlock) dst[i] = src[i];
    }
    __syncthreads();
    const uint8_t *R2 = lds_raw, *G2 = lds_raw + 256, *B2 = lds_raw + 512;
    // per-wave stage: a lane's 48 output bytes are interleaved RGB of ITS 16 pixels; written straight
    // to memory each of the three 16-B stores of a wave would touch 64 separate 48-B-strided pieces.
    // Staging them in LDS and reading back lane-linear makes every store instruction 1 KiB contiguous.
    uint4 *stage = reinterpret_cast<uint4 *>(lds_raw + kComposeTableBytes + (threadIdx.x >> 6) * 3072);
    const int lane = threadIdx.x & 63;
    const uint32_t vpr = (a.cols + VEC - 1) / VEC;
    const uint64_t total = (uint64_t)a.rows * vpr;
    const uint64_t nwaves = (uint64_t)gridDim.x * (kComposeBlock / kWave);
    const uint64_t wave0 = (uint64_t)blockIdx.x * (kComposeBlock / kWave) + (threadIdx.x >> 6);
    if (VEC == 16) {
        // a wave owns 64 consecutive vectors of one row (1024 px); rows are walked wave-wide so the
        // transposed stores stay inside one row
        const uint32_t wpr = (vpr + 63) / 64; // wave-chunks per row
        const uint64_t chunks = (uint64_t)a.rows * wpr;
        // software pipeline: the wave's NEXT chunk is loaded before this one is looked up and stored (with 16 waves per CU the
        // loop was exposed to the load latency); the loads are unconditional -- clamped to the last chunk and, past the row's
        // last full vector, to the end of the pitch -- so that the compiler can count them
        const uint32_t last_vec_off = a.in_pitch >= 16 ? (uint32_t)a.in_pitch - 16u : 0u;
        auto chunk_off = [&](uint64_t ch) -> size_t {
            const uint32_t r = (uint32_t)(ch / wpr);
            const uint32_t col = ((uint32_t)(ch - (uint64_t)r * wpr) * 64 + lane) * 16;
            return (size_t)r * a.in_pitch + min(col, last_vec_off);
        };
        uint4 n1 = make_uint4(0, 0, 0, 0), n2 = n1;
        if (wave0 < chunks) {
            const size_t o0 = chunk_off(wave0);
            n1 = *reinterpret_cast<const uint4 *>(a.b1 + o0);
            n2 = *reinterpret_cast<const uint4 *>(a.b2 + o0);
        }
        for (uint64_t ch = wave0; ch < chunks; ch += nwaves) {
            const uint32_t r = (uint32_t)(ch / wpr);
            const uint32_t v0 = (uint32_t)(ch - (uint64_t)r * wpr) * 64;
            const uint32_t v = v0 + lane;
            const uint32_t col = v * 16;
            const bool fullv = col + 16 <= a.cols;
            const uint32_t nfull = (a.cols / 16 > v0) ? min(64u, a.cols / 16 - v0) : 0u; // full vectors in this chunk (prefix)
            uint32_t o[12];
            const uint4 q1 = n1, q2 = n2;
            {
                const size_t on = chunk_off(min(ch + nwaves, chunks - 1));
                n1 = *reinterpret_cast<const uint4 *>(a.b1 + on);
                n2 = *reinterpret_cast<const uint4 *>(a.b2 + on);
            }
            if (fullv) {
                const uint32_t w1[4] = {q1.x, q1.y, q1.z, q1.w}, w2[4] = {q2.x, q2.y, q2.z, q2.w};
                if (SPEC) {
                    n_vec_px += 32u;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int k = 0; k < 3; ++k) sad[k] = __builtin_amdgcn_sad_u8(w2[g], t4[k], __builtin_amdgcn_sad_u8(w1[g], t4[k], sad[k]));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) { // 4 px -> 12 bytes -> 3 dwords
                    uint32_t px[4][3];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t v1 = (w1[g] >> (8 * j)) & 0xFF, v2 = (w2[g] >> (8 * j)) & 0xFF;
                        px[j][0] = R2[v1]; px[j][1] = G2[v2]; px[j][2] = B2[(v1 << 8) | v2];
                    }
                    o[3 * g + 0] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
                    o[3 * g + 1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
                    o[3 * g + 2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
                }
                stage[lane * 3 + 0] = make_uint4(o[0], o[1], o[2], o[3]);
                stage[lane * 3 + 1] = make_uint4(o[4], o[5], o[6], o[7]);
                stage[lane * 3 + 2] = make_uint4(o[8], o[9], o[10], o[11]);
            }
            // (same wave wrote and reads the stage: program order inside a wave, no barrier needed)
            uint8_t *rowp = a.rgb + ((size_t)r * a.rgb_pitch_px + (size_t)v0 * 16) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t slot = k * 64 + lane; // 16-B slot of the chunk's 3 * nfull slots
                if (slot < nfull * 3) store_stream16(reinterpret_cast<uint4 *>(rowp) + slot, stage[slot]);
            }
            if (!fullv && col < a.cols) { // ragged tail of the row: scalar
                const uint8_t *p1 = a.b1 + (size_t)r * a.in_pitch + col, *p2 = a.b2 + (size_t)r * a.in_pitch + col;
                uint8_t *po = a.rgb + ((size_t)r * a.rgb_pitch_px + col) * 3;
                for (uint32_t j = 0; col + j < a.cols; ++j) {
                    const uint32_t v1 = p1[j], v2 = p2[j];
                    po[3 * j + 0] = R2[v1]; po[3 * j + 1] = G2[v2]; po[3 * j + 2] = B2[(v1 << 8) | v2];
                    if (SPEC) {
                        n_px += 2u;
                        n_ge[0] += (v1 >= fpred) + (v2 >= fpred);
                        n_ge[1] += (v1 >= fpred + 1u) + (v2 >= fpred + 1u);
                    }
                }
            }
        }
    } else {
        for (uint64_t idx = (uint64_t)blockIdx.x * kComposeBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kComposeBlock) {
            const uint32_t r = (uint32_t)(idx / vpr);
            const uint32_t col = (uint32_t)(idx - (uint64_t)r * vpr) * VEC;
            const uint8_t *p1 = a.b1 + (size_t)r * a.in_pitch + col, *p2 = a.b2 + (size_t)r * a.in_pitch + col;
            uint8_t *po = a.rgb + ((size_t)r * a.rgb_pitch_px + col) * 3;
            for (int j = 0; j < VEC && col + j < a.cols; ++j) {
                const uint32_t v1 = p1[j], v2 = p2[j];
                po[3 * j + 0] = R2[v1]; po[3 * j + 1] = G2[v2]; po[3 * j + 2] = B2[(v1 << 8) | v2];
            }
        }
    }
    if (SPEC) { // counts -> workgroup -> device; the workgroup that arrives last decides
        // vectors: #{x >= F} = (N + A_(F-1) - A_F) / 2 (all of them when F = 0), #{x >= F + 1} = (N + A_F - A_(F+1)) / 2: exact per lane
        n_ge[0] += fpred ? (n_vec_px + sad[0] - sad[1]) >> 1 : n_vec_px;
        n_ge[1] += (n_vec_px + sad[1] - sad[2]) >> 1;
        n_px += n_vec_px;
        uint32_t lt0 = n_px - n_ge[0], lt1 = n_px - n_ge[1];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { lt0 += __shfl_xor(lt0, m, 64); lt1 += __shfl_xor(lt1, m, 64); }
        // scratch in the table area (static __shared__ beside the opted-in 160 KiB of dynamic LDS is refused at launch); the
        // barrier makes sure no wave still reads the tables
        uint32_t *s_lt = reinterpret_cast<uint32_t *>(lds_raw);
        __syncthreads();
        if (threadIdx.x == 0) { s_lt[0] = 0u; s_lt[1] = 0u; }
        __syncthreads();
        if (lane == 0) { atomicAdd(&s_lt[0], lt0); atomicAdd(&s_lt[1], lt1); }
        __syncthreads();
        if (threadIdx.x == 0) {
            ChainSpecState *sp = a.spec;
            atomicAdd(&sp->n_lt[0], (unsigned long long)s_lt[0]);
            atomicAdd(&sp->n_lt[1], (unsigned long long)s_lt[1]);
            __threadfence();
            if (atomicAdd(&sp->done, 1u) == gridDim.x - 1u) {
                __threadfence();
                const unsigned long long c0 = atomicAdd(&sp->n_lt[0], 0ull), c1 = atomicAdd(&sp->n_lt[1], 0ull), target = sp->target;
                // floor_pred < kSpecFloorCap: F is the floor iff cum(F-1) < target <= cum(F); == kSpecFloorCap stands for "at least
                // that" (the +3 cushion is capped at 40 either way): true iff cum(kSpecFloorCap - 1) < target
                const bool ok = sp->floor_pred >= kSpecFloorCap ? c0 < target : (c0 < target && target <= c1);
                sp->verdict = ok ? 0u : 1u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// 6a. CLAHE blend of BOTH bands + suppressed synRGB composition in ONE sweep (the headline's fused pass): 4 B/px read (two u16
//     DN rasters) + 3 B/px written (interleaved RGB) = the 7 B/px of algorithmic traffic of save.rs:317-367 at native
//     resolution; no u8 level raster exists.  What stands between the blend and the composition in the reference -- the u8
//     rescale of each band (autoscale.rs:348-364) and the floor of the combined level histogram (synthetic_rgb.rs:99-113) -- is
//     proven / predicted from a sample-only pass over the same blend (kernel 4b, HIST == 3) by k_chain_predict, and the floor is
//     VERIFIED here exactly (the v_sad_u8 counts of kernel 6's speculative form, over the level bytes of both bands while they
//     are in registers); refuted, the gated apply -> finish -> compose kernels produce the raster instead.
//     Round 2's first attempt at this sweep lost to apply + compose; what differs: line-aligned column strips
//     (the misaligned ones cost the apply pass 10 %), 8 pixels per lane, the exact path inline instead of a queue + fixup pass
//     (0.18 ms), a sample-only pre-pass over tall items instead of a stratified sample (0.07 + 0.02 ms).
//     One persistent 1024-thread workgroup per CU; LDS (bytes): compose tables 66,048 | per-wave RGB stage 16 x 1,536 (it also
//     stages the bin-indexed entries while an item's tables are built) | f64 CDFs of both bands 2 x 257 x 32 (exact path) |
//     f64 dx of the strip's 512 columns | DN -> bin bytes | the pool of DN-INDEXED 16-byte entries (c00, c10, c01 - c00,
//     c11 - c10) of both bands: the pixel's gather address is its clamped DN x 16, no offset table in between.  Both windows
//     must fit the pool (kRgbPoolEntries); otherwise the pass leaves the verdict at "refuted" and the gated kernels run.
//     Per pixel and band: v_pk_min_u16 (clamp, 2 px) -> SDWA shift -> one 16-B gather -> 4 FMA + add -> 2 v_cvt_pk_u8_f32,
//     margin test and exact f64 path exactly as in kernel 4b (same bounds: the entry values and operation order are the same).
// ------------------------------------------------------------------------------------
// The blue table in LDS has rows of 260 bytes, not 256: the bank of B2[level1][level2] is then (level1 + level2 / 4) mod 32 instead of
// (level2 / 4) mod 32.  With 256-byte rows a band whose levels sit on a handful of values (a quantised cross-pol band: a dozen
// occupied CLAHE bins) sends all 32 lanes of a byte read to the two or three banks of those level2 values, in different rows:
// the pass ran 0.70 ms on such a scene against 0.64 on the others.  The index costs what it cost: one v_mad_u32_u24 (level1 x 260 +
// level2) instead of one v_perm_b32; the 1 KiB comes out of the pool (3008 DN-indexed entries instead of 3072).
#ifndef SARPRO_RGB_B2_STRIDE
#define SARPRO_RGB_B2_STRIDE 260
#endif
constexpr uint32_t kRgbB2Stride = SARPRO_RGB_B2_STRIDE;
static_assert(kRgbB2Stride == 256 || kRgbB2Stride == 260, "blue table rows: 256 (as in global memory) or 260 bytes");
#ifndef SARPRO_RGB_BLOCK // (experiments: -DSARPRO_RGB_BLOCK=512 = eight waves with up to 256 registers each -- 161 used, nothing spilled, and 0.65 ms against 0.60:
                         // the gathers want sixteen waves to hide behind; with -DSARPRO_RGB_POOL=2040 two such workgroups fit a compute unit)
#define SARPRO_RGB_BLOCK 1024
#endif
#ifndef SARPRO_RGB_POOL
#define SARPRO_RGB_POOL (SARPRO_RGB_B2_STRIDE == 256 ? 3072 : 3008)
#endif
constexpr int kRgbBlock = SARPRO_RGB_BLOCK, kRgbWaves = kRgbBlock / kWave;
constexpr uint32_t kRgbPoolEntries = SARPRO_RGB_POOL;
// (Round 5's "LITE" form -- no blue table in LDS, blue = rne(Pv[level1] * Qv[level2]) from two 256-entry f32 tables, so that a histogram
// workgroup fits beside the pass on a compute unit -- ran 8-10 % slower alone and 1.43 ms per pair co-resident: removed, NOTEBOOK.md 6e item 1.)
constexpr uint32_t kRgbTableBytes = 512 + 256 * kRgbB2Stride;
struct RgbLds {
    static constexpr uint32_t tables = 0;                                   // R2[256] | G2[256] | B2[256][kRgbB2Stride]  
    static constexpr uint32_t stage = kRgbTableBytes;                       // [16][1536] RGB of a wave-row | tmp32 [2][257] float4 while staging
    static constexpr uint32_t cdf64 = stage + kRgbWaves * 1536;             // [2][257][4] double
    static constexpr uint32_t colw = cdf64 + 2 * 257 * 32;                  // [512] double
    static constexpr uint32_t binof = colw + 512 * 8;                       // [kRgbPoolEntries] u8: CLAHE bin of the entry's DN
    static constexpr uint32_t pool = binof + kRgbPoolEntries;               // [kRgbPoolEntries] float4
    static constexpr uint32_t misc = pool + kRgbPoolEntries * 16;           // scratch words of the epilogue
    static constexpr uint32_t total = misc + 128;                           // (words 16..27: the next item's Rect)
};
// WIDE form (DN windows that do not fit the pool): the region [binof, misc) holds the DN -> bin bytes of both windows (loaded once per
// workgroup) and, at its end, the item's 2 x 257 bin-indexed 16-byte entries; a sample costs one more LDS byte read (its bin).
constexpr uint32_t kWideEnt = RgbLds::misc - 2u * 257u * 16u;    // [2][257] float4
constexpr uint32_t kWideBytes = kWideEnt - RgbLds::binof;         // capacity: win_hi[0] + win_hi[1] + 2 bytes
static_assert(kWideEnt % 16 == 0, "alignment");
static_assert(RgbLds::total <= 160 * 1024, "fused pass: LDS budget");
static_assert(RgbLds::stage % 16 == 0 && RgbLds::pool % 16 == 0 && RgbLds::cdf64 % 16 == 0, "alignment");

// GENERAL (ChainSpecState::spec_ok == kSpecRescaled): a band's lowest level is a prediction too.  The compose tables hold that band's
// u8 rescale folded in (k_chain_predict), the floor counts compare each band's level bytes with ITS thresholds (the lowest levels whose
// final values reach F and F + 1: the rescale is strictly increasing from min_pred on), and a fourth count -- level bytes below
// min_pred -- must be zero.  Eight more v_sad_u8 per lane and row pair than the identity form, which is why it is its own body.
// A refuted floor F: which floor the counts point to (c0 = band-pixels with final level < F).  c0 >= target: the floor lies below F;
// otherwise (the pass refuted it, so target > c1) above.  -1: no room on that side.
__device__ __forceinline__ int spec_retry_floor(int f, unsigned long long c0, unsigned long long target) {
    if (c0 >= target) return f > 0 ? f - 1 : -1;
    return f < kSpecFloorCap ? f + 1 : -1;
}
template <bool GENERAL>
__device__ __forceinline__ void clahe_rgb_fused_body(const ClaheRgbArgs &a) {
    // (declared HERE, not handed in by the kernel: as a pointer argument the compiler treated it as a flat address -- 104 bytes of scratch
    // and 25 spilled registers in the identity form against 76 and 18, and 8 % more bytes fetched by the pass, profiles/r5/pmc_traffic.txt)
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr int VEC = 8;
    ChainSpecState *sp = a.spec;
#ifdef SARPRO_RGB_WG_TIMES
    if (threadIdx.x == 0) g_rgb_wg_times[blockIdx.x & 1023][0] = wall_clock64();
    unsigned long long wg_t_wait = 0ull, wg_t_pro = 0ull, wg_t_rows = 0ull, wg_items = 0ull, wg_ta = 0ull, wg_tb = 0ull, wg_tc = 0ull;
#endif
    const uint32_t win_hi[2] = {a.dev_state[0].win_hi, a.dev_state[1].win_hi};
    const uint64_t nwin = (uint64_t)win_hi[0] + win_hi[1] + 2u;
    const bool wide = nwin > kRgbPoolEntries; // the windows do not fit the DN-indexed pool: bin-indexed entries behind a DN -> bin byte table
    if (wide && nwin > kWideBytes) { // ... nor the byte table: the verdict stays "refuted", the gated kernels run
        if (blockIdx.x == 0 && threadIdx.x == 0) sp->pool_overflow = 1u;
        return;
    }
    const uint32_t kb[2] = {0u, win_hi[0] + 1u}; // first pool entry (WIDE: first table byte) of each band
    const int lane = lane_id(), wave = wave_id();
    {   // the DN -> bin bytes of both windows, once per workgroup (the item loop's first barrier publishes them)
        for (int b = 0; b < 2; ++b)
            for (uint32_t i = threadIdx.x; i <= win_hi[b]; i += kRgbBlock) lds[RgbLds::binof + kb[b] + i] = a.binlut[b][i];
    }
    {   // compose tables, once per workgroup: R2 | G2 as they are, the blue table's 256-byte rows at their LDS stride
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.tables);
        for (int i = threadIdx.x; i < 512 / 4; i += kRgbBlock) LDS_AT(uint32_t, RgbLds::tables + 4 * i) = src[i];
        for (int i = threadIdx.x; i < 65536 / 4; i += kRgbBlock)
            LDS_AT(uint32_t, RgbLds::tables + 512 + (uint32_t)(i >> 6) * kRgbB2Stride + (uint32_t)(i & 63) * 4u) = src[128 + i];
    }
    const uint32_t fpred = (uint32_t)sp->floor_pred;
    const uint32_t t4[3] = {(fpred ? fpred - 1u : 0u) * 0x01010101u, fpred * 0x01010101u, (fpred + 1u) * 0x01010101u};
    uint32_t sad[3] = {0u, 0u, 0u}, n_all = 0u, n_kept = 0u; // this lane's |x - T| sums, level bytes seen (8 per band-row), kept ones
    // GENERAL: per band |x - T| sums at T0 - 1, T0, T0 + 1 (T0 = thr[b][0]; thr[b][1] is T0 or T0 + 1) and at min_pred - 1, min_pred
    uint32_t tg[2][5] = {{0u, 0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u, 0u}}, sadg[2][5] = {{0u, 0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u, 0u}};
    uint32_t min_track = 0u; // GENERAL: bit b = band b's lowest level is a prediction in 1 .. 127 (the byte test below needs min <= 127)
    if (GENERAL) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (sp->min_pred[b] >= 1u && sp->min_pred[b] <= 127u) min_track |= 1u << b;
        }
        min_track = to_sgpr_u32(min_track);
    }
    if (GENERAL) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t T0 = sp->thr[b][0], mn = sp->min_pred[b];
            tg[b][0] = min(T0 ? T0 - 1u : 0u, 255u) * 0x01010101u; tg[b][1] = min(T0, 255u) * 0x01010101u; tg[b][2] = min(T0 + 1u, 255u) * 0x01010101u;
            tg[b][3] = (mn ? mn - 1u : 0u) * 0x01010101u; tg[b][4] = mn * 0x01010101u;
        }
    }
    const RowWeight *__restrict__ row_w = a.row_w + a.row_off;
    const uint32_t stage_w = RgbLds::stage + (uint32_t)wave * 1536u;

    // Items are handed out by a device counter (ChainSpecState::next_item, cleared by k_chain_predict), in the list's order (sweep order:
    // the planner puts small items last).  With the static stride `item = blockIdx.x + k * gridDim.x` the workgroups ended up to 18 % apart
    // (13 items of different sizes each; the mean workgroup was done 7 % before the last: tools/rgb_wg_times.py).  The counter's round trip
    // is hidden behind the prologue: thread 0 asks for the NEXT item when this one starts and puts the answer into LDS after the
    // prologue's last barrier, so the value lives in a register through the prologue only -- not through the rows.
    uint32_t *const s_next = reinterpret_cast<uint32_t *>(lds + RgbLds::misc) + 8;
    // An item's Rect comes through LDS: twelve lanes of the first wave request the NEXT item's twelve words when this item's rows start and
    // put them into LDS when the rows are done -- the round trip that opened every prologue (Rect, then the CDFs it names) runs beside the
    // rows, for one register in the row loop.  (0.5-0.8 % of the pass, profiles/r6/ab_rect_prefetch.txt; with the 21 spilled registers the
    // pass had until its edge bytes were rewritten, the request was caught by the vmcnt(0) of a reload in the item's set-up and bought nothing.)
    uint32_t *const s_rect = reinterpret_cast<uint32_t *>(lds + RgbLds::misc) + 16;
    uint32_t rect_word = 0u;
    {
        uint32_t first = 0u;
        if (threadIdx.x == 0) { first = atomicAdd(&sp->next_item, 1u); s_next[0] = first; }
        if (wave == 0) {
            const int it0 = (int)to_sgpr_u32(first);
            if (lane < 12 && it0 < a.nrects) s_rect[lane] = reinterpret_cast<const uint32_t *>(a.rects + it0)[lane];
        }
    }
    for (;;) {
        uint32_t *const s_bsat = reinterpret_cast<uint32_t *>(lds + RgbLds::misc) + 4; // WIDE: per band the first saturated bin (256: none)
        if (wide && threadIdx.x < 2) s_bsat[threadIdx.x] = 256u; // (only the prologue reads it: no wave of the previous item does)
#ifdef SARPRO_RGB_WG_TIMES
        if (threadIdx.x == 0) { wg_ta = wall_clock64(); if (wg_items) wg_t_rows += wg_ta - wg_tc; }
#endif
        __syncthreads(); // the previous item's rows are done (its tables may go; the compose tables have landed)
#ifdef SARPRO_RGB_WG_TIMES
        if (threadIdx.x == 0) { wg_tb = wall_clock64(); wg_t_wait += wg_tb - wg_ta; }
#endif
        if (threadIdx.x == 0) s_next[1] = 2u * kRgbWaves; // the item's row counter (see the row loop): rows 0 .. 2 * 16 - 1 are assigned by wave number
        const int item = (int)to_sgpr_u32(s_next[0]);
        if (item >= a.nrects) break;
        uint32_t next_item = 0u;
        if (threadIdx.x == 0) next_item = atomicAdd(&sp->next_item, 1u); // (every workgroup overshoots the list once: the counter ends at nrects + grid)
        Rect rc;
        rc.r0 = (int)to_sgpr_u32(s_rect[0]); rc.r1 = (int)to_sgpr_u32(s_rect[1]); rc.c0 = (int)to_sgpr_u32(s_rect[2]); rc.c1 = (int)to_sgpr_u32(s_rect[3]);
        rc.cstart = (int)to_sgpr_u32(s_rect[4]);
#pragma unroll
        for (int k = 0; k < 4; ++k) rc.id[k] = (int)to_sgpr_u32(s_rect[5 + k]);
        rc.pad[0] = (int)to_sgpr_u32(s_rect[9]); rc.pad[1] = 0; rc.pad[2] = 0;
        // (Requesting the wave's first row of both bands HERE, before the tables are built, so that its round trip does not open the rows:
        // measured 0.620 ms against 0.600 -- ten more registers live through the prologue, 34 spilled instead of 21.)
        {
        // ---- this item's tables: bin-indexed (f64 for the exact path, biased f32 for staging), then expanded by DN
        for (int t = threadIdx.x; t < 2 * 257; t += kRgbBlock) {
            const int b = t / 257, bin = t - b * 257;
            double c[4] = {0.0, 0.0, 0.0, 0.0};
            if (bin < 256) {
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = a.cdfs[b][(size_t)rc.id[k] * 256 + bin];
            }
            *reinterpret_cast<double4 *>(lds + RgbLds::cdf64 + (b * 257 + bin) * 32) = make_double4(c[0], c[1], c[2], c[3]);
            // the f32 entry: as kernel 4b stages it (saturated interior bins -> 1.001, all-zero bins and the invalid entry -> 0.5 / 255)
            // (extrapolating cells: only with the saturation tables -- the entry then yields 255 and the row loop takes the 254s from them)
            const bool saturated = bin < 256 && c[0] == 1.0 && c[1] == 1.0 && c[2] == 1.0 && c[3] == 1.0 && (!(rc.pad[0] & 1) || a.sat_ok);
            const bool zero = c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0 && c[3] == 0.0;
            const float kz = 0.5f / 255.0f;
            const float c00 = (float)c[0], c01 = (float)c[1], c10 = (float)c[2], c11 = (float)c[3];
            const float4 e32 = saturated ? make_float4(1.001f, 1.001f, 0.0f, 0.0f)
                               : zero    ? make_float4(kz, kz, 0.0f, 0.0f)
                                         : make_float4(c00, c10, c01 - c00, c11 - c10);
            *reinterpret_cast<float4 *>(lds + RgbLds::stage + (b * 257 + bin) * 16) = e32;
            if (wide && saturated && (rc.pad[0] & 1)) atomicMin(&s_bsat[b], (uint32_t)bin);
        }
        for (int i = threadIdx.x; i < 512; i += kRgbBlock) {
            const int c2 = rc.cstart + i;
            *reinterpret_cast<double *>(lds + RgbLds::colw + i * 8) = a.col_w[(c2 >= rc.c0 && c2 < rc.c1) ? c2 : rc.c0].d;
        }
        uint32_t *const s_dnsat = reinterpret_cast<uint32_t *>(lds + RgbLds::misc) + 2; // per band: the first DN of a saturated bin (0xFFFF: none)
        if (threadIdx.x < 2) s_dnsat[threadIdx.x] = 0xFFFFu;
        __syncthreads();
        if (wide) { // the item's bin-indexed entries to their place; the first DN of a saturated bin by bisection of the (monotone) byte table
            for (int t = threadIdx.x; t < 2 * 257; t += kRgbBlock)
                *reinterpret_cast<float4 *>(lds + kWideEnt + t * 16) = *reinterpret_cast<const float4 *>(lds + RgbLds::stage + t * 16);
            if (threadIdx.x < 2 && (rc.pad[0] & 1)) {
                const uint32_t b = threadIdx.x, bs = s_bsat[b];
                uint32_t lo = 1u, hi = win_hi[b] + 1u; // first dn in [1, win_hi] whose bin >= bs; win_hi + 1: none
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint32_t)lds[RgbLds::binof + kb[b] + mid] >= bs) hi = mid; else lo = mid + 1u;
                }
                s_dnsat[b] = (bs < 256u && lo <= win_hi[b]) ? lo : 0xFFFFu;
            }
        } else { // the DN-indexed pool: consecutive threads take consecutive DNs (a wave's 16-byte writes fall side by side; the DN -> bin bytes
                 // stay in LDS for the whole pass -- round 5 fetched them per item, four DNs per thread: 0.4 % of the pass, profiles/r6/ab_expansion.txt)
            const uint32_t nwin32 = win_hi[0] + win_hi[1] + 2u;
            for (uint32_t i = threadIdx.x; i < nwin32; i += kRgbBlock) {
                const int qb = i >= kb[1] ? 1 : 0;
                const uint32_t dn = i - kb[qb];
                const uint32_t bin = dn ? (uint32_t)lds[RgbLds::binof + i] : 256u; // (DN = 0: the invalid entry; its byte in the table is never looked at)
                const float4 e = *reinterpret_cast<const float4 *>(lds + RgbLds::stage + ((uint32_t)qb * 257u + bin) * 16u);
                *reinterpret_cast<float4 *>(lds + RgbLds::pool + i * 16) = e;
                if (rc.pad[0] & 1) { // (bins and CDFs are monotone: every DN from the first saturated one on is saturated too -- a wave's first speaks for it)
                    const bool sat = e.x > 1.0005f;
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const unsigned long long m = __ballot(sat && qb == bb);
                        if (m && lane == __builtin_ctzll(m)) atomicMin(&s_dnsat[bb], dn);
                    }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_next[0] = next_item; // (every thread read the slot before the prologue's barriers; the next read is behind the barrier that ends this item)
        if (wave == 0) { // (the counter's answer is in thread 0's register: the first wave reads it from there)
            const int nx = (int)to_sgpr_u32(next_item);
            if (lane < 12 && nx < a.nrects) rect_word = reinterpret_cast<const uint32_t *>(a.rects + nx)[lane];
        }
        }
#ifdef SARPRO_RGB_WG_TIMES
        if (threadIdx.x == 0) { wg_tc = wall_clock64(); wg_t_pro += wg_tc - wg_tb; ++wg_items; }
#endif

        // The rows of the item, compiled twice and chosen per item by a wave-uniform branch: items of extrapolating cells (EDGE) and
        // the others.
        auto item_rows = [&](auto edge_tag, auto wide_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value, WIDE = decltype(wide_tag)::value;
        // EDGE: the lane's eight columns by saturation class (byte j = 1: column j is of class k), the bands' first saturated DN
        uint32_t satM[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
        uint32_t dnsat2[2] = {0xFFFEFFFEu, 0xFFFEFFFEu}; // (first saturated DN - 1) in both halves
        if (EDGE && a.sat_ok) {
            const int scol = min(rc.cstart + lane * VEC, (int)a.sat_cols - VEC);
            const uint2 tc = *reinterpret_cast<const uint2 *>(a.sat_col + scol);
            const uint32_t tcw[2] = {tc.x, tc.y};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const uint32_t x = tcw[g] ^ (0x01010101u * (uint32_t)k); // classes are 0..2: a byte is zero iff its two low bits are
                    satM[k][g] = ((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u;
                }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint32_t d = to_sgpr_u32(reinterpret_cast<const uint32_t *>(lds + RgbLds::misc)[2 + b]) - 1u;
                dnsat2[b] = (d & 0xFFFFu) | (d << 16);
            }
        }
        // ---- per-lane geometry of the strip
        const int col = rc.cstart + lane * VEC;
        uint32_t keep[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (col + j >= rc.c0 && col + j < rc.c1) keep[j >> 2] |= 0xFFu << (8 * (j & 3));
        const uint32_t nkeep = (uint32_t)__builtin_popcount(keep[0] & 0x01010101u) + (uint32_t)__builtin_popcount(keep[1] & 0x01010101u);
        // transposed store: the wave-row's 1536 RGB bytes leave as 96 chunks of 16 B (lanes 0..63, then lanes 0..31).  A chunk is
        // stored whole only if every pixel it touches belongs to the item; a kept pixel with a byte in a chunk that is not
        // stored goes out byte by byte from its own lane.  All-full strips (three of five per cell row): every chunk is safe.
        auto chunk_safe = [&](int k) {
            const int plo = (16 * k) / 3, phi = (16 * k + 15) / 3;
            return rc.cstart + plo >= rc.c0 && rc.cstart + phi < rc.c1;
        };
        const bool safe1 = chunk_safe(lane), safe2 = lane < 32 && chunk_safe(64 + lane);
        // The kept bytes of the chunks that are NOT stored -- at most fifteen at the item's left edge (from its first pixel to the end of the
        // last chunk that also holds a neighbour's bytes) and fifteen at its right edge -- leave as ONE byte store per row: lane l < 32 takes
        // byte l of the left run, lane 32 + l byte l of the right run, out of the wave's staged row; every other lane passes an out-of-range
        // offset.  (Round 3-5 wrote them from the lane that computed the pixel, inside a divergent branch: with stores on one path only, the
        // wait for the prefetched row at the loop's bottom -- vmcnt(2), the smaller of the two paths' counts -- waited for the write
        // acknowledgements of that row's own chunk stores in every row of the two strips in five that have such an edge: 5 % of the pass,
        // profiles/r6/ab_edge_bytes.txt.)
        uint32_t edge_voff = 0xFFFFFFFFu, edge_lds = stage_w;
        {
            const int L = rc.c0 - rc.cstart, R = min(rc.c1 - rc.cstart, 64 * VEC); // the item's pixels of the strip: [L, R)
            int nleft = 0, start_r = 0, nright = 0;
            if (L > 0) nleft = min(16 * ((3 * L - 1) / 16 + 1), 3 * R) - 3 * L;
            if (R < 64 * VEC) { start_r = max(16 * ((3 * R) / 16), 3 * L + nleft); nright = max(3 * R - start_r, 0); }
            const int eb = lane < 32 ? (lane < nleft ? 3 * L + lane : -1) : (lane - 32 < nright ? start_r + lane - 32 : -1);
            if (eb >= 0) { edge_voff = (uint32_t)eb; edge_lds = stage_w + (uint32_t)eb; }
        }
        // The two chunk stores of a row go through a buffer descriptor over the item's rows: a lane whose chunk is not stored passes
        // an out-of-range offset and the hardware drops its write.  EVERY path through a row therefore holds the same two store
        // (+ the edge bytes' one) instructions, and the wait for the prefetched row at the loop top is vmcnt(3) instead of vmcnt(0) -- with the stores inside
        // divergent branches the compiler must assume the row issued none, and vmcnt(0) also waits for the write acknowledgements
        // of the row just stored.  (Sending those lanes' chunks to a scratch line instead cost 13 %: a fifth of the pass's stores.)
        const size_t item_off = ((size_t)rc.r0 * a.rgb_pitch_px + (size_t)rc.cstart) * 3;
        const uint32_t row_bytes = (uint32_t)a.rgb_pitch_px * 3u;
        const __amdgpu_buffer_rsrc_t rgb_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.rgb + item_off, 0, (int)((uint32_t)(rc.r1 - rc.r0) * row_bytes), 0x00020000);
        const uint32_t voff1 = safe1 ? (uint32_t)lane * 16u : 0xFFFFFFFFu, voff2 = safe2 ? 1024u + (uint32_t)lane * 16u : 0xFFFFFFFFu;
        float dxf[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) dxf[j] = (float)*reinterpret_cast<const double *>(lds + RgbLds::colw + (lane * VEC + j) * 8);
        const float near_delta = spec_delta(rc.pad[0]);
        const float bias = -0.5f - near_delta, two_delta = 2.0f * near_delta;

        // the levels of one band's 8 samples, packed as bytes; masked to 0 outside the item
        auto band_levels = [&](int b, const uint4 &v, const double dy, const double omdy, const float wy1, const float wy2, const uint32_t (&satN)[2],
                               uint32_t (&pk)[2]) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const uint32_t hi2 = win_hi[b] | (win_hi[b] << 16);
            const uint32_t pbase = to_sgpr_u32(RgbLds::pool + kb[b] * 16u);
            uint32_t off[VEC];
            uint32_t sat01[4] = {0u, 0u, 0u, 0u}; // EDGE: per sample 1 (in its 16-bit half) when its bin is saturated
            uint32_t nz01[4] = {0u, 0u, 0u, 0u};  // WIDE: per sample 1 when it is valid (DN != 0): the byte table has no entry for "invalid"
            const uint32_t bbase = to_sgpr_u32(RgbLds::binof + kb[b]), ebase = to_sgpr_u32(kWideEnt + (uint32_t)b * 257u * 16u);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                typedef unsigned short v2us __attribute__((ext_vector_type(2)));
                const v2us c = __builtin_elementwise_min(__builtin_bit_cast(v2us, w[k]), __builtin_bit_cast(v2us, hi2));
                const uint32_t cw = __builtin_bit_cast(uint32_t, c);
                if (EDGE) { // min(sat_sub(DN, first saturated DN - 1), 1), both samples of the dword at once
                    const v2us one = {1, 1};
                    sat01[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_sub_sat(c, __builtin_bit_cast(v2us, dnsat2[b])), one));
                }
                uint32_t a0, a1;
                if (WIDE) { // the samples' bins first: byte address = table base + clamped DN
                    const v2us one = {1, 1};
                    nz01[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(c, one));
                    asm("v_mad_u32_u16 %0, %1, 1, %2" : "=v"(a0) : "v"(cw), "s"(bbase));
                    asm("v_mad_u32_u16 %0, %1, 1, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(cw), "s"(bbase));
                    off[2 * k] = a0; off[2 * k + 1] = a1;
                    continue;
                }
                // one instruction per sample: (16-bit half of the clamped pair) x 16 + the band's base in the pool
                asm("v_mad_u32_u16 %0, %1, 16, %2" : "=v"(a0) : "v"(cw), "s"(pbase));
                asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(cw), "s"(pbase));
                off[2 * k] = a0; off[2 * k + 1] = a1;
            }
            if (WIDE) { // bin -> entry address, all eight byte reads in flight before the first is used
                uint32_t bq[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) bq[j] = LDS_AT(uint8_t, off[j]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) off[j] = ebase + (bq[j] << 4);
            }
            uint32_t pb[2] = {0u, 0u};
            pk[0] = 0u; pk[1] = 0u;
#ifndef SARPRO_RGB_AHEAD
#define SARPRO_RGB_AHEAD 2
#endif
            constexpr int kAhead = SARPRO_RGB_AHEAD;
            v4f cq[VEC];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) cq[j] = LDS_AT(v4f, off[j]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (j + kAhead < VEC) cq[j + kAhead] = LDS_AT(v4f, off[j + kAhead]);
                const v4f c4 = cq[j];
                const float top = fmaf(c4.z, dxf[j], c4.x);
                const float bottom = fmaf(c4.w, dxf[j], c4.y);
                const float ya = fmaf(bottom, wy2, fmaf(top, wy1, bias));
                const float yb = ya + two_delta;
                pk[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(ya, j & 3, pk[j >> 2]);
                pb[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(yb, j & 3, pb[j >> 2]);
            }
            const uint32_t d0 = pk[0] ^ pb[0], d1 = pk[1] ^ pb[1];
            if (d0 | d1) { // rare: the exact f64 sequence of autoscale.rs:327-329, 602
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (((j < 4 ? d0 : d1) >> (8 * (j & 3))) & 0xFFu) {
                        const uint32_t bin = WIDE ? (off[j] - ebase) >> 4 : (uint32_t)lds[RgbLds::binof + ((off[j] - RgbLds::pool) >> 4)];
                        const double4 c4 = *reinterpret_cast<const double4 *>(lds + RgbLds::cdf64 + (b * 257 + bin) * 32);
                        const double dx = *reinterpret_cast<const double *>(lds + RgbLds::colw + (lane * VEC + j) * 8);
                        const double top = c4.x * (1.0 - dx) + c4.y * dx;
                        const double bottom = c4.z * (1.0 - dx) + c4.w * dx;
                        double o = top * omdy + bottom * dy;
                        o = fmin(fmax(o, 0.0), 1.0);
                        const uint32_t level = (uint32_t)(o * 255.0);
                        const uint32_t sh = 8 * (j & 3);
                        pk[j >> 2] = (pk[j >> 2] & ~(0xFFu << sh)) | (level << sh);
                    }
                }
            }
            if (EDGE) { // saturated samples came out as 255 (biased entry): those whose (row, column class) rounds below 1.0 are 254
                pk[0] -= __builtin_amdgcn_perm(sat01[1], sat01[0], to_sgpr_u32(0x06040200u)) & satN[0];
                pk[1] -= __builtin_amdgcn_perm(sat01[3], sat01[2], to_sgpr_u32(0x06040200u)) & satN[1];
            }
            if (WIDE) { // invalid samples (DN = 0) went through some bin's entry: their level is 0
                pk[0] &= __builtin_amdgcn_perm(nz01[1], nz01[0], to_sgpr_u32(0x06040200u)) * 0xFFu;
                pk[1] &= __builtin_amdgcn_perm(nz01[3], nz01[2], to_sgpr_u32(0x06040200u)) * 0xFFu;
            }
            pk[0] &= keep[0];
            pk[1] &= keep[1];
        };

        {   // EVERY lane walks the rows, also those whose 8 columns lie outside the item (their levels are masked, nothing of
            // theirs is stored): the transposed store hands chunk `lane` to lane `lane`, whatever columns that lane computes
            const int step = kRgbWaves;
            const int lcol = min(col, (int)a.in_pitch - VEC); // a lane past the row's pitch loads (and ignores) the row's last vector
            const uint16_t *p0 = a.in[0] + lcol, *p1 = a.in[1] + lcol;
            int r = __builtin_amdgcn_readfirstlane(rc.r0 + wave);
            // The waves of a workgroup do not advance together (the oldest wave of a SIMD wins every arbitration): with rows r0 + wave + 16 k
            // the first wave stood at the item's closing barrier 18 us of every 70 (tools/rgb_wg_times.py), and while the waves trickle in the
            // compute unit runs ever emptier.  Rows are handed out by an LDS counter instead: a wave's first two rows by its number, then the
            // row after next with every row it starts -- the answer travels behind the row's own LDS gathers.
            int kn = wave + step; // the NEXT row of this wave, relative to r0 (>= the item's rows: none)
            const int nrows = rc.r1 - rc.r0;
            if (r < rc.r1) {
                uint4 c0 = *reinterpret_cast<const uint4 *>(p0 + (size_t)r * a.in_pitch), c1 = *reinterpret_cast<const uint4 *>(p1 + (size_t)r * a.in_pitch);
                double dyv = row_w[r].d;
                uint32_t srv = (EDGE && a.sat_ok) ? (uint32_t)a.sat_row[r] : 7u;
                // (three stores behind the first row's loads, as every later row has them behind its own: the loop is entered in its
                // steady state and its top waits with vmcnt(3))
                {
                    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                    const v4u z = {0u, 0u, 0u, 0u};
                    __builtin_amdgcn_raw_buffer_store_b128(z, rgb_rsrc, 0xFFFFFFFFu, 0, 2); // (out of range in every lane: nothing is written)
                    __builtin_amdgcn_raw_buffer_store_b128(z, rgb_rsrc, 0xFFFFFFF0u, 0, 2); // (another offset: two identical stores would be merged)
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rgb_rsrc, 0xFFFFFFFFu, 0, 0);
                }
                for (;;) {
                    const int rn = rc.r0 + min(kn, nrows - 1); // the next row of both bands is always in flight
                    uint32_t grab = 0u;
                    if (lane == 0) grab = atomicAdd(&s_next[1], 1u);
                    const uint4 n0 = *reinterpret_cast<const uint4 *>(p0 + (size_t)rn * a.in_pitch), n1 = *reinterpret_cast<const uint4 *>(p1 + (size_t)rn * a.in_pitch);
                    const double dyn = row_w[rn].d;
                    const uint32_t srn = (EDGE && a.sat_ok) ? (uint32_t)a.sat_row[rn] : 7u;
                    uint32_t satN[2] = {0u, 0u}; // bytes of the lane's samples that get 254 when saturated, in this row
                    if (EDGE) {
                        const uint32_t rb = to_sgpr_u32(srv);
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            if (!((rb >> k) & 1u)) { satN[0] |= satM[k][0]; satN[1] |= satM[k][1]; }
                    }
                    const double dy = to_sgpr(dyv), omdy = to_sgpr(1.0 - dy);
                    const float wy1 = to_sgpr((float)omdy * 255.0f), wy2 = to_sgpr((float)dy * 255.0f);
                    uint32_t l1[2], l2[2];
                    band_levels(0, c0, dy, omdy, wy1, wy2, satN, l1);
                    band_levels(1, c1, dy, omdy, wy1, wy2, satN, l2);
                    // verification counts (kernel 6, SPEC): |x - T| over the 16 level bytes
                    if (GENERAL) {
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            sadg[0][k] = __builtin_amdgcn_sad_u8(l1[1], tg[0][k], __builtin_amdgcn_sad_u8(l1[0], tg[0][k], sadg[0][k]));
                            sadg[1][k] = __builtin_amdgcn_sad_u8(l2[1], tg[1][k], __builtin_amdgcn_sad_u8(l2[0], tg[1][k], sadg[1][k]));
                        }
                        // A kept byte below its band's predicted lowest level refutes the prediction; which level it is decides what the second
                        // pass tries (ChainSpecState::true_min).  Per dword: bit 7 of byte i of ((x & 0x7f..) | 0x80..) - min x 0x01.. is set iff
                        // the byte's low seven bits reach min (no borrow crosses a byte: every byte starts at 0x80 or more and min <= 127), so
                        // a byte is below iff that bit AND its own bit 7 are clear; bytes outside the item (masked to 0) are taken out by `keep`.
                        // Rare by construction (the sample held every level but the rarest): the branch is taken on a handful of rows per scene.
                        if (min_track) {
                            const uint32_t lv[2][2] = {{l1[0], l1[1]}, {l2[0], l2[1]}};
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                uint32_t below[2];
#pragma unroll
                                for (int g = 0; g < 2; ++g)
                                    below[g] = ~(((lv[b][g] & 0x7F7F7F7Fu) | 0x80808080u) - tg[b][4]) & ~lv[b][g] & keep[g] & 0x80808080u;
                                if (((min_track >> b) & 1u) && (below[0] | below[1])) {
                                    uint32_t lowest = 256u;
#pragma unroll
                                    for (int j = 0; j < VEC; ++j)
                                        if ((below[j >> 2] >> (8 * (j & 3))) & 0x80u) lowest = min(lowest, (lv[b][j >> 2] >> (8 * (j & 3))) & 0xFFu);
                                    atomicMin(&sp->true_min[b], lowest);
                                    atomicAdd(&sp->below_hist[b][lowest & 127u], 1ull); // (row stripes: the ranks' sums say which levels occurred)
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            sad[k] = __builtin_amdgcn_sad_u8(l2[1], t4[k], __builtin_amdgcn_sad_u8(l2[0], t4[k], __builtin_amdgcn_sad_u8(l1[1], t4[k], __builtin_amdgcn_sad_u8(l1[0], t4[k], sad[k]))));
                    }
                    n_all += 16u; n_kept += 2u * nkeep;
                    // composition (synthetic_rgb.rs:158-175 through the folded tables): 8 px -> 24 bytes
                    uint32_t o[6];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        uint32_t px[4][3];
                        // Table addresses straight from the packed level bytes: one bit-field extract per R2 / G2 index, ONE v_perm_b32 per
                        // B2 index ((level1 << 8) | level2 from the two packed registers), the table bases in the instructions' offset
                        // fields; the 12 bytes are packed by v_lshl_or pairs (left to the compiler the same lines cost twice the VALU
                        // instructions: byte masks after zero-extending loads, multiply-adds for the pair index, three-step packing).
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t v1 = j == 0 ? (l1[g] & 0xFFu) : j == 3 ? (l1[g] >> 24) : __builtin_amdgcn_ubfe(l1[g], 8 * j, 8);
                            const uint32_t v2 = j == 0 ? (l2[g] & 0xFFu) : j == 3 ? (l2[g] >> 24) : __builtin_amdgcn_ubfe(l2[g], 8 * j, 8);
                            uint32_t pair; // byte offset of B2[level1][level2]
                            if (kRgbB2Stride == 256) pair = __builtin_amdgcn_perm(l1[g], l2[g], to_sgpr_u32(0x0c0c0400u + 0x0101u * (uint32_t)j)); // (0, 0, level1, level2)
                            else asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pair) : "v"(v1), "s"(kRgbB2Stride), "v"(v2));
                            px[j][0] = LDS_AT(uint8_t, RgbLds::tables + v1);
                            px[j][1] = LDS_AT(uint8_t, RgbLds::tables + 256 + v2);
                            px[j][2] = LDS_AT(uint8_t, RgbLds::tables + 512 + pair);
                        }
                        o[3 * g + 0] = pack4(px[0][0], px[0][1], px[0][2], px[1][0]);
                        o[3 * g + 1] = pack4(px[1][1], px[1][2], px[2][0], px[2][1]);
                        o[3 * g + 2] = pack4(px[2][2], px[3][0], px[3][1], px[3][2]);
                    }
                    // (same wave writes and reads its stage: program order inside a wave, no barrier)
                    *reinterpret_cast<uint2 *>(lds + stage_w + lane * 24) = make_uint2(o[0], o[1]);
                    *reinterpret_cast<uint2 *>(lds + stage_w + lane * 24 + 8) = make_uint2(o[2], o[3]);
                    *reinterpret_cast<uint2 *>(lds + stage_w + lane * 24 + 16) = make_uint2(o[4], o[5]);
                    {
                        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                        const uint32_t soff = (uint32_t)(r - rc.r0) * row_bytes;
                        // GUARDS of what follows (keep both): tools/check_store_hazard.py, run by __graft_entry__.build(), disassembles the
                        // library and fails the build when a VALU write of a 128-bit store's data register sits in the slot behind the
                        // store; tests/test_gpu_spec_chain.py::test_fused_rgb_route_equals_the_other_routes_at_36mp and the 400 MP rasters of
                        // tests/test_gpu_full_size_oracle.py catch the corrupted pixel pairs themselves (they only show from 36 MP up).
                        // gfx950 corrupts the first dword of a 128-bit buffer store's data when a VALU instruction writes that VGPR
                        // in the very next issue slot; the compiler only guards the form WITHOUT an SGPR offset (and, left alone,
                        // put the next chunk's address computation into the data register right behind the store: one pixel pair per
                        // chunk boundary came out wrong, at 36 MP and up).  Both chunks are read first, and an s_nop that names the
                        // data registers follows each store: they stay live, and two wait states pass, before anything may touch them.
                        const v4u d1 = LDS_AT(v4u, stage_w + lane * 16), d2 = LDS_AT(v4u, stage_w + 1024 + lane * 16);
                        const uint32_t e8 = LDS_AT(uint8_t, edge_lds);
                        __builtin_amdgcn_raw_buffer_store_b128(d1, rgb_rsrc, voff1, soff, 2 /* nt */);
                        asm volatile("s_nop 1" : : "v"(d1) : "memory");
                        __builtin_amdgcn_raw_buffer_store_b128(d2, rgb_rsrc, voff2, soff, 2);
                        asm volatile("s_nop 1" : : "v"(d2) : "memory");
                        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)e8, rgb_rsrc, edge_voff, soff, 0);
                    }
                    c0 = n0; c1 = n1; dyv = dyn; srv = srn;
                    if (kn >= nrows) break;
                    r = rc.r0 + kn;
                    kn = (int)to_sgpr_u32(grab);
                }
            }
        }
        };
        if (!wide) {
            if (rc.pad[0] & 1) item_rows(std::true_type{}, std::false_type{});
            else item_rows(std::false_type{}, std::false_type{});
        } else {
            if (rc.pad[0] & 1) item_rows(std::true_type{}, std::true_type{});
            else item_rows(std::false_type{}, std::true_type{});
        }
        if (wave == 0 && lane < 12) s_rect[lane] = rect_word; // (every wave read this item's Rect before the prologue's barriers)
    }
    // ---- counts -> workgroup -> device; the workgroup that arrives last decides (as kernel 6, SPEC)
#ifdef SARPRO_RGB_WG_TIMES
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *o = g_rgb_wg_times[blockIdx.x & 1023];
        o[1] = wall_clock64(); o[2] = wg_t_wait; o[3] = wg_t_pro; o[4] = wg_t_rows; o[5] = wg_items;
    }
#endif
    uint32_t lt0, lt1, below = 0u;
    if (GENERAL) {
        lt0 = 0u; lt1 = 0u;
        const uint32_t na = n_all >> 1, nk = n_kept >> 1; // per band
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint32_t T0 = sp->thr[b][0], T1 = sp->thr[b][1], mn = sp->min_pred[b];
            // bytes >= T of this band: everything kept for T = 0 (the masked bytes are 0), nothing for T = 256
            const uint32_t ge_t0 = T0 == 0u ? nk : T0 > 255u ? 0u : (na + sadg[b][0] - sadg[b][1]) >> 1;
            const uint32_t ge_t0p = T0 + 1u > 255u ? 0u : (na + sadg[b][1] - sadg[b][2]) >> 1;
            const uint32_t ge_t1 = T1 == T0 ? ge_t0 : ge_t0p;
            const uint32_t ge_mn = mn == 0u ? nk : (na + sadg[b][3] - sadg[b][4]) >> 1;
            lt0 += nk - ge_t0; lt1 += nk - ge_t1; below += nk - ge_mn;
        }
    } else {
        const uint32_t ge0 = fpred ? (n_all + sad[0] - sad[1]) >> 1 : n_kept, ge1 = (n_all + sad[1] - sad[2]) >> 1;
        lt0 = n_kept - ge0; lt1 = n_kept - ge1;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { lt0 += __shfl_xor(lt0, m, 64); lt1 += __shfl_xor(lt1, m, 64); if (GENERAL) below += __shfl_xor(below, m, 64); }
    uint32_t *s_lt = reinterpret_cast<uint32_t *>(lds + RgbLds::misc);
    __syncthreads();
    if (threadIdx.x == 0) { s_lt[0] = 0u; s_lt[1] = 0u; s_lt[2] = 0u; }
    __syncthreads();
    if (lane == 0) { atomicAdd(&s_lt[0], lt0); atomicAdd(&s_lt[1], lt1); if (GENERAL) atomicAdd(&s_lt[2], below); }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sp->n_lt[0], (unsigned long long)s_lt[0]);
        atomicAdd(&sp->n_lt[1], (unsigned long long)s_lt[1]);
        if (GENERAL) atomicAdd(&sp->n_below_min, (unsigned long long)s_lt[2]);
        __threadfence();
        if (!a.no_verdict && atomicAdd(&sp->done, 1u) == gridDim.x - 1u) {
            __threadfence();
            const unsigned long long c0 = atomicAdd(&sp->n_lt[0], 0ull), c1 = atomicAdd(&sp->n_lt[1], 0ull), target = sp->target;
            const unsigned long long under = GENERAL ? atomicAdd(&sp->n_below_min, 0ull) : 0ull;
            const bool ok = (sp->floor_pred >= kSpecFloorCap ? c0 < target : (c0 < target && target <= c1)) && under == 0ull;
            sp->verdict = ok ? 0u : 1u;
            sp->retry_floor = ok || under != 0ull ? -1 : spec_retry_floor(sp->floor_pred, c0, target);
            // an undercut lowest level: the second pass can run iff every undercut band's true lowest level was recorded (a band whose
            // prediction lies above 127 is not tracked) -- the bytes below were counted over BOTH bands, so both records are looked at
            bool min_known = false;
            if (GENERAL && under != 0ull) {
                const uint32_t t0 = atomicMin(&sp->true_min[0], 256u), t1 = atomicMin(&sp->true_min[1], 256u);
                min_known = (t0 < sp->min_pred[0] || t1 < sp->min_pred[1]) && (sp->min_pred[0] <= 127u) && (sp->min_pred[1] <= 127u);
            }
            sp->retry_min = min_known ? 1u : 0u;
            if (a.retry) sp->retried = 1u;
        }
    }
}
// Two kernels, two launches (the second returns at once on all but the rare scene): with both bodies in ONE kernel the identity form ran
// 2 % slower (0.697-0.707 ms against 0.681-0.693 on the same box, the register allocator spills for the union of the two), the extra
// launch costs nothing measurable (host time of the chain 1.179-1.190 ms against 1.186-1.195).
// (The rescaled body as a noinline function called from the one kernel: 0.86-0.89 ms against 0.715 -- the call ABI costs the hot loop far
// more than the spills.)  Inside a resident batch the second launch's empty workgroups still wait for whole compute units (160 KiB of LDS
// each): its bracket reads 15 us on average there, on the lane's own stream, behind which only the gated fallbacks follow.
__global__ __launch_bounds__(kRgbBlock) void k_clahe_rgb_fused(ClaheRgbArgs a) {
    if (a.spec->spec_ok == kSpecIdentity) clahe_rgb_fused_body<false>(a);
}
__global__ __launch_bounds__(kRgbBlock) void k_clahe_rgb_fused_rescaled(ClaheRgbArgs a) {
    if (a.spec->spec_ok == kSpecRescaled) clahe_rgb_fused_body<true>(a);
}
// The second pass of a refuted floor (k_chain_repredict has set it up): both forms in ONE kernel -- it runs on a scene in fifty, its launch
// is on every scene's chain (2 % of the pass against a launch).
__global__ __launch_bounds__(kRgbBlock) void k_clahe_rgb_fused_retry(ClaheRgbArgs a) {
    if (a.spec->retry_armed != 1u) return;
    if (a.spec->spec_ok == kSpecIdentity) clahe_rgb_fused_body<false>(a);
    else if (a.spec->spec_ok == kSpecRescaled) clahe_rgb_fused_body<true>(a);
}
// Row stripes: every rank's pass has added its counts, the ranks have summed them; the same verdict on every rank.
__global__ void k_spec_verdict(ChainSpecState *sp, const ChainBandState *state, int second) {
    if (second && !sp->retry_armed) { // no second pass ran: the all-reduce behind it summed zeros (k_chain_repredict set the counts aside); the first verdict stands
        sp->n_lt[0] = sp->saved_counts[0]; sp->n_lt[1] = sp->saved_counts[1]; sp->n_below_min = sp->saved_counts[2];
        return;
    }
    // (the windows are the same on every rank; `pool_overflow` is raised by block 0 of the pass, which a rank with an empty stripe
    // never launches: every rank derives it here, so that verdict, floor and report agree across the ranks)
    const uint64_t nwin = (uint64_t)state[0].win_hi + state[1].win_hi + 2u;
    if (sp->spec_ok && nwin > kRgbPoolEntries && nwin > kWideBytes) sp->pool_overflow = 1u;
    if (!sp->spec_ok || sp->pool_overflow) return; // the passes did not run: "refuted" stands
    const unsigned long long c0 = sp->n_lt[0], c1 = sp->n_lt[1], target = sp->target;
    const bool undercut = sp->spec_ok == kSpecRescaled && sp->n_below_min != 0ull;
    const bool ok = (sp->floor_pred >= kSpecFloorCap ? c0 < target : (c0 < target && target <= c1)) && !undercut;
    sp->verdict = ok ? 0u : 1u;
    sp->retry_floor = ok || undercut ? -1 : spec_retry_floor(sp->floor_pred, c0, target);
    if (undercut && !second) { // the true lowest levels from the ranks' summed presence counts; every rank reads the same sums
        bool known = sp->min_pred[0] <= 127u && sp->min_pred[1] <= 127u, any = false;
        for (int b = 0; b < 2; ++b) {
            uint32_t tm = 256u;
            for (uint32_t l = 0; l < sp->min_pred[b] && l < 128u; ++l)
                if (sp->below_hist[b][l]) { tm = l; break; }
            sp->true_min[b] = tm;
            any = any || tm < sp->min_pred[b];
        }
        sp->retry_min = known && any ? 1u : 0u;
    }
    if (sp->retry_armed) sp->retried = 1u; // (this is the second pass's verdict)
}

// ------------------------------------------------------------------------------------
// 6b. Fused calibrate -> stretch -> compose for the percentile strategies (dual-pol, u8): both DN rasters
//     in, interleaved RGB out, ONE pass.  Per band the whole chain dB -> clip -> gamma -> quantise -> u8
//     rescale is one host-built table of the DN (only its window [0, win_hi] is staged in LDS; above it
//     the table is constant), then the synRGB tables of kernel 6.  No intermediate raster exists.
//     Algorithmic traffic = actual traffic: 4 B/px read + 3 B/px written = 7 B/px.
// ------------------------------------------------------------------------------------
template <bool GLOBAL_LUT>
__device__ __forceinline__ void lut_compose_rows(const LutComposeArgs &a, unsigned char *lds_raw, uint32_t hi1, uint32_t hi2,
                                                 uint32_t lut2_off) {
    const uint8_t *R2 = lds_raw, *G2 = lds_raw + 256, *B2 = lds_raw + 512;
    const uint8_t *lut1 = lds_raw + kComposeTableBytes + kComposeStageBytes, *lut2 = lut1 + lut2_off;
    auto look1 = [&](uint32_t d) -> uint32_t { return GLOBAL_LUT ? a.lut[0][min(d, hi1)] : lut1[min(d, hi1)]; };
    auto look2 = [&](uint32_t d) -> uint32_t { return GLOBAL_LUT ? a.lut[1][min(d, hi2)] : lut2[min(d, hi2)]; };
    uint4 *stage = reinterpret_cast<uint4 *>(lds_raw + kComposeTableBytes + (threadIdx.x >> 6) * 3072);
    const int lane = threadIdx.x & 63;
    const uint32_t vpr = (a.cols + 15) / 16, wpr = (vpr + 63) / 64;
    const uint64_t chunks = (uint64_t)a.rows * wpr;
    const uint64_t nwaves = (uint64_t)gridDim.x * (kComposeBlock / kWave);
    // software pipeline: the loads of the wave's NEXT chunk are issued before this chunk is looked up and stored
    // (clamped to the last chunk, so the loads are unconditional and the compiler can count them)
    auto chunk_ptr = [&](uint64_t ch, int band) {
        const uint32_t r = (uint32_t)(ch / wpr);
        const uint32_t col = ((uint32_t)(ch - (uint64_t)r * wpr) * 64 + lane) * 16;
        return a.in[band] + (size_t)r * a.in_pitch + min(col, (a.in_pitch >= 16 ? (uint32_t)a.in_pitch - 16 : 0u));
    };
    uint64_t ch = (uint64_t)blockIdx.x * (kComposeBlock / kWave) + (threadIdx.x >> 6);
    uint4 na, nb, nc, nd;
    if (ch < chunks) {
        const uint4 *q1 = reinterpret_cast<const uint4 *>(chunk_ptr(ch, 0)), *q2 = reinterpret_cast<const uint4 *>(chunk_ptr(ch, 1));
        na = q1[0]; nb = q1[1]; nc = q2[0]; nd = q2[1];
    }
    for (; ch < chunks; ch += nwaves) {
        const uint32_t r = (uint32_t)(ch / wpr);
        const uint32_t v0 = (uint32_t)(ch - (uint64_t)r * wpr) * 64;
        const uint32_t col = (v0 + lane) * 16;
        const bool fullv = col + 16 <= a.cols;
        const uint32_t nfull = (a.cols / 16 > v0) ? min(64u, a.cols / 16 - v0) : 0u;
        const uint16_t *p1 = a.in[0] + (size_t)r * a.in_pitch + col, *p2 = a.in[1] + (size_t)r * a.in_pitch + col;
        const uint4 qa = na, qb = nb, qc = nc, qd = nd;
        {
            const uint64_t nx = ch + nwaves < chunks ? ch + nwaves : ch;
            const uint4 *q1 = reinterpret_cast<const uint4 *>(chunk_ptr(nx, 0)), *q2 = reinterpret_cast<const uint4 *>(chunk_ptr(nx, 1));
            na = q1[0]; nb = q1[1]; nc = q2[0]; nd = q2[1];
        }
        if (fullv) {
            const uint32_t w1[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            const uint32_t w2[8] = {qc.x, qc.y, qc.z, qc.w, qd.x, qd.y, qd.z, qd.w};
            uint32_t o[12];
#pragma unroll
            for (int g = 0; g < 4; ++g) { // 4 px -> 12 bytes -> 3 dwords
                uint32_t px[4][3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = g * 4 + j;
                    const uint32_t d1 = (k & 1) ? (w1[k >> 1] >> 16) : (w1[k >> 1] & 0xFFFFu);
                    const uint32_t d2 = (k & 1) ? (w2[k >> 1] >> 16) : (w2[k >> 1] & 0xFFFFu);
                    const uint32_t v1 = look1(d1), v2 = look2(d2);
                    px[j][0] = R2[v1]; px[j][1] = G2[v2]; px[j][2] = B2[(v1 << 8) | v2];
                }
                o[3 * g + 0] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
                o[3 * g + 1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
                o[3 * g + 2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
            }
            stage[lane * 3 + 0] = make_uint4(o[0], o[1], o[2], o[3]);
            stage[lane * 3 + 1] = make_uint4(o[4], o[5], o[6], o[7]);
            stage[lane * 3 + 2] = make_uint4(o[8], o[9], o[10], o[11]);
        }
        uint8_t *rowp = a.rgb + ((size_t)r * a.rgb_pitch_px + (size_t)v0 * 16) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t slot = k * 64 + lane;
            if (slot < nfull * 3) store_stream16(reinterpret_cast<uint4 *>(rowp) + slot, stage[slot]);
        }
        if (!fullv && col < a.cols) { // ragged tail of the row: scalar
            uint8_t *po = a.rgb + ((size_t)r * a.rgb_pitch_px + col) * 3;
            for (uint32_t j = 0; col + j < a.cols; ++j) {
                const uint32_t v1 = look1(p1[j]), v2 = look2(p2[j]);
                po[3 * j + 0] = R2[v1]; po[3 * j + 1] = G2[v2]; po[3 * j + 2] = B2[(v1 << 8) | v2];
            }
        }
    }
}

__global__ __launch_bounds__(kComposeBlock) void k_lut_compose_u16(LutComposeArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.tables);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_raw);
        for (int i = threadIdx.x; i < kComposeTableBytes / 16; i += kComposeBlock) dst[i] = src[i];
    }
    // window edges: from the host (host-orchestrated path) or from device memory (chain).  When a window exceeds
    // the LDS capacity chosen at launch both tables are gathered from global memory instead (same results).
    const uint32_t wh1 = a.dev_state ? a.dev_state[0].win_hi : a.win_hi[0], wh2 = a.dev_state ? a.dev_state[1].win_hi : a.win_hi[1];
    const uint32_t cap = a.dev_state ? a.lut_cap : 65536u;
    const bool global_lut = wh1 >= cap || wh2 >= cap;
    const uint32_t lut2_off = a.dev_state ? cap : ((wh1 + 16) & ~15u);
    if (!global_lut) {
        uint8_t *lut1 = lds_raw + kComposeTableBytes + kComposeStageBytes, *lut2 = lut1 + lut2_off;
        for (uint32_t i = threadIdx.x; i <= wh1; i += kComposeBlock) lut1[i] = a.lut[0][i];
        for (uint32_t i = threadIdx.x; i <= wh2; i += kComposeBlock) lut2[i] = a.lut[1][i];
    }
    __syncthreads();
    if (global_lut) lut_compose_rows<true>(a, lds_raw, wh1, wh2, lut2_off);
    else lut_compose_rows<false>(a, lds_raw, wh1, wh2, lut2_off);
}

// ------------------------------------------------------------------------------------
// 7. Polarisation operations (ops.rs:4-44), IEEE f32, correctly rounded division.
// ------------------------------------------------------------------------------------
__device__ inline float polop_one(int op, float x, float y) {
    switch (op) {
    case SARPRO_OP_SUM: return x + y;
    case SARPRO_OP_DIFF: return x - y;
    case SARPRO_OP_RATIO:
    case SARPRO_OP_LOGRATIO: return fabsf(y) > 1e-10f ? x / y : 0.0f;
    default: { const float d = x + y; return fabsf(d) > 1e-10f ? (x - y) / d : 0.0f; }
    }
}
__global__ __launch_bounds__(kBlock) void k_polop_f32(int op, const float *__restrict__ a,
                                                      const float *__restrict__ b, size_t n,
                                                      float *__restrict__ out, int vec_ok) {
    const size_t nv = vec_ok ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        float4 o;
        o.x = polop_one(op, x.x, y.x); o.y = polop_one(op, x.y, y.y);
        o.z = polop_one(op, x.z, y.z); o.w = polop_one(op, x.w, y.w);
        reinterpret_cast<float4 *>(out)[i] = o;
    }
    for (size_t i = nv * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = polop_one(op, a[i], b[i]);
}

// ------------------------------------------------------------------------------------
// 8. 256-bin histogram of a u8 raster (standalone suppressed synRGB, synthetic_rgb.rs:92-98).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_hist256_u8(const uint8_t *__restrict__ in, size_t pitch,
                                                       uint32_t rows, uint32_t cols,
                                                       unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[kWavesPerBlock][256]; // one copy per wave: fewer same-address collisions
    for (int i = threadIdx.x; i < kWavesPerBlock * 256; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
    const uint64_t total = (uint64_t)rows * cols;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(idx / cols);
        const uint32_t c = (uint32_t)(idx - (uint64_t)r * cols);
        atomicAdd(&mine[in[(size_t)r * pitch + c]], 1u);
    }
    __syncthreads();
    uint32_t n = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) n += h[w][threadIdx.x];
    if (n) atomicAdd(&hist[threadIdx.x], (unsigned long long)n);
}

// ------------------------------------------------------------------------------------
// 8b. In-place u8 -> u8 remap (the rare case where scale_u16_to_u8 of a CLAHE level raster is
//     not the identity and the caller wants the per-band u8 raster itself).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_remap_u8(uint8_t *__restrict__ buf, size_t pitch, uint32_t rows,
                                                     uint32_t cols, const uint8_t *__restrict__ map) {
    __shared__ uint8_t m[256];
    m[threadIdx.x] = map[threadIdx.x];
    __syncthreads();
    const uint64_t total = (uint64_t)rows * cols;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(idx / cols);
        const uint32_t c = (uint32_t)(idx - (uint64_t)r * cols);
        uint8_t *p = buf + (size_t)r * pitch + c;
        *p = m[*p];
    }
}

// ------------------------------------------------------------------------------------
// 9. Synthetic scene generator (bench / tests; mirrors sarpro_amd/synth.py exactly).
// ------------------------------------------------------------------------------------
__device__ inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void k_synth_scene_u16(uint64_t key, const uint16_t *__restrict__ q,
                                                            uint64_t rows_total, uint64_t cols, uint64_t row0,
                                                            uint64_t rows_local, uint64_t block,
                                                            uint16_t *__restrict__ out, size_t pitch, uint32_t flags) {
    const uint64_t total = rows_local * cols;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t rl = i / cols, c = i - rl * cols, r = row0 + rl;
        const uint64_t h = splitmix64((r * cols + c) ^ key);
        const uint32_t map = (flags >> 4) & 15u; // sarpro_hip.h: SARPRO_HIP_SYNTH_* (class map, wedges, bright targets)
        const uint64_t cls = map == 0 ? ((r / block) + 3 * (c / block)) & 3 : map == 1 ? ((r / block) ^ (c / block)) & 3
                             : map == 2 ? ((r + c) / block) & 3 : 1;
        uint32_t dn = q[cls * 65536 + (h >> 48)];
        if (!(flags & 2u) && ((h >> 20) % 10000ull) == 0) dn = (uint32_t)((20000ull + (h & 0x7FFFull)) & 0xFFFFull);
        const bool left = c * rows_total * 100ull < 3ull * cols * (rows_total - r);
        const bool right = (cols - 1 - c) * rows_total * 100ull < 3ull * cols * r;
        if (!(flags & 1u) && (left || right)) dn = 0;
        if (flags & 0x80000000u) dn = 0; // SARPRO_HIP_SYNTH_NO_BAND2 on the second band (decided by the launcher from `band`, not from the key's bits): no valid sample
        out[rl * pitch + c] = (uint16_t)dn;
    }
}

inline int stream_grid(uint64_t work_items, int block, int per_cu = 8) {
    const uint64_t want = (work_items + block - 1) / block;
    const uint64_t cap = 256ull * per_cu;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

} // namespace

hipError_t launch_dn_hist_u16(const DnHistArgs &a, int nrects, int nbands, bool vec, hipStream_t s) {
    if (nrects <= 0) return hipSuccess;
    const size_t lds = (size_t)a.lds_bins * sizeof(uint32_t);
    dim3 grid(nrects, nbands);
    if (vec) hipLaunchKernelGGL(k_dn_hist_u16<8>, grid, dim3(kBlock), lds, s, a);
    else hipLaunchKernelGGL(k_dn_hist_u16<1>, grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_dn_hist_u16_interior(const DnHistArgs &a, int nrects, int nbands, hipStream_t s) {
    if (nrects <= 0) return hipSuccess;
    const size_t lds = ((size_t)a.lds_bins + kWave) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_dn_hist_u16_interior, dim3(nrects, nbands), dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_dn_hist_u16_linear(const DnHistArgs &a, uint32_t rows, uint32_t cols, int nbands, hipStream_t s) {
    if (!rows || !cols) return hipSuccess;
    const size_t lds = ((size_t)a.lds_bins + kWave) * sizeof(uint32_t);
    // every workgroup ends by adding its occupied LDS bins onto ONE global histogram per band (hot words: ~88 adds per us and
    // address): below 32 K pixels per workgroup that flush, not the sweep, is the kernel (2048^2: 0.057 ms with 1024 workgroups)
    const uint64_t px = (uint64_t)rows * cols;
    const unsigned blocks = (unsigned)std::min<uint64_t>(256 * 4, std::max<uint64_t>(64, px / 32768));
    hipLaunchKernelGGL(k_dn_hist_u16_linear, dim3(blocks, nbands), dim3(kBlock), lds, s, a, rows, cols);
    return hipGetLastError();
}

hipError_t launch_sum_tile_hists(const SumTileHistArgs &a, int ntiles, int nbands, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_tile_hists, dim3(65536 / kBlock / 4, nbands), dim3(kBlock), 0, s, a, ntiles);
    return hipGetLastError();
}

hipError_t launch_tile_bin_hist(const TileBinHistArgs &a, int ntiles, int nbands, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_bin_hist, dim3(ntiles, nbands), dim3(kTileBinBlock), 0, s, a);
    return hipGetLastError();
}

size_t clahe_apply_lds_bytes(const ClaheApplyArgs &a, int nbands) {
    size_t win = 0;
    if (a.dev_state) win = a.lut_cap; // window read from device memory: capacity chosen by the caller
    else if (a.lut_in_lds)
        for (int b = 0; b < nbands; ++b) win = std::max<size_t>(win, a.win_hi[b] - a.win_lo[b] + 1);
    return (size_t)SARPRO_U16_CDF_COPIES * 256 * 4 * 8 + 256 * 4 + ((win + 15) & ~(size_t)15);
}

bool clahe_apply_spec_ok(const ClaheApplyArgs &, int) { return true; }

hipError_t launch_clahe_apply_u8_spec(ClaheApplyArgs a, int nrects, int nbands, hipStream_t s) {
    if (nrects <= 0) return hipSuccess;
    if (!a.dump) return hipErrorInvalidValue;
    if (a.dev_state) {
        // window only known on the device: capacity chosen by the caller (from the previous scene), global gather beyond it
        if (a.lut_cap < 256 || a.lut_cap > kSpecLutMaxEntries) a.lut_cap = kChainLutEntries;
    } else {
        uint32_t hi = 0;
        for (int b = 0; b < nbands; ++b) hi = std::max(hi, a.win_hi[b]);
        a.lut_cap = std::min<uint32_t>(hi + 1, kSpecLutMaxEntries);
    }
#ifndef SARPRO_LDS_PAD
#define SARPRO_LDS_PAD 0
#endif
    const size_t lds = SpecLds::lut + (((size_t)a.lut_cap) * 2 + 15 & ~(size_t)15) + SARPRO_LDS_PAD;
    if (a.hist_mode == 3) hipLaunchKernelGGL(k_clahe_apply_u8_spec<3>, dim3(nrects, nbands), dim3(kBlock), lds, s, a);
    else if (a.hist_mode == 2) hipLaunchKernelGGL(k_clahe_apply_u8_spec<2>, dim3(nrects, nbands), dim3(kBlock), lds, s, a);
    else if (a.hist_mode == 1) hipLaunchKernelGGL(k_clahe_apply_u8_spec<1>, dim3(nrects, nbands), dim3(kBlock), lds, s, a);
    else hipLaunchKernelGGL(k_clahe_apply_u8_spec<0>, dim3(nrects, nbands), dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_clahe_apply_u16(const ClaheApplyArgs &a, int nrects, int nbands, bool vec, bool out16,
                                  hipStream_t s) {
    if (nrects <= 0) return hipSuccess;
    const size_t lds = clahe_apply_lds_bytes(a, nbands);
    dim3 grid(nrects, nbands);
    if (vec) {
        if (out16) hipLaunchKernelGGL((k_clahe_apply_u16<8, true>), grid, dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_clahe_apply_u16<8, false>), grid, dim3(kBlock), lds, s, a);
    } else {
        if (out16) hipLaunchKernelGGL((k_clahe_apply_u16<1, true>), grid, dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_clahe_apply_u16<1, false>), grid, dim3(kBlock), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t opt_in_dynamic_lds(const void *kernel);

bool clahe_apply_u16_cf_supported(const ClaheApplyArgs &a, int nbands, size_t max_item_rows) {
    if (!a.dev_state || a.in_pitch % 8 != 0 || a.out_pitch % 8 != 0 || max_item_rows == 0 || max_item_rows > (size_t)kCfWaves * 64) return false;
    if (max_item_rows * std::max(a.in_pitch, a.out_pitch) * 2 >= (size_t)1 << 31) return false; // one buffer descriptor per item
    for (int b = 0; b < nbands; ++b)
        if ((reinterpret_cast<uintptr_t>(a.in[b]) & 15) || (reinterpret_cast<uintptr_t>(a.out[b]) & 15)) return false;
    return true;
}

hipError_t launch_clahe_apply_u16_cf(const ClaheApplyArgs &a, const int32_t *first, int nwg, int nbands, hipStream_t s) {
    if (nwg <= 0) return hipSuccess;
    hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_clahe_apply_u16_cf));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_clahe_apply_u16_cf, dim3(nwg * nbands), dim3(kCfBlock), CfLds::total, s, a, first, nbands); // workgroup i: band i % nbands, share i / nbands
    return hipGetLastError();
}

hipError_t launch_lut_apply_u16(const LutApplyArgs &a, bool vec, bool out16, hipStream_t s) {
    if (a.rows == 0 || a.cols == 0) return hipSuccess;
    const size_t esz = out16 ? 2 : 1;
    const size_t lds = a.dev_state ? (((size_t)a.lut_cap * esz + 15) & ~(size_t)15)
                                   : (a.lut_in_lds ? (((size_t)(a.win_hi - a.win_lo + 1) * esz + 15) & ~(size_t)15) : 0);
    const int V = vec ? 8 : 1;
    const uint64_t items = (uint64_t)a.rows * ((a.cols + V - 1) / V);
    dim3 grid(stream_grid(items, kBlock));
    if (vec) {
        if (out16) hipLaunchKernelGGL((k_lut_apply_u16<8, true>), grid, dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_lut_apply_u16<8, false>), grid, dim3(kBlock), lds, s, a);
    } else {
        if (out16) hipLaunchKernelGGL((k_lut_apply_u16<1, true>), grid, dim3(kBlock), lds, s, a);
        else hipLaunchKernelGGL((k_lut_apply_u16<1, false>), grid, dim3(kBlock), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t opt_in_dynamic_lds(const void *kernel) {
    static std::mutex m;
    static std::set<std::pair<int, const void *>> done; // the attribute is per device: the batch driver runs one thread per GPU
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(m);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

bool clahe_rgb_fused_supported(const ClaheRgbArgs &a) { // (nrects == 0: a rank whose stripe is empty -- nothing to launch, nothing to align)
    return a.nrects >= 0 && a.spec && a.dev_state && a.in_pitch % 8 == 0 && a.rgb_pitch_px % 16 == 0 &&
           (a.nrects == 0 || ((reinterpret_cast<uintptr_t>(a.in[0]) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.in[1]) & 15) == 0 &&
                              (reinterpret_cast<uintptr_t>(a.rgb) & 15) == 0));
}
hipError_t launch_spec_verdict(ChainSpecState *spec, const ChainBandState *state, hipStream_t s, int second) {
    hipLaunchKernelGGL(k_spec_verdict, dim3(1), dim3(1), 0, s, spec, state, second);
    return hipGetLastError();
}
hipError_t launch_clahe_rgb_fused(const ClaheRgbArgs &a, int grid, hipStream_t s) {
    if (!clahe_rgb_fused_supported(a) || grid <= 0) return hipErrorInvalidValue;
    if (a.nrects == 0) return hipSuccess;
    if (hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_clahe_rgb_fused))) return e;
    hipLaunchKernelGGL(k_clahe_rgb_fused, dim3(std::min(grid, a.nrects)), dim3(kRgbBlock), RgbLds::total, s, a);
    if (hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_clahe_rgb_fused_rescaled))) return e;
    hipLaunchKernelGGL(k_clahe_rgb_fused_rescaled, dim3(std::min(grid, a.nrects)), dim3(kRgbBlock), RgbLds::total, s, a);
    return hipGetLastError();
}
hipError_t launch_clahe_rgb_fused_retry(const ClaheRgbArgs &a, int grid, hipStream_t s) {
    if (!clahe_rgb_fused_supported(a) || grid <= 0 || !a.retry) return hipErrorInvalidValue;
    if (a.nrects == 0) return hipSuccess;
    if (hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_clahe_rgb_fused_retry))) return e;
    hipLaunchKernelGGL(k_clahe_rgb_fused_retry, dim3(std::min(grid, a.nrects)), dim3(kRgbBlock), RgbLds::total, s, a);
    return hipGetLastError();
}

// the in-process communicator's all-reduce (comm.cpp): out[i] = sum over the ranks' buffers
__global__ __launch_bounds__(kBlock) void k_sum_rank_buffers(const uint64_t *const *ptrs, int nranks, uint64_t *out, size_t count) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock) {
        uint64_t acc = 0;
        for (int r = 0; r < nranks; ++r) acc += ptrs[r][i];
        out[i] = acc;
    }
}
hipError_t launch_sum_rank_buffers(const uint64_t *const *d_ptrs, int nranks, uint64_t *out, size_t count, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(k_sum_rank_buffers, dim3((unsigned)std::min<size_t>((count + kBlock - 1) / kBlock, 1024)), dim3(kBlock), 0, s, d_ptrs, nranks, out, count);
    return hipGetLastError();
}

hipError_t launch_compose_u8(const ComposeArgs &a, int vec, hipStream_t s) {
    if (a.rows == 0 || a.cols == 0) return hipSuccess;
    if (a.speculative && (vec != 16 || !a.spec)) return hipErrorInvalidValue;
    const void *fn = a.speculative ? reinterpret_cast<const void *>(k_compose_u8<16, true>)
                     : vec == 16   ? reinterpret_cast<const void *>(k_compose_u8<16, false>) : reinterpret_cast<const void *>(k_compose_u8<1, false>);
    if (hipError_t e = opt_in_dynamic_lds(fn)) return e; // ~115 KiB
    const uint64_t items = (uint64_t)a.rows * ((a.cols + vec - 1) / vec);
    dim3 grid(stream_grid(items, kComposeBlock, 2));
    if (a.speculative) hipLaunchKernelGGL((k_compose_u8<16, true>), grid, dim3(kComposeBlock), kComposeTableBytes + kComposeStageBytes, s, a);
    else if (vec == 16) hipLaunchKernelGGL((k_compose_u8<16, false>), grid, dim3(kComposeBlock), kComposeTableBytes + kComposeStageBytes, s, a);
    else hipLaunchKernelGGL((k_compose_u8<1, false>), grid, dim3(kComposeBlock), kComposeTableBytes + kComposeStageBytes, s, a);
    return hipGetLastError();
}

bool lut_compose_fits(const LutComposeArgs &a) {
    if (a.dev_state) return true; // capacity-limited inside the kernel
    return (size_t)kComposeTableBytes + kComposeStageBytes + ((a.win_hi[0] + 16) & ~15u) + ((a.win_hi[1] + 16) & ~15u) <= 160 * 1024;
}

hipError_t launch_lut_compose_u16(const LutComposeArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.cols == 0) return hipSuccess;
    const size_t lds = a.dev_state ? (size_t)kComposeTableBytes + kComposeStageBytes + 2 * (size_t)a.lut_cap
                                   : (size_t)kComposeTableBytes + kComposeStageBytes + ((a.win_hi[0] + 16) & ~15u) + ((a.win_hi[1] + 16) & ~15u);
    if (hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_lut_compose_u16))) return e;
    const uint64_t items = (uint64_t)a.rows * ((a.cols + 15) / 16);
    hipLaunchKernelGGL(k_lut_compose_u16, dim3(stream_grid(items, kComposeBlock, 1)), dim3(kComposeBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_polop_f32(int op, const float *a, const float *b, size_t n, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                         reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    hipLaunchKernelGGL(k_polop_f32, dim3(stream_grid(n / 4 + 1, kBlock)), dim3(kBlock), 0, s, op, a, b, n, out, vec_ok);
    return hipGetLastError();
}

hipError_t launch_remap_u8(uint8_t *buf, size_t pitch, uint32_t rows, uint32_t cols, const uint8_t *d_map256,
                           hipStream_t s) {
    if (rows == 0 || cols == 0) return hipSuccess;
    hipLaunchKernelGGL(k_remap_u8, dim3(stream_grid((uint64_t)rows * cols, kBlock)), dim3(kBlock), 0, s, buf, pitch, rows,
                       cols, d_map256);
    return hipGetLastError();
}

hipError_t launch_hist256_u8(const uint8_t *in, size_t pitch, uint32_t rows, uint32_t cols,
                             unsigned long long *hist, hipStream_t s) {
    if (rows == 0 || cols == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hist256_u8, dim3(stream_grid((uint64_t)rows * cols, kBlock)), dim3(kBlock), 0, s, in, pitch,
                       rows, cols, hist);
    return hipGetLastError();
}

hipError_t launch_synth_scene_u16(uint64_t seed, int band, const uint16_t *d_q, size_t rows_total, size_t cols,
                                  size_t row0, size_t rows_local, uint16_t *d_out, size_t pitch, uint32_t flags, hipStream_t s) {
    if (rows_local == 0 || cols == 0) return hipSuccess;
    const uint64_t key = seed ^ ((uint64_t)band << 60);
    const uint64_t per_side = ((flags >> 8) & 255u) ? ((flags >> 8) & 255u) : 16u;
    const uint64_t block = std::max<uint64_t>((rows_total + per_side - 1) / per_side, 1);
    hipLaunchKernelGGL(k_synth_scene_u16, dim3(stream_grid((uint64_t)rows_local * cols, kBlock)), dim3(kBlock), 0, s,
                       key, d_q, (uint64_t)rows_total, (uint64_t)cols, (uint64_t)row0, (uint64_t)rows_local, block,
                       d_out, pitch, (flags & 0x7FFFFFFFu) | (((flags & 4u) && band == 1) ? 0x80000000u : 0u));
    return hipGetLastError();
}

} // namespace sarpro

#ifdef SARPRO_RGB_WG_TIMES
extern "C" int sarpro_hip_debug_rgb_wg_times(unsigned long long *out /* [1024][2] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sarpro::g_rgb_wg_times), sizeof(unsigned long long) * 8192) == hipSuccess ? 0 : -1;
}
#endif
#ifdef SARPRO_SPEC_MEASURE
// instrumented build only: reads and clears the largest speculation errors seen so far, per margin class (interior, dy < 0, dx < 0, corner)
extern "C" int sarpro_hip_debug_spec_max_err(float out[4]) {
    uint32_t h[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(sarpro::g_spec_max_err), sizeof(h)) != hipSuccess) return -1;
    const uint32_t z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(sarpro::g_spec_max_err), z, sizeof(z)) != hipSuccess) return -1;
    for (int i = 0; i < 4; ++i) { float f; __builtin_memcpy(&f, &h[i], 4); out[i] = f; }
    return 0;
}
#endif

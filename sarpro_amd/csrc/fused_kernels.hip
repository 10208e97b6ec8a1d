// fused_kernels.hip -- the headline pass on gfx950: CLAHE blend of BOTH bands + suppressed synRGB composition
// in ONE sweep.  4 B/px read (two u16 DN rasters) + 3 B/px written (interleaved RGB) = the 7 B/px of algorithmic
// traffic of save.rs:317-367 at native resolution; no u8 level raster ever exists.
//
// What stands between the blend and the composition in the reference is a global barrier: the u8 rescale of each
// band needs the min / max of its level raster (autoscale.rs:348-364) and the suppressed synRGB needs the floor of the
// combined level histogram (synthetic_rgb.rs:99-113).  This pass does not wait for them, it PREDICTS them and proves
// the prediction while it runs:
//   * rescale = identity  <=>  level 0 and level 255 both occur.  k_fused_prep establishes both from tables alone:
//     an invalid pixel (DN = 0) is level 0; a valid pixel of the top CLAHE bin inside a tile that only blends
//     non-empty tiles with weights in [0,1] is exactly 255 (all four CDFs are 1.0 and fl(fl(1-d) + d) = 1.0).
//     Scenes without either run the exact passes below instead (spec_ok = 0).
//   * floor F: predicted from a row sample (sample pass, every sample_stride-th step), then VERIFIED exactly: the
//     pass counts the band-pixels with level < F and < F + 1 (SWAR on the packed level bytes, no LDS atomics);
//     F is the reference's floor iff  cum(F-1) < target <= cum(F).
//   When the verification refutes the prediction (binomial noise of the sample, ~2 % of scenes) the scene takes the
//   exact passes: histogram pass (full per-band level histograms, no RGB) -> k_chain_finish -> final pass.  All of
//   them are enqueued unconditionally and gated by FusedState in device memory: the host never synchronises.
//
// Per pixel and band:  DN -> min(DN, win_hi) -> ONE 16-B LDS gather from a DN-indexed table of
// (A, B, C, D) = 255 (c00 | c01 - c00 | c10 - c00 | c11 - c10 - c01 + c00) -> y = (A + dx B) + dy (C + dx D) in three
// f32 FMAs -> v_cvt_pk_u8_f32 of y - 0.5 -/+ delta.  Where the two bytes agree they are floor(y) = the reference's
// level (error bound below); the ~0.2 % of pixels where they differ are QUEUED and recomputed with the reference's
// exact f64 sequence by k_fused_fixup, which patches their RGB bytes and corrects the counts -- the exact path is
// not in the row loop and its f64 tables are not in LDS.  Composition: R2[l1], G2[l2], B2[l1 << 8 | l2] from LDS
// (64.5 KiB), 12 B per lane stored as one dwordx3: a wave instruction writes 768 contiguous bytes.
//
// Error bound of the f32 value against the real y = A + dx B + dy (C + dx D) = 255 o - 0.5 - delta (u = 2^-24; the reference's own
// f64 rounding is ~1e-12).  Table entries are f64 values rounded once to f32, dx likewise (relative error <= u each); dy is
// rounded and then carries a flag in its last mantissa bit (below): relative error <= 3u.
//   interior cell (dx, dy in [0,1): |A| <= 255.5 + delta, |dx B| <= 255, |p| <= 255.5, |C| <= 255, |dx D| <= 510, |q| <= 255):
//     p = fl(dx B + A):   |dp| <= u |A| + 2u |dx B| + u |p|                         <= 1021 u
//     q = fl(dx D + C):   |dq| <= u |C| + 2u |dx D| + u |q|                         <= 1530 u
//     ya = fl(dy q + p):  |dya| <= |dy| (|dq| + 3u |q|) + |dp| + u |ya|              <= 3572 u,   yb = fl(ya + 2 delta): + 256 u
//     total 3828 u = 2.28e-4  <  kDeltaInner = 2^-12 = 2.44e-4
//   extrapolating cell (dx or dy in [-0.5, 0): |p| <= 510.5, |q| <= 510, |ya| <= 1020.5):
//     |dp| <= 1276 u, |dq| <= 1785 u, |dya| <= (1785 + 3 * 510) u + 1276 u + 1020.5 u = 5612 u, + 1021 u for yb
//     total 6633 u = 3.95e-4  <  kDeltaEdge = 2^-11 = 4.88e-4
// Both conversions giving the same byte n therefore means  n + (delta - E) <= 255 o <= n + 1 - (delta - E):  n is the level.
//
// Saturated bins in an extrapolating cell.  A pixel of the top CLAHE bin blends four CDFs that are exactly 1.0; in an interior
// cell the result is exactly 1.0 (fl(fl(1 - d) + d) = 1 for d in [0,1)) and the table entry says so.  With d < 0 the same
// expression can round to 1 - 2^-53, i.e. level 254, depending on the bits of the weight -- the speculative test can never
// settle such a pixel and bright regions would flood the queue (8 % of their pixels).  The level only depends on the
// position: T(c) = fl(fl(1 - dx) + dx) per column and, where T = 1.0, fl(fl(1 - dy) + dy) per row.  The pass keeps the
// column bits (T == 1.0) in a register and finds the row's bit in the last mantissa bit of its f32 row weight (set by the
// host: 1 = level 255), and settles these pixels itself; columns with T != 1.0 still go to the queue.
#include "fused_kernels.h"

#include <algorithm>
#include <type_traits>

namespace sarpro {

namespace {

constexpr int kFBlock = 1024, kFWaves = 16;
constexpr float kDeltaInner = 1.0f / 4096.0f, kDeltaEdge = 1.0f / 2048.0f;

// LDS map (bytes).  R2/G2, region A and B2 sit below 64 KiB so that their bases fit the 16-bit immediate offset of
// a ds_read: the per-pixel addresses are then the bare level / index / entry number.
constexpr uint32_t kLdsRG = 0;                    // R2[256] | G2[256]
constexpr uint32_t kLdsRegA = 512;                // DN-indexed entries, region A
constexpr uint32_t kLdsB2 = 49152;                // B2[65536]
constexpr uint32_t kLdsMisc = kLdsB2 + 65536;     // queue counter
constexpr uint32_t kLdsHist = kLdsMisc + 64;      // [2][256] u32 (sample / histogram pass)
constexpr uint32_t kLdsTmp = kLdsHist + 2048;     // bin-indexed entries [2][257] (staging; the table itself in two-level mode)
constexpr uint32_t kLdsRegB = kLdsTmp + 2 * 257 * 16 + 32; // DN-indexed entries, region B
#ifdef FUSED_ABL_LDS80
constexpr uint32_t kLdsTotal = 163840; constexpr uint32_t kLdsLaunch = 81920;
#else
constexpr uint32_t kLdsTotal = 163840; constexpr uint32_t kLdsLaunch = kLdsTotal;
#endif
constexpr uint32_t kCapA = (kLdsB2 - kLdsRegA) / 16, kCapB = (kLdsTotal - kLdsRegB) / 16;
static_assert(kLdsRegB % 16 == 0 && kLdsTmp % 16 == 0, "entry alignment");

#define LDS_AT(T, off) (*reinterpret_cast<__attribute__((address_space(3))) T *>((uint32_t)(off)))

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// LDS atomics through the address-space-3 pointer (the HIP overloads take generic pointers)
#define LDS_ADD(off, v) __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uint32_t)(off)), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

__device__ __forceinline__ int f_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int f_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ float sgpr_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

__device__ __forceinline__ bool pass_runs(const FusedState *fs, int mode) {
    const bool spec = fs->spec_ok != 0;
    return (mode == kFusedSample || mode == kFusedSpec) ? spec : (!spec || fs->verdict != 0);
}

// The reference's level of one sample (autoscale.rs:307-330, 600-606 at max_val 255), operation for operation.
__device__ __forceinline__ uint32_t exact_level(const double *__restrict__ cdfs, const uint8_t *__restrict__ binlut, uint32_t win_hi,
                                                uint32_t dn, const int32_t id[4], double dx, double dy) {
    if (!dn) return 0u;
    const uint32_t bin = binlut[min(dn, win_hi)];
    const double c00 = cdfs[(size_t)id[0] * 256 + bin], c01 = cdfs[(size_t)id[1] * 256 + bin];
    const double c10 = cdfs[(size_t)id[2] * 256 + bin], c11 = cdfs[(size_t)id[3] * 256 + bin];
    const double top = c00 * (1.0 - dx) + c01 * dx;
    const double bottom = c10 * (1.0 - dx) + c11 * dx;
    double o = top * (1.0 - dy) + bottom * dy;
    o = fmin(fmax(o, 0.0), 1.0);
    return (uint32_t)(o * 255.0);
}

// ------------------------------------------------------------------------------------
// Preconditions of the speculative pass + reset of the per-scene state.  One block.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fused_prep(FusedPrepArgs a) {
    __shared__ int bad;
    const int t = threadIdx.x;
    FusedState *fs = a.fs;
    if (t == 0) bad = 0;
    if (t < 32) fs->cum_est[t] = 0.0;
    for (int i = t; i < 4 * kFusedMaxGrid; i += 256) (&fs->qcount[0][0])[i] = 0u;
    __syncthreads();
    // every tile non-empty (its CDF ends at exactly 1.0) and, per band, a top-bin pixel in a tile of tile row >= 1 and
    // tile column >= 1 (those only lie in cells with weights in [0,1]): level 255 occurs.  An invalid pixel: level 0 occurs.
    int has_top[2] = {0, 0};
    for (int i = t; i < 2 * 64; i += 256) {
        const int b = i >> 6, tile = i & 63;
        if (a.cdfs[((size_t)b * 64 + tile) * 256 + 255] != 1.0) bad = 1;
        if ((tile >> 3) >= 1 && (tile & 7) >= 1 && a.tile_bins[((size_t)b * 64 + tile) * 256 + 255] != 0ull) has_top[b] = 1;
    }
    __shared__ int top[2];
    if (t < 2) top[t] = 0;
    __syncthreads();
    if (has_top[0]) top[0] = 1;
    if (has_top[1]) top[1] = 1;
    __syncthreads();
    if (t == 0) {
        bool ok = !bad && top[0] && top[1];
        for (int b = 0; b < 2; ++b) ok = ok && a.state[b].stats.valid_count < a.total_px; // an invalid pixel exists
        ok = ok && a.total_px < 0x7FFFFFFFull;     // the reference's u32 histogram counters do not saturate
        if (a.force & kFusedForceNoSpec) ok = false;
        const uint32_t w0 = a.state[0].win_hi + 1, w1 = a.state[1].win_hi + 1; // entries 0 .. win_hi
        uint32_t direct = 1, k0 = kLdsRegA / 16, k1 = kLdsRegB / 16;
        if (w0 <= kCapA && w1 <= kCapB) { k0 = kLdsRegA / 16; k1 = kLdsRegB / 16; }
        else if (w1 <= kCapA && w0 <= kCapB) { k0 = kLdsRegB / 16; k1 = kLdsRegA / 16; }
        else direct = 0;
        if (a.force & kFusedForceTwoLevel) direct = 0;
        fs->spec_ok = ok ? 1u : 0u;
        fs->direct = direct;
        fs->k_base[0] = k0; fs->k_base[1] = k1;
        fs->verdict = 0u;
        fs->floor_pred = 0;
        for (int m = 0; m < 4; ++m) fs->fix_done[m] = 0u;
        fs->n_lt[0] = 0ull; fs->n_lt[1] = 0ull;
        for (int k = 0; k < 8; ++k) fs->dbg[k] = 0ull;
        fs->dbg_n = 0u;
        fs->unsampled = 0.0; fs->predict_done = 0u; fs->total_px = a.total_px;
    }
}

// ------------------------------------------------------------------------------------
// Predicted floor from the sample, post-stratified by (band, tile, CLAHE bin): the EXACT number of valid pixels of every
// stratum is known (tile_bins), the sample only supplies the share of a stratum's pixels at each low level.  Most strata
// are pure (all their pixels on one side of a level boundary) and contribute no sampling noise at all; what is left is
// the binomial noise of the few strata whose level range straddles the boundary -- several times less than that of the
// plain sample histogram.  One block per (tile, band), one thread per bin; the last block to finish decides the floor.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fused_predict(FusedPredictArgs a) {
    FusedState *fs = a.fs;
    if (!fs->spec_ok) return;
    __shared__ float wgt[256];        // stratum weight: exact pixels / sampled pixels
    __shared__ uint32_t cumh[256][33]; // sampled pixels of the stratum at levels <= l (padded: no bank conflicts over bins)
    __shared__ double part[32];
    const int tile = blockIdx.x, band = blockIdx.y, t = threadIdx.x;
    const size_t stratum = ((size_t)band * 64 + tile) * 256 + t;
    const double n = (double)a.tile_bins[stratum];
    uint32_t *h = a.hist3 + stratum * 32;
    uint32_t run = 0;
    uint4 hq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) hq[q] = reinterpret_cast<const uint4 *>(h)[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        run += hq[q].x; cumh[t][4 * q] = run; run += hq[q].y; cumh[t][4 * q + 1] = run;
        run += hq[q].z; cumh[t][4 * q + 2] = run; run += hq[q].w; cumh[t][4 * q + 3] = run;
    }
    if (run) { // last reader: the next scene's sample pass starts on zeros
#pragma unroll
        for (int q = 0; q < 8; ++q) reinterpret_cast<uint4 *>(h)[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    wgt[t] = run ? (float)(n / (double)run) : 0.0f;
    if (t < 32) part[t] = 0.0;
    __syncthreads();
    // thread (l, g): level l < 31, bins g, g + 8, ...
    {
        const int l = t & 31, g = t >> 5;
        double acc = 0.0;
        if (l < 31)
            for (int b = g; b < 256; b += 8) acc += (double)wgt[b] * (double)cumh[b][l];
        if (l < 31 && acc != 0.0) atomicAdd(&part[l], acc);
    }
    __syncthreads();
    if (t < 31 && part[t] != 0.0) atomicAdd(&fs->cum_est[t], part[t]);
    if (!run && n > 0.0) atomicAdd(&fs->unsampled, n);
    __threadfence();
    __shared__ uint32_t ticket;
    __syncthreads();
    if (t == 0) ticket = atomicAdd(&fs->predict_done, 1u);
    __syncthreads();
    if (ticket != gridDim.x * gridDim.y - 1 || t != 0) return;
    __threadfence();
    // invalid pixels are level 0 in both bands; target as the verification computes it (synthetic_rgb.rs:99-101)
    double inv = 0.0;
    for (int b = 0; b < 2; ++b) inv += (double)(a.total_px - a.state[b].stats.valid_count);
    const uint32_t total = (uint32_t)(a.total_px + a.total_px);
    const double target = round((double)total * 0.05);
    int f = -1;
    for (int l = 0; l < 31; ++l) {
        const double c = atomicAdd(&fs->cum_est[l], 0.0) + inv;
        if (c >= target) { f = l; break; }
    }
    if (f < 0) { fs->spec_ok = 0u; f = 0; } // the floor lies beyond the levels the sample resolves: exact passes
    if (a.force & kFusedForceMispredict) f += 1;
    fs->floor_pred = f;
}

// ------------------------------------------------------------------------------------
// Predicted floor (from the sample histogram) and the compose tables it implies: identity rescale,
// suppressed variant (synthetic_rgb.rs:115-156 through the host-built powf tables).  64 blocks.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fused_tables_predict(FusedTablesArgs a) {
    FusedState *fs = a.fs;
    if (!fs->spec_ok) return;
    const int t = threadIdx.x;
    const int f = fs->floor_pred; // k_fused_predict
    const int fwc = f + 3 < 40 ? f + 3 : 40;
    const uint8_t *lut_r = a.supp_rg + (size_t)fwc * 512, *lut_g = lut_r + 256;
    uint8_t *R2 = a.tables, *G2 = a.tables + 256, *B2 = a.tables + 512;
    if (blockIdx.x == 0) {
        R2[t] = t <= fwc ? 0 : lut_r[t];
        G2[t] = t <= fwc ? 0 : lut_g[t];
    }
    for (int i = blockIdx.x * 256 + t; i < 65536; i += gridDim.x * 256) {
        const int v1 = i >> 8, v2 = i & 255;
        const bool water = v1 <= fwc && v2 <= fwc; // synthetic_rgb.rs:161-166
        B2[i] = water ? 0 : a.blue_pair_supp[((size_t)lut_r[v1] << 8) | lut_g[v2]];
    }
}

// ------------------------------------------------------------------------------------
// Table staging for one interpolation cell (all 1024 threads; callers put barriers around it).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ v4f make_entry(double c00, double c01, double c10, double c11, bool neg, float delta) {
    const bool sat = c00 == 1.0 && c01 == 1.0 && c10 == 1.0 && c11 == 1.0 && !neg; // exactly 255 (see kernels.hip, speculative apply)
    const bool zero = c00 == 0.0 && c01 == 0.0 && c10 == 0.0 && c11 == 0.0;       // exactly 0 whatever the weights
    const double bias = -0.5 - (double)delta;
    v4f e;
    if (sat) { e.x = (float)(255.25 + bias); e.y = 0.0f; e.z = 0.0f; e.w = 0.0f; }
    else if (zero) { e.x = (float)(0.25 + bias); e.y = 0.0f; e.z = 0.0f; e.w = 0.0f; }
    else {
        e.x = (float)(255.0 * c00 + bias);
        e.y = (float)(255.0 * (c01 - c00));
        e.z = (float)(255.0 * (c10 - c00));
        e.w = (float)(255.0 * ((c11 - c10) - (c01 - c00)));
    }
    return e;
}

__device__ __forceinline__ void stage_cell(const FusedArgs &a, const FusedItem &I, bool direct, const uint32_t kb[2], const uint32_t hi[2]) {
    const bool neg = (I.flags & 1) != 0;
    const float delta = neg ? kDeltaEdge : kDeltaInner;
    for (int i = threadIdx.x; i < 2 * 257; i += kFBlock) {
        const int b = i / 257, bin = i - b * 257;
        double c[4] = {0.0, 0.0, 0.0, 0.0}; // entry 256: invalid pixels
        if (bin < 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = a.cdfs[b][(size_t)I.id[k] * 256 + bin];
        }
        LDS_AT(v4f, kLdsTmp + (uint32_t)i * 16u) = make_entry(c[0], c[1], c[2], c[3], neg, delta);
        // the bin of every DN >= win_hi is saturated in all four tiles (the precondition of settling those pixels by position in an
        // extrapolating cell; not a given: a degenerate window can put them in any bin)
        if (bin == (int)a.binlut[b][hi[b]]) LDS_AT(uint32_t, kLdsMisc + 16u + 4u * (uint32_t)b) = (c[0] == 1.0 && c[1] == 1.0 && c[2] == 1.0 && c[3] == 1.0) ? 1u : 0u;
    }
    if (!direct) return;
    __syncthreads();
    for (int b = 0; b < 2; ++b) {
        const uint8_t *__restrict__ glut = a.binlut[b];
        for (uint32_t e = threadIdx.x; e <= hi[b]; e += kFBlock) {
            const uint32_t bin = e ? (uint32_t)glut[e] : 256u;
            LDS_AT(v4f, (kb[b] + e) * 16u) = LDS_AT(v4f, kLdsTmp + ((uint32_t)b * 257u + bin) * 16u);
        }
    }
}

// ------------------------------------------------------------------------------------
// The rows of one piece.  One step of a wave = one row x 256 columns (4 pixels per lane, both bands).
// Software pipeline inside the step loop (the LDS gathers are ~60 % bank conflicts: their latency is what a wave waits for):
//   band-1 entries of step s were gathered during step s-1;  step s:  issue the band-2 gathers -> blend band 1 -> issue the
//   band-1 gathers of step s+1 -> blend band 2 -> (queue) -> issue the 12 compose lookups -> count under them -> pack, store.
// The DN of four steps are in flight in four register sets (the loop is unrolled four times; a rotation through register
// moves would make every move wait for its load); rows past the piece are clamped, so every load is unconditional and
// hipcc's waitcnt pass counts them.  (Loads written as inline asm with hand-counted waits are NOT an option: the compiler
// copies and reuses registers whose loads are still in flight.)
// ------------------------------------------------------------------------------------
template <int MODE, bool DIRECT>
__device__ __forceinline__ void fused_rows(const FusedArgs &a, const FusedItem &I, const uint32_t kb[2], const uint32_t hi[2], uint32_t qcap,
                                           uint32_t thrK0, uint32_t thrK1, uint32_t &acc0, uint32_t &acc1, uint32_t &nsteps) {
    constexpr bool kCompose = MODE == kFusedSpec || MODE == kFusedFinal;
    constexpr bool kQueue = MODE != kFusedSample;
    const int gx = 1 << I.gx_log2, gy = kFWaves >> I.gx_log2;
    const int wave = f_wave();
    const int wx = wave & (gx - 1), wy = wave >> I.gx_log2;
    const int lane = f_lane();
    const int col = I.cstart + (wx * 64 + lane) * 4;
    if (col >= I.c1 || col + 4 <= I.c0) return; // lane outside the piece (ragged last wave column)
    const bool full = col >= I.c0 && col + 4 <= I.c1;
    // bytes of the packed levels that belong to pixels this lane owns (pixel j: bytes 2 (j & 1), 2 (j & 1) + 1 of dword j >> 1)
    uint32_t keep[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (col + j >= I.c0 && col + j < I.c1) keep[j >> 1] |= 0xFFFFu << (16 * (j & 1));
    float dxf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = min(max(col + j, I.c0), I.c1 - 1);
        dxf[j] = a.col_wf[c];
    }
    const bool neg = (I.flags & 1) != 0;
    const float two_delta = 2.0f * (neg ? kDeltaEdge : kDeltaInner);
    uint32_t colone = 0u; // extrapolating cell: pixel j's column has T(c) == 1.0 (see the header)
    const bool sat1 = LDS_AT(uint32_t, kLdsMisc + 16u) != 0u, sat2 = LDS_AT(uint32_t, kLdsMisc + 20u) != 0u; // staged with the cell's tables
    if (neg && kQueue) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double dx = a.col_w[min(max(col + j, I.c0), I.c1 - 1)].d;
            if ((1.0 - dx) + dx == 1.0) colone |= 1u << j;
        }
    }
    const uint32_t hi2[2] = {hi[0] | (hi[0] << 16), hi[1] | (hi[1] << 16)};
    const uint32_t kb2[2] = {kb[0] | (kb[0] << 16), kb[1] | (kb[1] << 16)};
    const float *__restrict__ row_wf = a.row_wf + a.row_off;
    uint4 *__restrict__ queue = a.queue + a.qoff[blockIdx.x];
    const uint16_t *__restrict__ p1 = a.in[0] + col, *__restrict__ p2 = a.in[1] + col;

    // entry addresses of one band's four samples, then the four 16-B gathers
    auto gather = [&](int b, const uint2 w, v4f e[4], uint32_t addr[4]) {
        const uint32_t ww[2] = {w.x, w.y};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (DIRECT) {
                // both samples of the dword: clamp to the window, add the band's entry base, one SDWA shift each -> byte offset
                const v2us cl = __builtin_elementwise_min(__builtin_bit_cast(v2us, ww[h]), __builtin_bit_cast(v2us, hi2[b]));
                const uint32_t cw = __builtin_bit_cast(uint32_t, (v2us)(cl + __builtin_bit_cast(v2us, kb2[b])));
                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(addr[2 * h]) : "v"(4u), "v"(cw));
                asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(addr[2 * h + 1]) : "v"(4u), "v"(cw));
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint32_t d = k ? (ww[h] >> 16) : (ww[h] & 0xFFFFu);
                    const uint32_t bin = d ? (uint32_t)a.binlut[b][min(d, hi[b])] : 256u;
                    addr[2 * h + k] = kLdsTmp + ((uint32_t)b * 257u + bin) * 16u;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef FUSED_ABL_NOGATHER
            e[j] = (v4f){__builtin_bit_cast(float, addr[j]), 1.0f, 2.0f, 3.0f};
#else
            e[j] = LDS_AT(v4f, addr[j]);
#endif
        }
    };
    // y = (A + dx B) + dy (C + dx D) for one band's four samples; both conversions into the packed index bytes
    auto blend = [&](int b, const v4f e[4], float dyf, uint32_t Da[2], uint32_t Db[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = fmaf(e[j].y, dxf[j], e[j].x);
            const float q = fmaf(e[j].w, dxf[j], e[j].z);
            const float ya = fmaf(q, dyf, p);
            const float yb = ya + two_delta;
            const int pos = 2 * (j & 1) + (b ? 0 : 1); // band 1 -> high byte of the pixel's index, band 2 -> low byte
            Da[j >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(ya, pos, Da[j >> 1]);
            Db[j >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(yb, pos, Db[j >> 1]);
        }
    };

    // one step; e1 holds the band-1 entries of THIS step on entry and those of the NEXT step (DN w1n) on return
    auto process = [&](auto allfull_t, int r, const uint2 w1, const uint2 w2, const float dyf, const uint2 w1n, v4f e1[4], uint32_t addr1[4]) {
        constexpr bool ALLFULL = decltype(allfull_t)::value;
        v4f e2[4];
        uint32_t addr2[4], cur1[4];
        gather(1, w2, e2, addr2);
        uint32_t Da[2] = {0u, 0u}, Db[2] = {0u, 0u};
        blend(0, e1, dyf, Da, Db);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur1[j] = addr1[j];
        gather(0, w1n, e1, addr1);
        blend(1, e2, dyf, Da, Db);
        uint32_t cnt[2] = {Da[0], Da[1]}; // the level bytes as counted: pixels this lane queues are forced to 0xFF below
        uint32_t skipq[2] = {0u, 0u};
#ifdef FUSED_ABL_NOQUEUE
        if (false) {
#else
        if (kQueue) {
#endif
            const uint32_t d0 = (Da[0] ^ Db[0]) & keep[0], d1 = (Da[1] ^ Db[1]) & keep[1];
            if (d0 | d1) { // rare: some level of this lane lies within the margin of an integer
                const uint32_t rowlvl = 254u + (__builtin_bit_cast(uint32_t, dyf) & 1u); // the row's saturated level (header)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t sh = 16 * (j & 1);
                    uint32_t dd = ((j < 2 ? d0 : d1) >> sh) & 0xFFFFu;
                    if (!dd) continue;
                    const uint32_t dn1 = ((j < 2 ? w1.x : w1.y) >> sh) & 0xFFFFu, dn2 = ((j < 2 ? w2.x : w2.y) >> sh) & 0xFFFFu;
                    if (neg && ((colone >> j) & 1u)) { // top-bin pixel of an extrapolating cell: its level is the row's
                        if ((dd & 0xFF00u) && sat1 && dn1 >= hi[0]) { Da[j >> 1] = (Da[j >> 1] & ~(0xFF00u << sh)) | (rowlvl << (sh + 8)); dd &= 0x00FFu; }
                        if ((dd & 0x00FFu) && sat2 && dn2 >= hi[1]) { Da[j >> 1] = (Da[j >> 1] & ~(0x00FFu << sh)) | (rowlvl << sh); dd &= 0xFF00u; }
                        if (!dd) continue;
                    }
                    // queue the pixel for k_fused_fixup; a full queue drops the entry: the workgroup's share is then redone exactly
                    skipq[j >> 1] |= 0xFFFFu << sh;
                    const uint32_t slot = LDS_ADD(kLdsMisc, 1u);
                    if (slot < qcap) queue[slot] = make_uint4((uint32_t)r, (uint32_t)(col + j), dn1 | (dn2 << 16), 0u);
                }
                cnt[0] = Da[0]; cnt[1] = Da[1];
                cnt[0] |= skipq[0]; cnt[1] |= skipq[1];
            }
        }
        uint32_t R[4], G[4], B[4];
        if (kCompose) { // the 12 table lookups are issued before the counting so that it runs under their latency
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ix = (j & 1) ? (Da[j >> 1] >> 16) : (Da[j >> 1] & 0xFFFFu);
#ifdef FUSED_ABL_NOCOMPOSE
                R[j] = ix >> 8; G[j] = ix & 0xFFu; B[j] = ix >> 4;
#else
                R[j] = LDS_AT(uint8_t, kLdsRG + (ix >> 8));
                G[j] = LDS_AT(uint8_t, kLdsRG + 256u + (ix & 0xFFu));
                B[j] = LDS_AT(uint8_t, kLdsB2 + ix);
#endif
            }
        }
#ifndef FUSED_ABL_NOCOUNT
        if (MODE == kFusedSpec) {
            // band-pixels below the two thresholds, counted byte-parallel: bit 7 of ((l & 0x7F) + 0x80 - t) | l  <=>  l >= t (t < 128);
            // bytes of pixels the lane does not own or has queued are forced to 0xFF (never below)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t x = ALLFULL ? cnt[h] : (cnt[h] | ~keep[h]);
                const uint32_t xl = x & 0x7F7F7F7Fu, xh = x & 0x80808080u;
                acc0 += (uint32_t)__builtin_popcount(((xl + thrK0) & 0x80808080u) | xh);
                acc1 += (uint32_t)__builtin_popcount(((xl + thrK1) & 0x80808080u) | xh);
            }
            nsteps += 1;
        }
#endif
        if (MODE == kFusedSample) {
            // sampled level counts per (band, CLAHE bin) of this piece's tile, levels >= 31 lumped (k_fused_predict)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t sh = 16 * (j & 1);
                    const uint32_t bin = ((b ? addr2[j] : cur1[j]) - kLdsTmp) / 16u - (uint32_t)b * 257u; // two-level addressing: the entry IS the bin
                    if (((keep[j >> 1] >> sh) & 1u) && bin < 256u) {
                        const uint32_t l = (Da[j >> 1] >> (sh + (b ? 0 : 8))) & 0xFFu;
                        LDS_ADD(kLdsB2 + (((uint32_t)b * 256u + bin) * 32u + min(l, 31u)) * 4u, 1u);
                    }
                }
        }
        if (MODE == kFusedHist) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t sh = 16 * (j & 1);
                if (((keep[j >> 1] & ~skipq[j >> 1]) >> sh) & 1u) {
                    const uint32_t l1 = (Da[j >> 1] >> (sh + 8)) & 0xFFu, l2 = (Da[j >> 1] >> sh) & 0xFFu;
                    if (l1) LDS_ADD(kLdsHist + l1 * 4u, 1u);
                    if (l2) LDS_ADD(kLdsHist + 1024u + l2 * 4u, 1u);
                }
            }
        }
        if (kCompose) {
            const uint32_t o0 = R[0] | (G[0] << 8) | (B[0] << 16) | (R[1] << 24);
            const uint32_t o1 = G[1] | (B[1] << 8) | (R[2] << 16) | (G[2] << 24);
            const uint32_t o2 = B[2] | (R[3] << 8) | (G[3] << 16) | (B[3] << 24);
            uint8_t *o = a.rgb + ((size_t)r * a.rgb_pitch_px + (size_t)col) * 3;
            struct __attribute__((packed, aligned(4))) U3 { uint32_t x, y, z; };
            if (ALLFULL) {
#ifdef FUSED_ABL_NOSTORE
                if (o0 == 0x12345678u && o1 == 0x9abcdef0u)
#endif
                *reinterpret_cast<U3 *>(o) = U3{o0, o1, o2}; // one dwordx3 per lane: 768 contiguous bytes per wave instruction
            } else {
                if (full) *reinterpret_cast<U3 *>(o) = U3{o0, o1, o2};
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((keep[j >> 1] >> (16 * (j & 1))) & 1u) { o[3 * j] = (uint8_t)R[j]; o[3 * j + 1] = (uint8_t)G[j]; o[3 * j + 2] = (uint8_t)B[j]; }
                }
            }
        }
    };

    // sample pass: every sample_stride-th step of the piece (a prediction only: no alignment across pieces needed)
    // (midpoint phase: a sample that starts every piece at its first row is a left Riemann sum over the blend weight and
    // biases the level counts in proportion to the stride)
    const int stride = (MODE == kFusedSample) ? gy * (int)a.sample_stride : gy;
    // ... and every wave column takes its own phase: the sampled positions form a lattice over the cell instead of a few full rows
    const int rfirst = __builtin_amdgcn_readfirstlane(I.r0 + wy + (MODE == kFusedSample ? gy * (int)((a.sample_stride * (2u * (uint32_t)wx + 1u)) / (2u * (uint32_t)gx)) : 0));
    if (rfirst >= I.r1) return;
    const int rlast = I.r1 - 1;
    // The row weight travels with the row as a vector load (a scalar load would share lgkmcnt with the LDS gathers).
    auto ld = [&](int rr, uint2 &u1, uint2 &u2, float &dy) {
        int rc = min(rr, rlast);
        u1 = *reinterpret_cast<const uint2 *>(p1 + (size_t)rc * a.in_pitch);
        u2 = *reinterpret_cast<const uint2 *>(p2 + (size_t)rc * a.in_pitch);
#ifdef FUSED_ABL_NODY
        dy = 0.25f;
#else
        asm volatile("" : "+v"(rc)); // keep the index in a VGPR: global_load, not s_load
        dy = row_wf[rc];
#endif
    };
#ifndef FUSED_RING
#define FUSED_RING 4
#endif
    auto run = [&](auto allfull_t) {
        constexpr int kRing = FUSED_RING;
        uint2 u1[kRing], u2[kRing];
        float dy[kRing];
        int r = rfirst;
#pragma unroll
        for (int k = 0; k < kRing; ++k) ld(r + k * stride, u1[k], u2[k], dy[k]);
        v4f e1[4];
        uint32_t addr1[4];
        gather(0, u1[0], e1, addr1);
        while (true) {
#pragma unroll
            for (int k = 0; k < kRing; ++k) {
                process(allfull_t, r, u1[k], u2[k], dy[k], u1[(k + 1) % kRing], e1, addr1);
                ld(r + kRing * stride, u1[k], u2[k], dy[k]);
                r += stride;
                if (r > rlast) return;
            }
        }
    };
#ifdef FUSED_EXP_STAGGER
    for (int w = 0; w < wave; ++w) __builtin_amdgcn_s_sleep(FUSED_EXP_STAGGER); // experiment: desynchronise the waves of the workgroup
#endif
    const bool allfull = __builtin_amdgcn_ballot_w64(!full) == 0ull;
    if (allfull) run(std::true_type{});
    else run(std::false_type{});
}

template <int MODE>
__global__ __launch_bounds__(kFBlock) void k_fused_main(FusedArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    FusedState *fs = a.fs;
    if (!pass_runs(fs, MODE)) return;
    const int first = a.wg_first[blockIdx.x], last = a.wg_first[blockIdx.x + 1];
    if (first >= last) return;
    const bool direct = fs->direct != 0 && MODE != kFusedSample; // the sample pass needs the bin of every pixel: two-level addressing
    const uint32_t kb[2] = {fs->k_base[0], fs->k_base[1]};
    const uint32_t hi[2] = {a.state[0].win_hi, a.state[1].win_hi};
    const uint32_t qcap = (a.force & kFusedForceTinyQueue) ? 4u : a.qoff[blockIdx.x + 1] - a.qoff[blockIdx.x];
    constexpr bool kCompose = MODE == kFusedSpec || MODE == kFusedFinal;
    if (kCompose) {
        const v4u *src = reinterpret_cast<const v4u *>(a.tables);
        for (int i = threadIdx.x; i < 512 / 16; i += kFBlock) LDS_AT(v4u, kLdsRG + i * 16) = src[i];
        for (int i = threadIdx.x; i < 65536 / 16; i += kFBlock) LDS_AT(v4u, kLdsB2 + i * 16) = src[32 + i];
    }
    for (int i = threadIdx.x; i < 512; i += kFBlock) LDS_AT(uint32_t, kLdsHist + i * 4) = 0u;
    if (MODE == kFusedSample)
        for (int i = threadIdx.x; i < 16384; i += kFBlock) LDS_AT(uint32_t, kLdsB2 + i * 4) = 0u;
    int cur_tile = -1;
    auto flush_sample = [&]() { // the piece's tile changes: publish and clear the sampled counts (all threads, between barriers)
        if (cur_tile < 0) return;
        uint32_t *g = a.hist3 + (size_t)cur_tile * 256 * 32;
        for (int i = threadIdx.x; i < 16384; i += kFBlock) {
            const uint32_t v = LDS_AT(uint32_t, kLdsB2 + i * 4);
            if (v) { atomicAdd(&g[(size_t)(i >> 13) * 64 * 256 * 32 + (i & 8191)], v); LDS_AT(uint32_t, kLdsB2 + i * 4) = 0u; }
        }
    };
    if (threadIdx.x == 0) LDS_AT(uint32_t, kLdsMisc) = 0u;
    uint32_t thrK0 = 0, thrK1 = 0;
    if (MODE == kFusedSpec) {
        const uint32_t f = (uint32_t)fs->floor_pred;
        thrK0 = (0x80u - f) * 0x01010101u;
        thrK1 = (0x80u - (f + 1u)) * 0x01010101u;
    }
    uint32_t acc0 = 0, acc1 = 0, nsteps = 0;
    int cur[4] = {-1, -1, -1, -1};
    int cur_flags = -1;
    for (int it = first; it < last; ++it) {
        const FusedItem I = a.items[it];
        if (I.id[0] != cur[0] || I.id[1] != cur[1] || I.id[2] != cur[2] || I.id[3] != cur[3] || (I.flags & 1) != cur_flags || (MODE == kFusedSample && I.tile != cur_tile)) {
            __syncthreads(); // everybody is done with the previous cell's tables (and the compose tables are in place)
            if (MODE == kFusedSample && I.tile != cur_tile) { flush_sample(); cur_tile = I.tile; }
            stage_cell(a, I, direct, kb, hi);
            __syncthreads();
            cur[0] = I.id[0]; cur[1] = I.id[1]; cur[2] = I.id[2]; cur[3] = I.id[3]; cur_flags = I.flags & 1;
        }
#ifdef FUSED_EXP_ROTATE
        {   // experiment: every workgroup starts its piece at a different relative row (decorrelates the address streams)
            const int gy = kFWaves >> I.gx_log2;
            const int nst = (I.r1 - I.r0 + gy - 1) / gy;
            const int rs = I.r0 + gy * (int)((blockIdx.x * 2654435761u >> 8) % (uint32_t)max(nst, 1));
            FusedItem I1 = I, I2 = I;
            I1.r0 = rs; I2.r1 = rs;
            if (direct) { fused_rows<MODE, true>(a, I1, kb, hi, qcap, thrK0, thrK1, acc0, acc1, nsteps); fused_rows<MODE, true>(a, I2, kb, hi, qcap, thrK0, thrK1, acc0, acc1, nsteps); }
            else { fused_rows<MODE, false>(a, I1, kb, hi, qcap, thrK0, thrK1, acc0, acc1, nsteps); fused_rows<MODE, false>(a, I2, kb, hi, qcap, thrK0, thrK1, acc0, acc1, nsteps); }
        }
#else
        if (direct) fused_rows<MODE, true>(a, I, kb, hi, qcap, thrK0, thrK1, acc0, acc1, nsteps);
        else fused_rows<MODE, false>(a, I, kb, hi, qcap, thrK0, thrK1, acc0, acc1, nsteps);
#endif
    }
    __syncthreads();
    // a workgroup whose queue overflowed publishes nothing: k_fused_fixup redoes its whole share exactly
    const uint32_t queued = (MODE != kFusedSample) ? LDS_AT(uint32_t, kLdsMisc) : 0u;
    const bool over = queued > qcap;
    if (MODE == kFusedSpec && !over) {
        // 8 level bytes per step: below threshold = 8 * steps - (bytes at or above it, foreign bytes included)
        unsigned long long lt0 = 8ull * nsteps - acc0, lt1 = 8ull * nsteps - acc1;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) { lt0 += __shfl_xor(lt0, m, 64); lt1 += __shfl_xor(lt1, m, 64); }
        if (f_lane() == 0) { atomicAdd(&fs->n_lt[0], lt0); atomicAdd(&fs->n_lt[1], lt1); }
    }
    if (MODE == kFusedSample) flush_sample();
    if (MODE == kFusedHist && !over) {
        for (int i = threadIdx.x; i < 512; i += kFBlock) {
            const uint32_t n = LDS_AT(uint32_t, kLdsHist + i * 4);
            if (n && (i & 255)) atomicAdd(&a.level_hist[i], (unsigned long long)n);
        }
    }
    if (MODE != kFusedSample && threadIdx.x == 0) fs->qcount[MODE][blockIdx.x] = over ? 0x80000000u : queued;
}

// ------------------------------------------------------------------------------------
// The queued pixels, exactly: RGB patch (speculative / final pass), count corrections (speculative pass),
// histogram corrections (histogram pass).  The last workgroup of the speculative pass's fixup verifies the floor.
// ------------------------------------------------------------------------------------
// Flat over ALL queued pixels of the scene: the queues differ in length by 40x (extrapolating cells, bright regions), so
// every block takes an equal slice of the concatenated queues (prefix sums of the per-workgroup counts in LDS, binary
// search per entry).  A workgroup whose queue overflowed contributes its whole share instead (every pixel, exactly).
constexpr int kFixupBlocks = 256, kFixupThreads = 1024; // few fat blocks: the prefix sums are the fixed cost of a block
template <int MODE>
__global__ __launch_bounds__(kFixupThreads) void k_fused_fixup(FusedArgs a, unsigned long long total_px, int nwg) {
    FusedState *fs = a.fs;
    if (!pass_runs(fs, MODE)) return;
    __shared__ unsigned long long pre[kFusedMaxGrid + 1]; // work units before workgroup w: queue entries, or pixels of a share to redo
    const int t = threadIdx.x;
    {   // inclusive scan of the (<= 1024) unit counts, one per thread
        unsigned long long n = 0;
        if (t < nwg) {
            const uint32_t qc = fs->qcount[MODE][t];
            if (qc & 0x80000000u) {
                for (int it = a.wg_first[t]; it < a.wg_first[t + 1]; ++it) n += (unsigned long long)(a.items[it].r1 - a.items[it].r0) * (unsigned long long)(a.items[it].c1 - a.items[it].c0);
            } else n = qc;
        }
        if (t == 0) pre[0] = 0ull;
        pre[t + 1] = n;
        __syncthreads();
        for (int off = 1; off < kFusedMaxGrid; off <<= 1) {
            const unsigned long long x = (t + 1 > off) ? pre[t + 1 - off] : 0ull;
            __syncthreads();
            pre[t + 1] += x;
            __syncthreads();
        }
    }
    const unsigned long long total = pre[kFusedMaxGrid];
    const uint32_t hi[2] = {a.state[0].win_hi, a.state[1].win_hi};
    const uint32_t F = (uint32_t)fs->floor_pred;
    const uint8_t *R2 = a.tables, *G2 = a.tables + 256, *B2 = a.tables + 512;
    unsigned long long d0 = 0, d1 = 0;
    // the main pass left these pixels out of its counts / histograms and wrote a provisional RGB for them
    auto exact_px = [&](uint32_t r, uint32_t c, uint32_t dn1, uint32_t dn2) {
        const RowWeight rw = a.row_w[a.row_off + r], cw = a.col_w[c];
        const int32_t id[4] = {rw.t0 * kTiles + cw.t0, rw.t0 * kTiles + cw.t1, rw.t1 * kTiles + cw.t0, rw.t1 * kTiles + cw.t1};
#ifdef FIX_ABL_NOMATH
        const uint32_t l1 = (dn1 + id[0]) & 255u, l2 = (dn2 + id[3]) & 255u;
#else
        const uint32_t l1 = exact_level(a.cdfs[0], a.binlut[0], hi[0], dn1, id, cw.d, rw.d);
        const uint32_t l2 = exact_level(a.cdfs[1], a.binlut[1], hi[1], dn2, id, cw.d, rw.d);
#endif
        if (MODE == kFusedSpec) { d0 += (uint32_t)(l1 < F) + (uint32_t)(l2 < F); d1 += (uint32_t)(l1 < F + 1) + (uint32_t)(l2 < F + 1); }
        if (MODE == kFusedHist) { if (l1) atomicAdd(&a.level_hist[l1], 1ull); if (l2) atomicAdd(&a.level_hist[256 + l2], 1ull); }
#ifdef FIX_ABL_NOSTORE
        if (l1 == 300u)
#endif
        if (MODE == kFusedSpec || MODE == kFusedFinal) {
            uint8_t *o = a.rgb + ((size_t)r * a.rgb_pitch_px + c) * 3;
            o[0] = R2[l1]; o[1] = G2[l2]; o[2] = B2[(l1 << 8) | l2];
        }
    };
    for (unsigned long long i = (unsigned long long)blockIdx.x * kFixupThreads + t; i < total; i += (unsigned long long)gridDim.x * kFixupThreads) {
        int lo = 0, hi_w = kFusedMaxGrid; // largest w with pre[w] <= i
        while (hi_w - lo > 1) { const int mid = (lo + hi_w) >> 1; if (pre[mid] <= i) lo = mid; else hi_w = mid; }
        const int w = lo;
        unsigned long long k = i - pre[w];
        if (!(fs->qcount[MODE][w] & 0x80000000u)) {
            const uint4 q = a.queue[(size_t)a.qoff[w] + k]; // (row, column, DN1 | DN2 << 16): no second visit to the rasters
            exact_px(q.x, q.y, q.z & 0xFFFFu, q.z >> 16);
        } else { // pixel k of the workgroup's share
            for (int it = a.wg_first[w]; it < a.wg_first[w + 1]; ++it) {
                const FusedItem I = a.items[it];
                const unsigned long long wpx = (unsigned long long)(I.c1 - I.c0), npx = (unsigned long long)(I.r1 - I.r0) * wpx;
                if (k < npx) {
                    const uint32_t r = (uint32_t)I.r0 + (uint32_t)(k / wpx), c = (uint32_t)I.c0 + (uint32_t)(k % wpx);
                    exact_px(r, c, a.in[0][(size_t)r * a.in_pitch + c], a.in[1][(size_t)r * a.in_pitch + c]);
                    break;
                }
                k -= npx;
            }
        }
    }
    if (MODE != kFusedSpec) return;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { d0 += __shfl_xor(d0, m, 64); d1 += __shfl_xor(d1, m, 64); }
    if ((t & 63) == 0) {
        if (d0) atomicAdd(&fs->n_lt[0], d0);
        if (d1) atomicAdd(&fs->n_lt[1], d1);
    }
    __threadfence();
    __syncthreads();
    __shared__ uint32_t ticket;
    if (t == 0) ticket = atomicAdd(&fs->fix_done[MODE], 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1 || t != 0) return;
    __threadfence();
    // synthetic_rgb.rs:99-113: floor = first level whose cumulative count of BOTH final bands reaches round(total * 0.05)
    const unsigned long long lt0 = atomicAdd(&fs->n_lt[0], 0ull), lt1 = atomicAdd(&fs->n_lt[1], 0ull); // cum(F-1), cum(F)
    const uint32_t tot2 = (uint32_t)(total_px + total_px);
    const double tc = round((double)tot2 * 0.05);
    const unsigned long long target = tc >= 4294967295.0 ? 4294967295ull : (unsigned long long)tc;
    const bool ok = lt1 >= target && (F == 0 || lt0 < target);
    fs->verdict = ok ? 0u : 1u;
}

} // namespace

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
hipError_t fused_configure() {
    hipError_t e;
    if ((e = opt_in_dynamic_lds(reinterpret_cast<const void *>(&k_fused_main<kFusedSample>))) != hipSuccess) return e;
    if ((e = opt_in_dynamic_lds(reinterpret_cast<const void *>(&k_fused_main<kFusedSpec>))) != hipSuccess) return e;
    if ((e = opt_in_dynamic_lds(reinterpret_cast<const void *>(&k_fused_main<kFusedHist>))) != hipSuccess) return e;
    return opt_in_dynamic_lds(reinterpret_cast<const void *>(&k_fused_main<kFusedFinal>));
}

hipError_t launch_fused_prep(const FusedPrepArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_fused_prep, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fused_predict(const FusedPredictArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_fused_predict, dim3(64, 2), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_fused_tables_predict(const FusedTablesArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_fused_tables_predict, dim3(64), dim3(256), 0, s, a);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------
// The per-tile DN histograms of BOTH bands over the fused pass's pieces: persistent workgroups of 1024 threads, each on its
// static share of the scene (strips of 2^k wave columns x row ranges inside one tile), reading the two rasters at the rate the
// piece traversal streams them (tools/stream_bench.hip: 5.6-6.2 TB/s against 4.6 for the 256-thread strip items of
// kernels.hip 1b).  Counting as there: one unconditional ds_add_u32 per pixel -- DN in [1, W) to its bin, DN = 0 and the
// bright tail to a per-lane dummy word, tail pixels to the tile's global histogram -- and DN = 0 is not accumulated at all
// (bin 0 is what is left of the tile, restored by the consumer).  The LDS histograms are published when the tile changes.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFBlock) void k_dn_hist_pieces(DnHistPiecesArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    uint32_t *h = reinterpret_cast<uint32_t *>(lds); // band b: [b * (W + 64), + W) bins, then 64 dummy words
    const uint32_t W = a.lds_bins, S = W + 64u;
    const int first = a.wg_first[blockIdx.x], last = a.wg_first[blockIdx.x + 1];
    if (first >= last) return;
    for (uint32_t i = threadIdx.x; i < 2u * S; i += kFBlock) h[i] = 0u;
    __syncthreads();
    const int wave = f_wave(), lane = f_lane();
    const uint32_t dummy[2] = {(W + (uint32_t)lane) * 4u, (S + W + (uint32_t)lane) * 4u};
    int cur_tile = -1;
    auto publish = [&]() { // all threads, between barriers
        if (cur_tile < 0) return;
        for (int b = 0; b < 2; ++b) {
            uint32_t *g = a.tile_hist[b] + (size_t)cur_tile * 65536u;
            for (uint32_t i = threadIdx.x + 1; i < W; i += kFBlock) {
                const uint32_t n = h[b * S + i];
                if (n) { atomicAdd(&g[i], n); h[b * S + i] = 0u; }
            }
        }
    };
    for (int it = first; it < last; ++it) {
        const FusedItem I = a.items[it];
        if (I.tile != cur_tile) {
            __syncthreads();
            publish();
            cur_tile = I.tile;
            __syncthreads();
        }
        const int gx = 1 << I.gx_log2, gy = kFWaves >> I.gx_log2;
        const int wx = wave & (gx - 1), wy = wave >> I.gx_log2;
        const int col = I.cstart + (wx * 64 + lane) * 4;
        if (col >= I.c1 || col + 4 <= I.c0) continue;
        // samples outside the piece's columns become DN = 0 (never counted)
        uint32_t m[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col + j >= I.c0 && col + j < I.c1) m[j >> 1] |= 0xFFFFu << (16 * (j & 1));
        uint32_t *const gt[2] = {a.tile_hist[0] + (size_t)I.tile * 65536u, a.tile_hist[1] + (size_t)I.tile * 65536u};
        auto consume = [&](int b, uint2 w) {
            const uint32_t ww[2] = {w.x & m[0], w.y & m[1]};
            uint32_t big = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t d = (j & 1) ? (ww[j >> 1] >> 16) : (ww[j >> 1] & 0xFFFFu);
                const bool in_lds = d - 1u < W - 1u; // 1 <= d < W
                big |= (d >= W ? 1u : 0u) << j;
                const uint32_t off = in_lds ? (b ? S * 4u : 0u) + d * 4u : dummy[b];
#ifdef PIECE_HIST_NO_ATOMICS // timing experiment: the traversal and the address arithmetic without the LDS atomics
                big += off;
#else
                LDS_ADD(off, 1u);
#endif
            }
#ifdef PIECE_HIST_NO_ATOMICS
            if (big == 0xFFFFFFFFu) atomicAdd(&gt[b][0], 1u);
#else
            if (big) { // bright tail: rare
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((big >> j) & 1u) atomicAdd(&gt[b][(j & 1) ? (ww[j >> 1] >> 16) : (ww[j >> 1] & 0xFFFFu)], 1u);
            }
#endif
        };
        const uint16_t *__restrict__ p1 = a.in[0] + col, *__restrict__ p2 = a.in[1] + col;
        int r = I.r0 + wy;
        if (r < I.r1) { // two rows in flight per wave: the next row's loads are issued before this row is counted
            const int lastr = I.r1 - 1;
            uint2 n1 = *reinterpret_cast<const uint2 *>(p1 + (size_t)r * a.pitch), n2 = *reinterpret_cast<const uint2 *>(p2 + (size_t)r * a.pitch);
            for (; r < I.r1; r += gy) {
                const uint2 c1 = n1, c2 = n2;
                const int rn = min(r + gy, lastr);
                n1 = *reinterpret_cast<const uint2 *>(p1 + (size_t)rn * a.pitch);
                n2 = *reinterpret_cast<const uint2 *>(p2 + (size_t)rn * a.pitch);
                consume(0, c1);
                consume(1, c2);
            }
        }
    }
    __syncthreads();
    publish();
}

hipError_t launch_dn_hist_pieces(const DnHistPiecesArgs &a, int grid, hipStream_t s) {
    if (grid <= 0 || grid > kFusedMaxGrid) return hipErrorInvalidValue;
    const size_t lds = 2 * ((size_t)a.lds_bins + 64) * sizeof(uint32_t);
    if (lds > kLdsLaunch) return hipErrorInvalidValue;
    if (hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_dn_hist_pieces))) return e;
    hipLaunchKernelGGL(k_dn_hist_pieces, dim3(grid), dim3(kFBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fused_main(const FusedArgs &a, int mode, int grid, hipStream_t s) {
    if (grid <= 0 || grid > kFusedMaxGrid) return hipErrorInvalidValue;
    switch (mode) {
    case kFusedSample: hipLaunchKernelGGL(k_fused_main<kFusedSample>, dim3(grid), dim3(kFBlock), kLdsLaunch, s, a); break;
    case kFusedSpec: hipLaunchKernelGGL(k_fused_main<kFusedSpec>, dim3(grid), dim3(kFBlock), kLdsLaunch, s, a); break;
    case kFusedHist: hipLaunchKernelGGL(k_fused_main<kFusedHist>, dim3(grid), dim3(kFBlock), kLdsLaunch, s, a); break;
    case kFusedFinal: hipLaunchKernelGGL(k_fused_main<kFusedFinal>, dim3(grid), dim3(kFBlock), kLdsLaunch, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_fused_fixup(const FusedArgs &a, int mode, int grid, unsigned long long total_px, hipStream_t s) {
    if (grid <= 0 || grid > kFusedMaxGrid) return hipErrorInvalidValue;
    switch (mode) {
    case kFusedSpec: hipLaunchKernelGGL(k_fused_fixup<kFusedSpec>, dim3(kFixupBlocks), dim3(kFixupThreads), 0, s, a, total_px, grid); break;
    case kFusedHist: hipLaunchKernelGGL(k_fused_fixup<kFusedHist>, dim3(kFixupBlocks), dim3(kFixupThreads), 0, s, a, total_px, grid); break;
    case kFusedFinal: hipLaunchKernelGGL(k_fused_fixup<kFusedFinal>, dim3(kFixupBlocks), dim3(kFixupThreads), 0, s, a, total_px, grid); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace sarpro

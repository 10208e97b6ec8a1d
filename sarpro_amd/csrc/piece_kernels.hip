// piece_kernels.hip -- per-tile DN histograms of a whole dual-pol scene on gfx950, walked as cost-balanced "pieces" by
// persistent 1024-thread workgroups (the first pass of the CLAHE chain: autoscale.rs:259-268 needs the per-tile counts,
// autoscale.rs:35-160 their sum).  Round 2's one-sweep CLAHE -> RGB pass lived on the same traversal; it measured slower
// than apply + compose (NOTEBOOK.md section 6b keeps the numbers) and was removed in round 3.
#include "piece_kernels.h"

namespace sarpro {

namespace {

constexpr int kPBlock = kPieceBlock, kPWaves = kPieceWaves;
// DN 1..kLowBins-1 are counted in the LANE'S OWN copy of those bins (round 5 layout: [band][lane][kLowStride = 129 words], so that a
// sample's word is at DN * 4 + a per-lane constant -- the same multiply as the shared bins, one select instead of two -- and the word
// of lane l for bin d lies on bank (l + d) mod 64: lanes that hold the same DN never meet on a bank; rounds 2-4: [band][DN][64 lanes], and
// lanes l and l + 32 belong to different halves of a ds_add_u32 -- no two lanes of an instruction ever meet, neither on a bank nor on
// an address).  A band of a few distinct low amplitudes (cross-pol over open water: most samples on five or ten DN values) otherwise
// sends a whole wave's adds to a handful of LDS words, which serialise: 1.34 ms instead of 0.31 for the pass on a scene whose VH
// band holds DN 1..10 only; eight lane-selected copies (the first form of this) left 0.72.  The copies are summed on publish.
#ifndef SARPRO_PIECE_LOWBINS
#define SARPRO_PIECE_LOWBINS 64
#endif
// LDS footprint (round 5): 2 x 8256 shared bins + 2 x 64 lanes x 65 low words + the tail queue = 104 KiB (rounds 2-4: 144 KiB with 128
// low bins per lane).  What is left of the CU's 160 KiB lets the chain's short kernels (statistics: 48 KiB, tile bins, CDFs, sample,
// prediction) start beside this pass when another lane of a resident batch has them queued, instead of behind it.  The pass itself
// runs the same with 32, 64 or 128 low bins per lane (0.295-0.300 ms).
constexpr uint32_t kLowBins = SARPRO_PIECE_LOWBINS, kLowReps = 64, kLowStride = kLowBins + 1, kLowWords = kLowReps * kLowStride; // words per band
// LDS atomics through the address-space-3 pointer (the HIP overloads take generic pointers)
#define LDS_ADD(off, v) __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uint32_t)(off)), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// The bright tail (DN >= the LDS bins: point targets, one sample in 10^4 on a GRD scene) goes to the tile's GLOBAL histogram.  Round 5: not
// from inside the row loop.  With global atomics in a divergent branch of the loop the compiler cannot count the memory operations in
// flight and closed every iteration with s_waitcnt vmcnt(0) -- the next row's loads, issued at the top of the iteration, had to be back
// before the iteration ended: one row in flight per wave, whatever the source said.  The tail samples are queued in LDS instead
// ((band << 16) | DN, a slot reserved by one returning ds_add per lane and band-row) and added to the global histogram when the
// workgroup publishes its tile; the row loop holds no vector-memory instruction but its loads and waits with counted vmcnt.  A lane
// that finds the queue full (a raster that is mostly brighter than the LDS bins) stops queueing for the rest of the piece and
// recounts its tail samples from that band-row on in a second loop behind the first: slower, never wrong.
#ifndef SARPRO_PIECE_TAILCAP
#define SARPRO_PIECE_TAILCAP 1024
#endif
constexpr uint32_t kTailCap = SARPRO_PIECE_TAILCAP;
#ifndef SARPRO_PIECE_AHEAD
#define SARPRO_PIECE_AHEAD 1 // rows of both bands in flight per wave beyond the one being counted (1, 2, 3, 4 measured: 0.300, 0.312, 0.311, 0.313 ms)
#endif
constexpr int kPieceAhead = SARPRO_PIECE_AHEAD;
static_assert(kPieceAhead >= 1 && kPieceAhead <= 4, "piece histogram: 1..4 rows ahead");
struct PieceRow { uint32_t w[kPieceVec / 2]; }; // kPieceVec samples of a band-row, two per dword
__device__ __forceinline__ PieceRow piece_load(const uint16_t *p) {
    PieceRow r;
    if (kPieceVec == 8) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        r.w[0] = t.x; r.w[1] = t.y; r.w[kPieceVec / 2 - 2] = t.z; r.w[kPieceVec / 2 - 1] = t.w;
    } else {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        r.w[0] = t.x; r.w[1] = t.y;
    }
    return r;
}
#define PIECE_LOAD(p) piece_load(p)
__device__ __forceinline__ int p_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int p_lane() { return (int)(threadIdx.x & 63); }

// ------------------------------------------------------------------------------------
// The per-tile DN histograms of BOTH bands over balanced pieces of the scene: persistent workgroups of 1024 threads, each on its
// static share of the scene (strips of 2^k wave columns x row ranges inside one tile), reading the two rasters at the rate the
// piece traversal streams them (tools/stream_bench.hip: 5.6-6.2 TB/s against 4.6 for the 256-thread strip items of
// kernels.hip 1b).  Counting: one unconditional ds_add_u32 per pixel -- DN in [kLowBins, W) to its shared bin, DN below kLowBins (0, the
// invalid sample, included) to the lane's own word of that bin, the bright tail to a per-lane dummy word and, by a rare branch, to
// the tile's global histogram -- ; bin 0 is not published (it is what is left of the tile, restored by the consumer).  The LDS
// histograms are published when the tile changes.
// ------------------------------------------------------------------------------------
#ifdef SARPRO_PIECE_WG_TIMES // instrumented build (tools/rgb_wg_times.py): when each persistent workgroup of the histogram pass started and ended (100 MHz clock)
__device__ unsigned long long g_piece_wg_times[1024][2];
#endif
__global__ __launch_bounds__(kPBlock) void k_dn_hist_pieces(DnHistPiecesArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
#ifdef SARPRO_PIECE_WG_TIMES
    if (threadIdx.x == 0) g_piece_wg_times[blockIdx.x & 1023][0] = wall_clock64();
#endif
    uint32_t *h = reinterpret_cast<uint32_t *>(lds); // band b: [b * (W + 64), + W) bins, then 64 dummy words
    const uint32_t W = a.lds_bins, S = W + 64u;
    const int first = a.wg_first[blockIdx.x], last = a.wg_first[blockIdx.x + 1];
    if (first >= last) return;
    for (uint32_t i = threadIdx.x; i < 2u * S + 2u * kLowWords; i += kPBlock) h[i] = 0u;
    uint32_t *const q = h + 2u * S + 2u * kLowWords; // q[0]: entries reserved so far; q[1 + k]: entry k
    if (threadIdx.x == 0) q[0] = 0u;
    const uint32_t q_off = (2u * S + 2u * kLowWords) * 4u; // byte offset of q[0]
    __syncthreads();
    const int wave = p_wave(), lane = p_lane();
    // byte offset of this lane's word of low bin 0, per band: behind the two histograms, [band][lane][kLowStride]
    const uint32_t low_off[2] = {(2u * S + (uint32_t)lane * kLowStride) * 4u, (2u * S + kLowWords + (uint32_t)lane * kLowStride) * 4u};
    int cur_tile = -1;
    auto publish = [&]() { // all threads, between barriers
        if (cur_tile < 0) return;
        // the lanes' own words of the low bins first, folded into the shared bins: thread (band, DN) sums its bin over the 64 lanes (a wave's
        // threads hold consecutive DN of one lane's row at a time: consecutive banks)
        if (threadIdx.x < 2u * kLowBins) {
            const uint32_t b = threadIdx.x / kLowBins, dn = threadIdx.x % kLowBins;
            uint32_t *p = &h[2u * S + b * kLowWords + dn];
            uint32_t n = 0u;
#pragma unroll 8
            for (uint32_t l = 0; l < kLowReps; ++l) { const uint32_t v = p[l * kLowStride]; if (v) { n += v; p[l * kLowStride] = 0u; } }
            if (n && dn) LDS_ADD((b * S + dn) * 4u, n); // (low bin 0 holds the invalid samples: bin 0 is what is left of the tile, restored by the consumer)
        }
        {   // the queued tail samples of this tile (entry 0 = a slot reserved by a lane that then found the queue full)
            const uint32_t nq = min(q[0], kTailCap);
            for (uint32_t i = threadIdx.x; i < nq; i += kPBlock) {
                const uint32_t e = q[1u + i];
                if (e) atomicAdd(&a.tile_hist[e >> 16][(size_t)cur_tile * 65536u + (e & 0xFFFFu)], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) q[0] = 0u; // (the caller's barrier stands between this and the next piece's reservations)
        for (int b = 0; b < 2; ++b) {
            uint32_t *g = a.tile_hist[b] + (size_t)cur_tile * 65536u;
            for (uint32_t i = threadIdx.x + 1; i < W; i += kPBlock) {
                const uint32_t n = h[b * S + i];
                if (n) { h[b * S + i] = 0u; atomicAdd(&g[i], n); }
            }
        }
    };
    for (int it = first; it < last; ++it) {
        const PieceItem I = a.items[it];
        if (I.tile != cur_tile) {
            __syncthreads();
            publish();
            cur_tile = I.tile;
            __syncthreads();
        }
        const int gx = 1 << I.gx_log2, gy = kPWaves >> I.gx_log2;
        const int wx = wave & (gx - 1), wy = wave >> I.gx_log2;
        constexpr int V = kPieceVec;
        const int col = I.cstart + (wx * 64 + lane) * V;
        if (col >= I.c1 || col + V <= I.c0) continue;
        // samples outside the piece's columns become DN = 0 (never counted)
        uint32_t m[V / 2];
#pragma unroll
        for (int k = 0; k < V / 2; ++k) m[k] = 0u;
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (col + j >= I.c0 && col + j < I.c1) m[j >> 1] |= 0xFFFFu << (16 * (j & 1));
        uint32_t *const gt[2] = {a.tile_hist[0] + (size_t)I.tile * 65536u, a.tile_hist[1] + (size_t)I.tile * 65536u};
        // Per sample: min(DN, W + lane) straight from its 16-bit half (a tail sample lands on the lane's dummy word behind the bins), one
        // compare against the low range -- DN = 0 included: it goes to the lane's own word of low bin 0, which nobody reads (the no-data
        // wedge never meets another lane) --, stride and base by that compare, one multiply-add, one ds_add_u32: five VALU
        // instructions.  The bright tail is found by ONE packed max per dword and a wave-level test per band-row; its samples then go
        // to the tile's global histogram one by one (rare).
        const uint32_t Wl = W + (uint32_t)lane;
        const uint32_t tail_mask = ~(W - 1u) & 0xFFFFu; // (W is a power of two: a half >= W has one of these bits)
        bool failed = false; // this lane found the tail queue full in this piece: from (fail_r, fail_b) on its tail samples are recounted below
        int fail_r = 0, fail_b = 0;
        auto consume = [&](int b, int r, const PieceRow &w) {
            uint32_t ww[V / 2], mx = 0u;
            typedef unsigned short v2us __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int k = 0; k < V / 2; ++k) {
                ww[k] = w.w[k] & m[k];
                mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2us, mx), __builtin_bit_cast(v2us, ww[k])));
            }
            const uint32_t band_base = b ? S * 4u : 0u;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                // min(DN, W + lane) straight from the sample's 16-bit half (SDWA source select: no extract), then DN * 4 + (low ? this lane's
                // row of low bins : the band's shared bins): four vector instructions per sample (rounds 2-4: six)
                uint32_t dc;
                if (j & 1) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(dc) : "v"(ww[j >> 1]), "v"(Wl));
                else asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(dc) : "v"(ww[j >> 1]), "v"(Wl));
                const uint32_t off = dc * 4u + (dc < kLowBins ? low_off[b] : band_base);
#ifdef PIECE_HIST_NO_ATOMICS // timing experiment: the traversal and the address arithmetic without the LDS atomics
                mx += off;
#else
                LDS_ADD(off, 1u);
#endif
            }
#ifdef PIECE_HIST_NO_ATOMICS
            if (mx == 0xFFFFFFFFu) LDS_ADD(q_off, 1u);
#else
            if (((mx | (mx >> 16)) & tail_mask) && !failed) { // bright tail: rare
                uint32_t nt = 0u;
#pragma unroll
                for (int j = 0; j < V; ++j) nt += (((j & 1) ? (ww[j >> 1] >> 16) : (ww[j >> 1] & 0xFFFFu)) >= W) ? 1u : 0u;
                uint32_t k = LDS_ADD(q_off, nt); // (returns the old count: this lane's first slot)
                if (k + nt <= kTailCap) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const uint32_t d = (j & 1) ? (ww[j >> 1] >> 16) : (ww[j >> 1] & 0xFFFFu);
                        if (d >= W) { q[1u + k] = ((uint32_t)b << 16) | d; ++k; }
                    }
                } else { // full: the slots this lane holds inside the queue become no-ops, the lane recounts from here on
                    for (uint32_t e = k; e < min(k + nt, kTailCap); ++e) q[1u + e] = 0u;
                    failed = true; fail_r = r; fail_b = b;
                }
            }
#endif
        };
        const uint16_t *__restrict__ p1 = a.in[0] + col, *__restrict__ p2 = a.in[1] + col;
        int r = I.r0 + wy;
        if (r < I.r1) { // kPieceAhead rows of both bands in flight per wave behind the one being counted
            // A ring of kPieceAhead + 1 row slots with STATIC indices (the trip below is unrolled over the slots): rotating the rows
            // through registers instead costs a v_mov of a register whose load is still in flight, i.e. a wait for that load.
            const int lastr = I.r1 - 1;
            constexpr int NS = kPieceAhead + 1;
            PieceRow n1[NS], n2[NS];
#pragma unroll
            for (int k = 0; k < kPieceAhead; ++k) {
                const int rk = min(r + k * gy, lastr);
                n1[k] = PIECE_LOAD(p1 + (size_t)rk * a.pitch);
                n2[k] = PIECE_LOAD(p2 + (size_t)rk * a.pitch);
            }
            // whole trips of NS rows without a branch between the slots (a conditional load makes the compiler wait for every load in
            // flight at the join); the last rows of the piece one by one behind them, from the same ring
            while (r + (NS - 1) * gy < I.r1) {
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int rn = min(r + kPieceAhead * gy, lastr);
                    n1[(sl + kPieceAhead) % NS] = PIECE_LOAD(p1 + (size_t)rn * a.pitch);
                    n2[(sl + kPieceAhead) % NS] = PIECE_LOAD(p2 + (size_t)rn * a.pitch);
                    consume(0, r, n1[sl]);
                    consume(1, r, n2[sl]);
                    r += gy;
                }
            }
#pragma unroll
            for (int sl = 0; sl < NS - 1; ++sl) { // (slot sl holds row r: the trips above end on slot 0; the rows ahead were loaded clamped to the last row)
                if (r < I.r1) {
                    consume(0, r, n1[sl]);
                    consume(1, r, n2[sl]);
                    r += gy;
                }
            }
        }
#ifndef PIECE_HIST_NO_ATOMICS
        if (failed) { // the queue was full: this lane's tail samples from (fail_r, fail_b) on, straight to the tile's global histogram
            for (int r2 = fail_r; r2 < I.r1; r2 += gy) {
                const PieceRow wa = PIECE_LOAD(p1 + (size_t)r2 * a.pitch), wb = PIECE_LOAD(p2 + (size_t)r2 * a.pitch);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (r2 == fail_r && b < fail_b) continue;
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const uint32_t wj = (b ? wb : wa).w[j >> 1] & m[j >> 1];
                        const uint32_t d = (j & 1) ? (wj >> 16) : (wj & 0xFFFFu);
                        if (d >= W) atomicAdd(&gt[b][d], 1u);
                    }
                }
            }
        }
#endif
    }
    __syncthreads();
    publish();
#ifdef SARPRO_PIECE_WG_TIMES
    __syncthreads();
    if (threadIdx.x == 0) g_piece_wg_times[blockIdx.x & 1023][1] = wall_clock64();
#endif
}

} // namespace
#ifdef SARPRO_PIECE_WG_TIMES
extern "C" int sarpro_hip_debug_piece_wg_times(unsigned long long *out /* [1024][2] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sarpro::g_piece_wg_times), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_dn_hist_pieces(const DnHistPiecesArgs &a, int grid, hipStream_t s) {
    if (grid <= 0 || grid > kPieceMaxGrid) return hipErrorInvalidValue;
    const size_t lds = (2 * ((size_t)a.lds_bins + 64) + 2 * (size_t)kLowWords + 1 + kTailCap) * sizeof(uint32_t);
    if (lds > 160 * 1024 || a.lds_bins < kLowBins || (a.lds_bins & (a.lds_bins - 1)) != 0) return hipErrorInvalidValue; // (a power of two: the tail test is a mask)
    if (hipError_t e = opt_in_dynamic_lds(reinterpret_cast<const void *>(k_dn_hist_pieces))) return e;
    hipLaunchKernelGGL(k_dn_hist_pieces, dim3(grid), dim3(kPBlock), lds, s, a);
    return hipGetLastError();
}

} // namespace sarpro

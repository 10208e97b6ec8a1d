// device_cdf.h -- clip / redistribute / CDF of one CLAHE tile (autoscale.rs:271-302) as a block-level device function, shared by
// k_chain_cdfs (chain_kernels.hip) and the tile-bin kernel that computes the CDFs in the same launch (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "host_logic.h"

namespace sarpro {

// Every thread of the block must call it (it holds barriers); threads with active == true and b in [0, 256) carry bin b of `tile`
// in hv and get its CDF value back.  scr / cum: 256-entry LDS scratch arrays.  All sums are exact (the clip threshold is a
// multiple of 2^-7) and order-free.
__device__ inline double clahe_tile_cdf_entry(unsigned long long hv, int tile, int b, bool active, uint32_t rows, uint32_t cols, double *scr,
                                              unsigned long long *cum) {
    const uint32_t tile_h = (rows + kTiles - 1) / kTiles, tile_w = (cols + kTiles - 1) / kTiles;
    const uint32_t ty = tile / kTiles, tx = tile % kTiles;
    const uint32_t r0 = min(ty * tile_h, rows), r1 = min((ty + 1) * tile_h, rows);
    const uint32_t c0 = min(tx * tile_w, cols), c1 = min((tx + 1) * tile_w, cols);
    const double avg = (double)((unsigned long long)(r1 - r0) * (unsigned long long)(c1 - c0)) / 256.0;
    const double thr = fmax(kClipLimit * avg, 1.0);
    double ex = 0.0;
    if (active && (double)hv > thr) { ex = (double)hv - thr; hv = (unsigned long long)(uint32_t)thr; } // `as u32` truncates (thr < 2^32 here)
    if (active) scr[b] = ex;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (active && b < s) scr[b] += scr[b + s]; __syncthreads(); }
    const double excess = scr[0];
    __syncthreads();
    const double add = floor(excess / 256.0);
    const unsigned long long remainder = (unsigned long long)round(excess - add * 256.0);
    const double hd = (double)hv + add;
    unsigned long long hn = hd >= 4294967295.0 ? 4294967295ull : (unsigned long long)hd; // `as u32` saturates
    hn += remainder / 256 + ((unsigned long long)b < remainder % 256 ? 1 : 0);         // round-robin from bin 0
    if (active) cum[b] = hn;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long v = (active && b >= off) ? cum[b - off] : 0ull;
        __syncthreads();
        if (active) cum[b] += v;
        __syncthreads();
    }
    const double total = fmax((double)cum[255], 1.0);
    const double c = active ? (double)cum[b] / total : 0.0;
    return c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
}

} // namespace sarpro

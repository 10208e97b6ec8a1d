// chain_kernels.h -- device-side "host half" of the CLAHE u8 chain (chain_kernels.hip).
#pragma once
#include "kernels.h"

namespace sarpro {

struct ChainBandState { // lives in device memory, one per band
    sarpro_hip_stats stats;
    uint32_t win_hi; // first DN whose dB value reached the high clip (the bin table is constant above)
    uint32_t pad;
};

struct ChainStatsArgs {
    const unsigned long long *ghist; // [nbands][65536]
    const double *db;                // [65536] dB value of every DN (host-built, glibc)
    ChainBandState *state;           // [nbands]
    uint8_t *binlut;                 // per band, binlut_stride bytes apart
    size_t binlut_stride;
};

struct ChainFinishArgs {
    const unsigned long long *level_hist; // [nbands][256]; bin 0 is restored here
    unsigned long long total_px;          // pixels per band of the whole scene
    int nbands;
    uint8_t *resc_out;                    // [2][256] u8 rescale maps
    uint8_t *identity_out;                // [2] 1 when the band's rescale is the identity on the occupied levels
    uint8_t *tables;                      // compose tables R2|G2|B2 (nbands == 2), may be null
    const uint8_t *supp_rg;               // [41][512] suppressed lut_r|lut_g for every floor value
    const uint8_t *blue_pair_supp;        // [256][256]
    int *floor_out;                       // optional
};

hipError_t launch_chain_stats(const ChainStatsArgs &a, int nbands, hipStream_t s);
hipError_t launch_chain_cdfs(const unsigned long long *tile_bins, double *cdfs, uint32_t rows, uint32_t cols, int nbands,
                             hipStream_t s);
hipError_t launch_chain_finish(const ChainFinishArgs &a, hipStream_t s);
// dst = map[src] unless skip_flag && *skip_flag (device byte) is non-zero and src == dst
hipError_t launch_chain_remap(const uint8_t *src, size_t src_pitch, uint8_t *dst, size_t dst_pitch, uint32_t rows,
                              uint32_t cols, const uint8_t *d_map, const uint8_t *d_skip_flag, hipStream_t s);

} // namespace sarpro

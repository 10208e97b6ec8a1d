// piece_kernels.h -- the scene cut into cost-balanced pieces for persistent workgroups, and the DN-histogram pass over them
// (piece_kernels.hip).
#pragma once
#include "chain_kernels.h"

namespace sarpro {

// One piece of a workgroup's share of the scene: a strip of `1 << gx_log2` wave columns (256 px each) inside one
// interpolation cell x a row range.  The 16 waves of the workgroup stand gx wide x 16/gx tall on it.
struct PieceItem {
    int32_t r0, r1;   // local rows [r0, r1)
    int32_t c0, c1;   // columns this piece owns
    int32_t cstart;   // column of lane 0 of wave column 0: a multiple of kPieceVec, <= c0
    int32_t gx_log2;
    int32_t flags;    // bit 0: the cell extrapolates (negative blend weights)
    int32_t id[4];    // tiles t00, t01, t10, t11
    int32_t tile;     // the tile this piece lies in
};
static_assert(sizeof(PieceItem) == 48, "PieceItem layout");
constexpr int kPieceMaxGrid = 1024;
// threads of a workgroup of the piece histogram (1024: sixteen waves, one workgroup per CU; 512: eight waves, so that another
// workgroup -- the fused pass of the previous scene -- fits beside it) and the DN range it keeps in LDS
#ifndef SARPRO_PIECE_BLOCK
#define SARPRO_PIECE_BLOCK 1024
#endif
#ifndef SARPRO_PIECE_BINS
#define SARPRO_PIECE_BINS 8192
#endif
constexpr int kPieceBlock = SARPRO_PIECE_BLOCK, kPieceWaves = kPieceBlock / 64;
constexpr int kPieceWavesLog2 = kPieceWaves == 16 ? 4 : 3;
static_assert(kPieceWaves == 16 || kPieceWaves == 8, "piece histogram: 1024 or 512 threads");
constexpr uint32_t kPieceLdsBins = SARPRO_PIECE_BINS;
// samples per lane and band-row of the piece histogram; a wave column is 64 x kPieceVec px wide.  4 (8-byte loads) is the default:
// the read-only piece traversal streams faster with 16 bytes per lane (6.1-6.3 TB/s against 5.6: profiles/r2/stream_bench.txt), but
// the histogram pass with 8 samples per lane -- sixteen LDS adds per lane and row behind two loads -- ran at 0.39 ms against 0.32
// (round 4, one call, alternating builds: profiles/r4/variants_piece_vec.txt).  -DSARPRO_PIECE_VEC=8 (api.cpp and piece_kernels.hip).
#ifndef SARPRO_PIECE_VEC
#define SARPRO_PIECE_VEC 4
#endif
constexpr int kPieceVec = SARPRO_PIECE_VEC;
constexpr int kPieceChunk = 64 * kPieceVec;
static_assert(kPieceVec == 4 || kPieceVec == 8, "piece histogram: 8- or 16-byte loads");

struct DnHistPiecesArgs { // per-tile DN histograms of both bands (k_dn_hist_pieces)
    const uint16_t *in[kMaxBands];
    uint32_t *tile_hist[kMaxBands]; // [64][65536], zeroed by the caller
    size_t pitch;                   // elements, % 4 == 0
    const PieceItem *items;
    const int32_t *wg_first;        // [grid + 1]
    uint32_t lds_bins;              // DN < lds_bins are privatised in LDS
};
hipError_t launch_dn_hist_pieces(const DnHistPiecesArgs &a, int grid, hipStream_t s);

} // namespace sarpro

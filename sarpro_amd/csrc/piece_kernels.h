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
    int32_t cstart;   // column of lane 0 of wave column 0: a multiple of 4, <= c0
    int32_t gx_log2;
    int32_t flags;    // bit 0: the cell extrapolates (negative blend weights)
    int32_t id[4];    // tiles t00, t01, t10, t11
    int32_t tile;     // the tile this piece lies in
};
static_assert(sizeof(PieceItem) == 48, "PieceItem layout");
constexpr int kPieceMaxGrid = 1024;

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

// resize_kernels.h -- launch interface of the Lanczos3 resample passes (resize_kernels.hip).
#pragma once
#include "kernels.h"

namespace sarpro {

struct ResizePassArgs {
    const void *src;
    void *dst;
    size_t src_pitch, dst_pitch; // elements
    uint32_t in_size, out_size;  // along the resampled axis
    uint32_t width;              // vertical pass: number of columns
    int precision;
    uint32_t max_val;            // 255 or 65535
    const uint32_t *start, *size; // [out_size]
    const int32_t *k;             // [window][out_size]
    uint32_t window;              // largest tap count
    uint32_t block_span;          // horizontal pass: max over groups of kResizeHBlock consecutive outputs of start[last] - start[first]
    uint32_t k_small;             // every coefficient < 32640: k = 256 hi + lo with both halves signed bytes (else the horizontal pass's BIG form)
    uint32_t generic;             // context attribute RESIZE_GENERIC: the tap-by-tap kernels (cross-check twin of the register-resident forms)
    // vertical pass over a WINDOW of the axis (a row stripe's share, stripe_resized.cpp): output rows [oy0, oy0 + oy_n) (oy_n = 0: all);
    // source row y is at src + (y - src_row0) * src_pitch, output row oy at dst + (oy - dst_row0) * dst_pitch
    uint32_t oy0, oy_n;
    int32_t src_row0, dst_row0;
};
constexpr uint32_t kResizeHBlock = 256;
constexpr uint32_t kResizeHMaxChunks = 16; // 8-byte chunks of taps a thread of the register-resident horizontal pass can hold (windows up to 114 taps)
// does the register-resident horizontal pass take this (block span, window, LDS bytes reserved for a DN table)?  (the launcher's own test: callers probe with it before they enqueue anything)
bool resize_h_dot_fits(uint32_t block_span, uint32_t window, bool src16, size_t lut_cap);

hipError_t launch_resize_h(const ResizePassArgs &a, uint32_t rows, int elem_size, hipStream_t s);
// The horizontal u8 pass reading u16 DN through the band's DN -> u8 table (the percentile strategies' whole autoscale is that table,
// kernels.hip 5): level = DN ? table[min(DN, win_hi)] : 0 while the row windows are staged, so the u8 level raster is neither
// written nor read.  a.src = the DN raster, a.src_pitch in u16 elements.  Returns hipErrorNotSupported when the register-resident
// form does not apply (window too wide, unaligned raster): the caller materialises the level raster and takes launch_resize_h.
struct ResizeLutSrc {
    const uint8_t *lut;                     // 65536 final u8 values
    const struct ChainBandState *dev_state; // [band].win_hi: the table is constant from there on
    int band;
    uint32_t lut_cap;                       // the window is staged in LDS when win_hi < lut_cap (bytes of LDS reserved for it)
};
hipError_t launch_resize_h_lut(const ResizePassArgs &a, const ResizeLutSrc &l, uint32_t rows, hipStream_t s);
hipError_t launch_resize_v(const ResizePassArgs &a, int elem_size, hipStream_t s);
constexpr size_t kResizeRowLdsMax = 160 * 1024; // a source row up to this size is staged in LDS; longer rows read their taps from memory

} // namespace sarpro

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
};
constexpr uint32_t kResizeHBlock = 256;

hipError_t launch_resize_h(const ResizePassArgs &a, uint32_t rows, int elem_size, hipStream_t s);
hipError_t launch_resize_v(const ResizePassArgs &a, int elem_size, hipStream_t s);
constexpr size_t kResizeRowLdsMax = 160 * 1024; // a source row up to this size is staged in LDS; longer rows read their taps from memory

} // namespace sarpro

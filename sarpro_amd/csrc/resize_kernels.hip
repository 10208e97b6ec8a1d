// resize_kernels.hip -- separable Lanczos3 resample of u8 / u16 rasters with host-built fixed-point
// coefficient tables (host_logic.cpp: build_resize_coeffs), restating the convolution of the
// `fast_image_resize` crate the reference calls (resize.rs:32-89).  Horizontal pass into an integer
// intermediate, then vertical pass.  Both passes are bound by reading their input once from HBM.
#include "resize_kernels.h"
#include "kernels.h"

namespace sarpro {
namespace {

constexpr int kBlock = 256;

// Horizontal pass: one block per source row; the row is staged in LDS, then each thread produces
// output pixels ox = t, t+256, ... (coefficients tap-major: lanes over ox read consecutive words).
// STAGE = false: rows too long for LDS (the reference's resize takes any width, resize.rs:32-89) read the taps from memory.
template <typename T, typename Acc, bool STAGE>
__global__ __launch_bounds__(kBlock) void k_resize_h(ResizePassArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const uint32_t r = blockIdx.x;
    const T *__restrict__ src = reinterpret_cast<const T *>(a.src) + (size_t)r * a.src_pitch;
    const T *row = src;
    if (STAGE) {
        T *lrow = reinterpret_cast<T *>(lds_raw);
        for (uint32_t x = threadIdx.x; x < a.in_size; x += kBlock) lrow[x] = src[x];
        __syncthreads();
        row = lrow;
    }
    T *__restrict__ dst = reinterpret_cast<T *>(a.dst) + (size_t)r * a.dst_pitch;
    const Acc initial = a.precision > 0 ? (Acc)1 << (a.precision - 1) : 0;
    for (uint32_t ox = threadIdx.x; ox < a.out_size; ox += kBlock) {
        const uint32_t x0 = a.start[ox], n = a.size[ox];
        Acc ss = initial;
        for (uint32_t t = 0; t < n; ++t) ss += (Acc)row[x0 + t] * (Acc)a.k[(size_t)t * a.out_size + ox];
        Acc o = ss >> a.precision;
        o = o < 0 ? 0 : (o > (Acc)a.max_val ? (Acc)a.max_val : o);
        dst[ox] = (T)o;
    }
}

// Vertical pass: thread per output pixel; lanes run along x, so every tap is a coalesced row read.
template <typename T, typename Acc>
__global__ __launch_bounds__(kBlock) void k_resize_v(ResizePassArgs a) {
    const uint32_t oy = blockIdx.y;
    const uint32_t x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= a.width) return;
    const T *__restrict__ src = reinterpret_cast<const T *>(a.src);
    const uint32_t y0 = a.start[oy], n = a.size[oy];
    Acc ss = a.precision > 0 ? (Acc)1 << (a.precision - 1) : 0;
    for (uint32_t t = 0; t < n; ++t)
        ss += (Acc)src[(size_t)(y0 + t) * a.src_pitch + x] * (Acc)a.k[(size_t)t * a.out_size + oy];
    Acc o = ss >> a.precision;
    o = o < 0 ? 0 : (o > (Acc)a.max_val ? (Acc)a.max_val : o);
    reinterpret_cast<T *>(a.dst)[(size_t)oy * a.dst_pitch + x] = (T)o;
}

} // namespace

hipError_t launch_resize_h(const ResizePassArgs &a, uint32_t rows, int elem_size, hipStream_t s) {
    if (!rows || !a.out_size) return hipSuccess;
    const size_t lds = ((size_t)a.in_size * elem_size + 15) & ~(size_t)15;
    if (lds > kResizeRowLdsMax) {
        if (elem_size == 1) hipLaunchKernelGGL((k_resize_h<uint8_t, int32_t, false>), dim3(rows), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_resize_h<uint16_t, long long, false>), dim3(rows), dim3(kBlock), 0, s, a);
        return hipGetLastError();
    }
    const void *fn = elem_size == 1 ? reinterpret_cast<const void *>(k_resize_h<uint8_t, int32_t, true>) : reinterpret_cast<const void *>(k_resize_h<uint16_t, long long, true>);
    if (lds > 64 * 1024)
        if (hipError_t e = opt_in_dynamic_lds(fn)) return e;
    if (elem_size == 1) hipLaunchKernelGGL((k_resize_h<uint8_t, int32_t, true>), dim3(rows), dim3(kBlock), lds, s, a);
    else hipLaunchKernelGGL((k_resize_h<uint16_t, long long, true>), dim3(rows), dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_resize_v(const ResizePassArgs &a, int elem_size, hipStream_t s) {
    if (!a.width || !a.out_size) return hipSuccess;
    dim3 grid((a.width + kBlock - 1) / kBlock, a.out_size);
    if (elem_size == 1) hipLaunchKernelGGL((k_resize_v<uint8_t, int32_t>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((k_resize_v<uint16_t, long long>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

} // namespace sarpro

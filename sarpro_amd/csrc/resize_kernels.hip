// resize_kernels.hip -- separable Lanczos3 resample of u8 / u16 rasters with host-built fixed-point
// coefficient tables (host_logic.cpp: build_resize_coeffs), restating the convolution of the
// `fast_image_resize` crate the reference calls (resize.rs:32-89).  Horizontal pass into an integer
// intermediate, then vertical pass.  Both passes are bound by reading their input once from HBM.
#include "resize_kernels.h"
#include "kernels.h"
#include "chain_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace sarpro {
namespace {

constexpr int kBlock = 256;
#ifndef SARPRO_RESIZE_H_ROWS
#define SARPRO_RESIZE_H_ROWS 2
#endif
constexpr int kResizeHRows = SARPRO_RESIZE_H_ROWS;  // rows per step of the register-resident horizontal pass
constexpr int kResizeHVecs = SARPRO_RESIZE_H_ROWS;  // 16-byte vectors a thread stages per step: kResizeHRows * span <= kResizeHVecs * 256 * 16 bytes

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

// Horizontal pass, u8, the form the 400 MP -> 2048^2 products take (resize.rs:32-89 on a level raster): a thread OWNS one
// output column and walks down the rows; its taps never change, so their coefficients live in registers for the whole
// kernel -- packed the way the pixels arrive.  Per row the block stages the input window of its 256 outputs in LDS (16-byte
// coalesced loads, double-buffered), and a thread reads the NCHUNK 16-byte chunks that cover its taps.  The i16-range
// coefficient k of the byte at chunk position p is split into k = 256 hi + lo (hi signed, lo unsigned byte), both stored at byte
// p of their packed words, zero outside the thread's taps; then
//     sum px k  =  sum px lo  +  256 (sum (px - 128) hi  +  128 sum hi)
// is two 4-way byte dot products per dword (v_dot4_u32_u8, v_dot4_i32_i8 on px ^ 0x80) instead of four extracts and four
// multiply-adds, and every term is an exact integer well inside i32 (|sum px k| < 2^31 is the crate's own guarantee), so the
// result equals the tap-by-tap i32 sum of the generic kernel bit for bit.  Algorithmic traffic: the band once (1 B/px) in,
// out_size / in_size of it out.
// SRC16: the rows are u16 DN and become u8 levels through the band's table while they are staged (ResizeLutSrc); the table's window
// sits in LDS ahead of the row buffers.
template <int NCHUNK, bool SRC16>
__global__ __launch_bounds__(kResizeHBlock) void k_resize_h_u8_dot(ResizePassArgs a, uint32_t rows, uint32_t span_bytes /* LDS bytes per staged row */, ResizeLutSrc lsrc) {
    extern __shared__ __align__(16) unsigned char lds_all[];
    const uint32_t win_hi = SRC16 ? lsrc.dev_state[lsrc.band].win_hi : 0u;
    const bool lut_lds = SRC16 && win_hi < lsrc.lut_cap;
    unsigned char *lds_raw = lds_all + (SRC16 ? lsrc.lut_cap : 0u); // (lut_cap is a multiple of 16)
    if (lut_lds) {
        for (uint32_t i = threadIdx.x; i <= win_hi; i += kResizeHBlock) lds_all[i] = i ? lsrc.lut[i] : (uint8_t)0; // (DN = 0 is invalid: level 0 whatever the table says)
        __syncthreads();
    }
    constexpr int R = kResizeHRows; // rows per step: one barrier per R rows, R independent sums per thread
    const uint32_t first = blockIdx.x * kResizeHBlock;
    const uint32_t ox = first + threadIdx.x;
    const bool valid = ox < a.out_size;
    const uint32_t oxc = valid ? ox : a.out_size - 1;
    const uint32_t x0 = a.start[oxc], n = a.size[oxc];
    const uint32_t bx0 = a.start[first] & ~15u;   // the block's window starts here (start[] is non-decreasing)
    const uint32_t xb = x0 & ~15u, lead = x0 - xb; // this thread's first chunk, and where its first tap sits in it
    uint32_t lo[NCHUNK * 4], hi[NCHUNK * 4];
    int32_t sumhi = 0;
#pragma unroll
    for (int w = 0; w < NCHUNK * 4; ++w) {
        uint32_t l = 0, h = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int32_t t = (int32_t)(w * 4 + b) - (int32_t)lead;
            const int32_t k = (valid && t >= 0 && (uint32_t)t < n) ? a.k[(size_t)t * a.out_size + ox] : 0;
            l |= (uint32_t)(k & 0xFF) << (8 * b);
            h |= (uint32_t)((k >> 8) & 0xFF) << (8 * b);
            sumhi += k >> 8;
        }
        lo[w] = l; hi[w] = h;
    }
    const int32_t base = (a.precision > 0 ? (int32_t)1 << (a.precision - 1) : 0) + 256 * 128 * sumhi;
    const uint32_t loff = xb - bx0; // byte offset of this thread's first chunk in a staged row
    const uint8_t *__restrict__ src = reinterpret_cast<const uint8_t *>(a.src);
    uint8_t *__restrict__ dst = reinterpret_cast<uint8_t *>(a.dst);
    // staging: the step's R row windows are nvec 16-byte vectors each; thread t carries vectors t, t + 256, ... (kResizeHVecs at most)
    const uint32_t nvec = span_bytes / 16, ntot = nvec * R;
    const size_t row_bytes = a.src_pitch; // the window may reach past the row's pitch at the right edge (zero coefficients there): clamped
    constexpr int QW = SRC16 ? 2 : 1; // 16 staged pixels are 16 bytes of levels or 32 bytes of DN
    auto fetch = [&](uint32_t r0, uint4 (&q)[kResizeHVecs * QW]) {
#pragma unroll
        for (int i = 0; i < kResizeHVecs; ++i) {
            const uint32_t v = threadIdx.x + (uint32_t)i * kResizeHBlock;
#pragma unroll
            for (int w = 0; w < QW; ++w) q[i * QW + w] = make_uint4(0, 0, 0, 0);
            if (v < ntot) {
                const uint32_t rr = v / nvec, vv = v - rr * nvec;
                const uint32_t r = min(r0 + rr, rows - 1);
                const size_t off = (size_t)bx0 + (size_t)vv * 16; // in pixels
                if (off < row_bytes) { // off and the pitch are multiples of 16 pixels: never partial
                    if (SRC16) {
                        const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.src) + (size_t)r * row_bytes + off);
                        q[i * QW] = p[0]; q[i * QW + QW - 1] = p[1];
                    } else q[i * QW] = *reinterpret_cast<const uint4 *>(src + (size_t)r * row_bytes + off);
                }
            }
        }
    };
    // DN pair -> two level bytes (kernels.hip 5: DN ? table[min(DN, win_hi)] : 0)
    const uint32_t hi2 = win_hi | (win_hi << 16);
    auto levels2 = [&](uint32_t w) -> uint32_t {
        typedef unsigned short v2us __attribute__((ext_vector_type(2)));
        const uint32_t c = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2us, w), __builtin_bit_cast(v2us, hi2))); // both DNs clamped at once
        const uint32_t c0 = c & 0xFFFFu, c1 = c >> 16;
        if (lut_lds) return (uint32_t)lds_all[c0] | ((uint32_t)lds_all[c1] << 8); // (entry 0 of the staged copy is 0)
        return (c0 ? (uint32_t)lsrc.lut[c0] : 0u) | ((c1 ? (uint32_t)lsrc.lut[c1] : 0u) << 8);
    };
    auto put = [&](const uint4 (&q)[kResizeHVecs * QW], unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < kResizeHVecs; ++i) {
            const uint32_t v = threadIdx.x + (uint32_t)i * kResizeHBlock;
            if (v < ntot) {
                uint4 o = q[i * QW];
                if (SRC16) {
                    const uint4 lo4 = q[i * QW], hi4 = q[i * QW + QW - 1];
                    o.x = levels2(lo4.x) | (levels2(lo4.y) << 16); o.y = levels2(lo4.z) | (levels2(lo4.w) << 16);
                    o.z = levels2(hi4.x) | (levels2(hi4.y) << 16); o.w = levels2(hi4.z) | (levels2(hi4.w) << 16);
                }
                *reinterpret_cast<uint4 *>(buf + (size_t)v * 16) = o;
            }
        }
    };
    const uint32_t step = gridDim.y * R;
    uint32_t r = blockIdx.y * R;
    if (r >= rows) return;
    const size_t buf_bytes = (size_t)span_bytes * R;
    int cur = 0;
    uint4 q[kResizeHVecs * QW];
    fetch(r, q);
    put(q, lds_raw);
    __syncthreads();
    for (; r < rows; r += step) {
        const bool more = r + step < rows;
        if (more) fetch(r + step, q); // the next step's rows are in flight (registers) while this step is summed
        const unsigned char *buf = lds_raw + (size_t)cur * buf_bytes + loff;
        int32_t acc_hi[R];
        uint32_t acc_lo[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { acc_hi[j] = 0; acc_lo[j] = 0; }
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint4 px4 = *reinterpret_cast<const uint4 *>(buf + (size_t)j * span_bytes + c * 16);
                const uint32_t px[4] = {px4.x, px4.y, px4.z, px4.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    acc_lo[j] = __builtin_amdgcn_udot4(px[w], lo[c * 4 + w], acc_lo[j], false);
                    acc_hi[j] = __builtin_amdgcn_sdot4((int32_t)(px[w] ^ 0x80808080u), (int32_t)hi[c * 4 + w], acc_hi[j], false);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            int32_t o = (base + (int32_t)acc_lo[j] + 256 * acc_hi[j]) >> a.precision;
            o = o < 0 ? 0 : (o > 255 ? 255 : o);
            if (valid && r + j < rows) dst[(size_t)(r + j) * a.dst_pitch + ox] = (uint8_t)o;
        }
        if (more) put(q, lds_raw + (size_t)(cur ^ 1) * buf_bytes);
        __syncthreads(); // the next step is staged, and nobody reads `cur` any more
        cur ^= 1;
    }
}

// Vertical pass, u8: a thread owns 8 neighbouring columns of one output row; every tap is one 8-byte load (a wave reads 512
// contiguous bytes of an intermediate row), the tap's coefficient is block-uniform (scalar).  Exact i32 sums, as the generic
// kernel's.
__global__ __launch_bounds__(256) void k_resize_v_u8_x8(ResizePassArgs a) {
    const uint32_t oy = blockIdx.y;
    const uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (x >= a.width) return;
    const uint8_t *__restrict__ src = reinterpret_cast<const uint8_t *>(a.src);
    const uint32_t y0 = a.start[oy], n = a.size[oy];
    const int32_t initial = a.precision > 0 ? (int32_t)1 << (a.precision - 1) : 0;
    int32_t ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ss[j] = initial;
    const bool full = x + 8 <= a.width && a.src_pitch % 8 == 0;
    for (uint32_t t = 0; t < n; ++t) {
        const int32_t k = a.k[(size_t)t * a.out_size + oy];
        const uint8_t *p = src + (size_t)(y0 + t) * a.src_pitch + x;
        uint32_t w0 = 0, w1 = 0;
        if (full) { const uint2 q = *reinterpret_cast<const uint2 *>(p); w0 = q.x; w1 = q.y; }
        else
            for (uint32_t j = 0; j < 8 && x + j < a.width; ++j) { if (j < 4) w0 |= (uint32_t)p[j] << (8 * j); else w1 |= (uint32_t)p[j] << (8 * (j - 4)); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ss[j] += (int32_t)((w0 >> (8 * j)) & 0xFFu) * k;
            ss[4 + j] += (int32_t)((w1 >> (8 * j)) & 0xFFu) * k;
        }
    }
    uint8_t *d = reinterpret_cast<uint8_t *>(a.dst) + (size_t)oy * a.dst_pitch + x;
    uint32_t o0 = 0, o1 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int32_t o = ss[j] >> a.precision;
        o = o < 0 ? 0 : (o > 255 ? 255 : o);
        if (j < 4) o0 |= (uint32_t)o << (8 * j); else o1 |= (uint32_t)o << (8 * (j - 4));
    }
    if (x + 8 <= a.width && ((reinterpret_cast<uintptr_t>(d) & 7) == 0)) *reinterpret_cast<uint2 *>(d) = make_uint2(o0, o1);
    else
        for (uint32_t j = 0; j < 8 && x + j < a.width; ++j) d[j] = (uint8_t)((j < 4 ? o0 >> (8 * j) : o1 >> (8 * (j - 4))) & 0xFFu);
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

// the register-resident form: SRC16 = false reads u8 levels, true reads u16 DN through a table (lut_cap bytes of LDS ahead of the rows)
template <bool SRC16>
static hipError_t launch_resize_h_dot(const ResizePassArgs &a, const ResizeLutSrc &l, uint32_t rows, hipStream_t s) {
    if (!a.window) return hipErrorNotSupported;
    const uint32_t nchunk = (15 + a.window + 15) / 16;
    const uint32_t span = (a.block_span + 15 + nchunk * 16 + 15) / 16 * 16; // bytes of a row that one block's 256 outputs read
    if (!(nchunk <= 8 && (size_t)span * kResizeHRows <= (size_t)kResizeHVecs * kResizeHBlock * 16 && (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 &&
          a.src_pitch % 16 == 0))
        return hipErrorNotSupported;
    const uint32_t gx = (a.out_size + kResizeHBlock - 1) / kResizeHBlock;
    const uint32_t steps = (rows + kResizeHRows - 1) / kResizeHRows;
    const uint32_t blocks = 2048; // ~2048 blocks, each walks its share of the rows
    const uint32_t gy = std::min<uint32_t>(steps, std::max<uint32_t>(1, blocks / gx));
    const dim3 grid(gx, gy), block(kResizeHBlock);
    const size_t lds = (size_t)span * kResizeHRows * 2 + (SRC16 ? l.lut_cap : 0u);
    if (lds > 64 * 1024) return hipErrorNotSupported;
    switch (nchunk) {
    case 1: hipLaunchKernelGGL((k_resize_h_u8_dot<1, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    case 2: hipLaunchKernelGGL((k_resize_h_u8_dot<2, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    case 3: hipLaunchKernelGGL((k_resize_h_u8_dot<3, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    case 4: hipLaunchKernelGGL((k_resize_h_u8_dot<4, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    case 5: hipLaunchKernelGGL((k_resize_h_u8_dot<5, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    case 6: hipLaunchKernelGGL((k_resize_h_u8_dot<6, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    case 7: hipLaunchKernelGGL((k_resize_h_u8_dot<7, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    default: hipLaunchKernelGGL((k_resize_h_u8_dot<8, SRC16>), grid, block, lds, s, a, rows, span, l); break;
    }
    return hipGetLastError();
}

hipError_t launch_resize_h_lut(const ResizePassArgs &a, const ResizeLutSrc &l, uint32_t rows, hipStream_t s) {
    if (!rows || !a.out_size) return hipSuccess;
    if (!l.lut || !l.dev_state || l.lut_cap % 16 != 0 || a.generic) return hipErrorNotSupported;
    return launch_resize_h_dot<true>(a, l, rows, s);
}

hipError_t launch_resize_h(const ResizePassArgs &a, uint32_t rows, int elem_size, hipStream_t s) {
    if (!rows || !a.out_size) return hipSuccess;
    if (elem_size == 1 && a.window && !a.generic) { // the register-resident form where its window fits
        const hipError_t e = launch_resize_h_dot<false>(a, ResizeLutSrc{}, rows, s);
        if (e != hipErrorNotSupported) return e;
    }
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
    if (elem_size == 1 && !a.generic) {
        hipLaunchKernelGGL(k_resize_v_u8_x8, dim3((a.width + kBlock * 8 - 1) / (kBlock * 8), a.out_size), dim3(kBlock), 0, s, a);
        return hipGetLastError();
    }
    if (elem_size == 1) hipLaunchKernelGGL((k_resize_v<uint8_t, int32_t>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((k_resize_v<uint16_t, long long>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

} // namespace sarpro

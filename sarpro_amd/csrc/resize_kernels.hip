// resize_kernels.hip -- separable Lanczos3 resample of u8 / u16 rasters with host-built fixed-point
// coefficient tables (host_logic.cpp: build_resize_coeffs), restating the convolution of the
// `fast_image_resize` crate the reference calls (resize.rs:32-89).  Horizontal pass into an integer
// intermediate, then vertical pass.  Both passes are bound by reading their input once from HBM.
#include "resize_kernels.h"
#include "kernels.h"
#include "chain_kernels.h"

#include <algorithm>
#include <type_traits>
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
// coalesced loads, the step after next in flight), and a thread reads the NC8 8-byte chunks that cover its taps.
// Arithmetic (round 5: one third fewer vector instructions per output than the unsigned form it replaces).  The staged bytes are
// s = px - 128 as SIGNED bytes (px ^ 0x80, once per pixel while the row is staged -- a pixel is read by ~6 outputs); the i16-range
// coefficient k of the byte at chunk position p is split into k = 256 hi + lo with BOTH halves signed bytes (lo = the low byte
// sign-extended, hi = (k - lo) >> 8), stored at byte p of their packed words, zero outside the thread's taps.  A coefficient of
// 32640 or more (build_resize_coeffs scales the largest weight into [2^14, 2^15): a near-unity scale factor puts its centre tap
// there) has hi = 128, one more than a signed byte holds: such tables run the BIG instantiation, which keeps hi = 127 and a
// third packed word with a 1 at that tap (one more dot product per dword; the others never pay for it).  Then
//     sum px k  =  sum s lo  +  256 sum s hi  +  128 sum k
// is two signed 4-way byte dot products per dword (v_dot4_i32_i8) and a per-thread constant; every term is an exact integer well
// inside i32 (|sum px k| < 2^31 is the crate's own guarantee; |sum s lo|, |sum s hi| <= 128 x 128 x 128), so the result equals the
// tap-by-tap i32 sum of the generic kernel bit for bit.  Algorithmic traffic: the band once (1 B/px) in, out_size / in_size of it out.
// SRC16: the rows are u16 DN and become u8 levels through the band's table while they are staged (ResizeLutSrc); the table's window
// sits in LDS ahead of the row buffers.
#ifndef SARPRO_RESIZE_H_MINWAVES
#define SARPRO_RESIZE_H_MINWAVES 1
#endif
// Bytes of a tap chunk (one LDS read): 16 (ds_read_b128: 5 reads and 40 dot products for the 59 taps of 20000 -> 2048) or 8 (ds_read_b64:
// 9 reads, 36 dot products).  Measured both ways for both sources (profiles/r5/resize_variants.txt): the pass that turns DN into
// levels while it stages (SRC16) is bound by its LDS cycles -- the table gathers come on top of the window reads -- and is 3 % faster
// with the wide read (0.415 against 0.433 ms for two 400 MP bands); the pass over a u8 raster is 15 % faster with the narrow one
// (0.257 against 0.297 ms: four dot products less per output, and its lanes' 8-byte chunks meet on LDS banks less than they cost).
constexpr uint32_t rz_chunk(bool src16) { return src16 ? 16u : 8u; }
template <int NC8, bool SRC16, bool BIG>
__global__ __launch_bounds__(kResizeHBlock, SARPRO_RESIZE_H_MINWAVES) void k_resize_h_u8_dot(ResizePassArgs a, uint32_t rows, uint32_t span_bytes /* LDS bytes per staged row */, ResizeLutSrc lsrc) {
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
    constexpr uint32_t kRzChunk = rz_chunk(SRC16), kRzW = kRzChunk / 4; // bytes / dwords per chunk
    const uint32_t xb = x0 & ~(kRzChunk - 1u), lead = x0 - xb; // this thread's first chunk, and where its first tap sits in it
    uint32_t lo[NC8 * kRzW], hi[NC8 * kRzW], ex[BIG ? NC8 * kRzW : 1];
    int32_t sumk = 0;
#pragma unroll
    for (int w = 0; w < NC8 * (int)kRzW; ++w) {
        uint32_t l = 0, h = 0, e = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int32_t t = (int32_t)(w * 4 + b) - (int32_t)lead;
            const int32_t k = (valid && t >= 0 && (uint32_t)t < n) ? a.k[(size_t)t * a.out_size + ox] : 0;
            const int32_t kl = (int32_t)(int8_t)(k & 0xFF); // k = 256 kh + kl, kl in [-128, 127], kh in [-128, 128]
            int32_t kh = (k - kl) >> 8;
            if (BIG && kh > 127) { kh = 127; e |= 1u << (8 * b); } // (k >= 32640)
            l |= (uint32_t)(kl & 0xFF) << (8 * b);
            h |= (uint32_t)(kh & 0xFF) << (8 * b);
            sumk += k;
        }
        lo[w] = l; hi[w] = h;
        if (BIG) ex[w] = e;
    }
    const int32_t base = (a.precision > 0 ? (int32_t)1 << (a.precision - 1) : 0) + 128 * sumk;
    const uint32_t loff = xb - bx0; // byte offset of this thread's first chunk in a staged row (a multiple of 8)
    const uint8_t *__restrict__ src = reinterpret_cast<const uint8_t *>(a.src);
    uint8_t *__restrict__ dst = reinterpret_cast<uint8_t *>(a.dst);
    // staging: the step's R row windows are nvec 16-byte vectors each; thread t carries vectors t, t + 256, ... (kResizeHVecs at most)
    const uint32_t nvec = span_bytes / 16, ntot = nvec * R;
    const size_t row_bytes = a.src_pitch; // the window may reach past the row's pitch at the right edge (zero coefficients there): clamped
    constexpr int QW = SRC16 ? 2 : 1; // 16 staged pixels are 16 bytes of levels or 32 bytes of DN
    // The vectors this thread stages: which row of the step, where in the row (loop-invariant).  Loads and stores go through buffer
    // descriptors over the step's R rows: a lane with nothing to load or store passes an out-of-range offset -- the hardware returns
    // zeros / drops the write -- so that every step issues the SAME vector-memory instructions in every lane.  With the loads and
    // stores inside `if` blocks the compiler could not count the operations in flight and opened every step with s_waitcnt
    // vmcnt(0): the rows requested a moment earlier had to be back before anything was summed, one memory round trip per step
    // (2.6 us for two rows of a 256-output block: 2.4 TB/s whatever else was changed; profiles/r5/pmc_resize_h_before.txt).
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr uint32_t kOob = 0xFFFFFFFFu;
    const uint32_t in_row_bytes = (uint32_t)row_bytes * (SRC16 ? 2u : 1u);
    uint32_t v_boff[kResizeHVecs]; // byte offset of vector i inside the step's rows, or out of range
#pragma unroll
    for (int i = 0; i < kResizeHVecs; ++i) {
        const uint32_t v = threadIdx.x + (uint32_t)i * kResizeHBlock;
        const uint32_t rr = v / nvec, vv = v - rr * nvec;
        const uint32_t off_px = bx0 + vv * 16u; // off and the pitch are multiples of 16 pixels: never partial
        v_boff[i] = (v < ntot && (size_t)off_px < row_bytes) ? rr * in_row_bytes + off_px * (SRC16 ? 2u : 1u) : kOob;
    }
    auto fetch = [&](uint32_t r0, uint4 (&q)[kResizeHVecs * QW]) {
        const uint32_t rc = min(r0, rows);                        // (a step past the end: no row in range, every lane reads zeros)
        const uint32_t nrow = min((uint32_t)R, rows - rc);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)rc * in_row_bytes, 0, (int)(nrow * in_row_bytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < kResizeHVecs; ++i) {
            const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rs, v_boff[i], 0, 0);
            q[i * QW] = make_uint4(x.x, x.y, x.z, x.w);
            if (SRC16) {
                const v4u y = __builtin_amdgcn_raw_buffer_load_b128(rs, v_boff[i] == kOob ? kOob : v_boff[i] + 16u, 0, 0);
                q[i * QW + QW - 1] = make_uint4(y.x, y.y, y.z, y.w);
            }
        }
    };
    // DN pair -> two level bytes (kernels.hip 5: DN ? table[min(DN, win_hi)] : 0).  The table's place (LDS when its window fits, global
    // memory otherwise) is decided ONCE per workgroup: the row loop is compiled for either (as a per-sample test it put seventeen
    // branches and both table paths into every step).
    const uint32_t hi2 = win_hi | (win_hi << 16);
    const uint32_t step = gridDim.y * R;
    const uint32_t r_first = blockIdx.y * R;
    if (r_first >= rows) return;
    const size_t buf_bytes = (size_t)span_bytes * R;
    auto run = [&](auto lut_tag) {
        constexpr bool LUT_LDS = decltype(lut_tag)::value;
        auto levels2 = [&](uint32_t w) -> uint32_t {
            typedef unsigned short v2us __attribute__((ext_vector_type(2)));
            const uint32_t c = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(v2us, w), __builtin_bit_cast(v2us, hi2))); // both DNs clamped at once
            const uint32_t c0 = c & 0xFFFFu, c1 = c >> 16;
            if (LUT_LDS) return (uint32_t)lds_all[c0] | ((uint32_t)lds_all[c1] << 8); // (entry 0 of the staged copy is 0)
            const uint32_t a0 = lsrc.lut[c0], a1 = lsrc.lut[c1]; // unconditional loads, DN = 0 masked afterwards: no branch per sample
            return (a0 & (0u - (uint32_t)(c0 != 0u))) | ((a1 & (0u - (uint32_t)(c1 != 0u))) << 8);
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
                    o.x ^= 0x80808080u; o.y ^= 0x80808080u; o.z ^= 0x80808080u; o.w ^= 0x80808080u; // px -> px - 128 as signed bytes
                    *reinterpret_cast<uint4 *>(buf + (size_t)v * 16) = o;
                }
            }
        };
        // One step = R rows: sum the step staged in LDS, stage the next one, and have the step AFTER that in flight in registers --
        // a row window requested at the top of a step is converted and written to LDS at the bottom of the NEXT step, two step times
        // later.  The two register sets alternate by unrolling.
        auto sum_step = [&](uint32_t r, int cur) {
            const unsigned char *buf = lds_raw + (size_t)cur * buf_bytes + loff;
            int32_t acc_hi[R], acc_lo[R];
#pragma unroll
            for (int j = 0; j < R; ++j) { acc_hi[j] = 0; acc_lo[j] = 0; }
#pragma unroll
            for (int j = 0; j < R; ++j) {
#pragma unroll
                for (int c = 0; c < NC8; ++c) {
                    uint32_t px[kRzW];
                    if (kRzChunk == 16) {
                        const uint4 p4 = *reinterpret_cast<const uint4 *>(buf + (size_t)j * span_bytes + c * 16);
                        px[0] = p4.x; px[1] = p4.y; px[kRzW - 2] = p4.z; px[kRzW - 1] = p4.w;
                    } else {
                        const uint2 p2 = *reinterpret_cast<const uint2 *>(buf + (size_t)j * span_bytes + c * 8);
                        px[0] = p2.x; px[1] = p2.y;
                    }
#pragma unroll
                    for (int w = 0; w < (int)kRzW; ++w) {
                        acc_lo[j] = __builtin_amdgcn_sdot4((int32_t)px[w], (int32_t)lo[c * kRzW + w], acc_lo[j], false);
                        acc_hi[j] = __builtin_amdgcn_sdot4((int32_t)px[w], (int32_t)hi[c * kRzW + w], acc_hi[j], false);
                        if (BIG) acc_hi[j] = __builtin_amdgcn_sdot4((int32_t)px[w], (int32_t)ex[c * kRzW + w], acc_hi[j], false);
                    }
                }
            }
            const uint32_t nrow = min((uint32_t)R, rows - r);
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)r * a.dst_pitch, 0, (int)(nrow * (uint32_t)a.dst_pitch), 0x00020000);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                int32_t o = (base + acc_lo[j] + 256 * acc_hi[j]) >> a.precision;
                o = o < 0 ? 0 : (o > 255 ? 255 : o);
                // (rows past the end lie outside the descriptor; a lane without an output column passes an out-of-range offset)
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)o, rd, valid ? (uint32_t)j * (uint32_t)a.dst_pitch + ox : kOob, 0, 0);
            }
        };
        uint32_t r = r_first;
        uint4 qa[kResizeHVecs * QW], qb[kResizeHVecs * QW];
        fetch(r, qa);
        put(qa, lds_raw);
        fetch(r + step, qa); // (fetch clamps its rows: a step past the end reads the last row and is never staged)
        __syncthreads();
        for (; r < rows; r += 2 * step) {
            // even step: `r` is staged in buffer 0, r + step is in flight in qa, r + 2 step is requested into qb
            fetch(r + 2 * step, qb);
            sum_step(r, 0);
            if (r + step < rows) put(qa, lds_raw + buf_bytes);
            __syncthreads();
            if (r + step >= rows) break;
            // odd step: r + step is staged in buffer 1, r + 2 step is in flight in qb, r + 3 step is requested into qa
            fetch(r + 3 * step, qa);
            sum_step(r + step, 1);
            if (r + 2 * step < rows) put(qb, lds_raw);
            __syncthreads();
        }
    };
    if (lut_lds || !SRC16) run(std::true_type{});
    else run(std::false_type{});
}

// Vertical pass, u8: a thread owns 8 neighbouring columns of one output row; every tap is one 8-byte load (a wave reads 512
// contiguous bytes of an intermediate row), the tap's coefficient is block-uniform (scalar).  Exact i32 sums, as the generic
// kernel's.
__global__ __launch_bounds__(256) void k_resize_v_u8_x8(ResizePassArgs a) {
    const uint32_t oy = a.oy0 + blockIdx.y;
    const uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (x >= a.width) return;
    const uint8_t *__restrict__ src = reinterpret_cast<const uint8_t *>(a.src) - (ptrdiff_t)a.src_row0 * (ptrdiff_t)a.src_pitch;
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
    uint8_t *d = reinterpret_cast<uint8_t *>(a.dst) + ((ptrdiff_t)oy - (ptrdiff_t)a.dst_row0) * (ptrdiff_t)a.dst_pitch + x;
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
    const uint32_t oy = a.oy0 + blockIdx.y;
    const uint32_t x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= a.width) return;
    const T *__restrict__ src = reinterpret_cast<const T *>(a.src) - (ptrdiff_t)a.src_row0 * (ptrdiff_t)a.src_pitch;
    const uint32_t y0 = a.start[oy], n = a.size[oy];
    Acc ss = a.precision > 0 ? (Acc)1 << (a.precision - 1) : 0;
    for (uint32_t t = 0; t < n; ++t)
        ss += (Acc)src[(size_t)(y0 + t) * a.src_pitch + x] * (Acc)a.k[(size_t)t * a.out_size + oy];
    Acc o = ss >> a.precision;
    o = o < 0 ? 0 : (o > (Acc)a.max_val ? (Acc)a.max_val : o);
    reinterpret_cast<T *>(a.dst)[((ptrdiff_t)oy - (ptrdiff_t)a.dst_row0) * (ptrdiff_t)a.dst_pitch + x] = (T)o;
}

} // namespace

// the register-resident form: SRC16 = false reads u8 levels, true reads u16 DN through a table (lut_cap bytes of LDS ahead of the rows)
template <bool SRC16, int NC8>
static void launch_resize_h_dot_n(dim3 grid, size_t lds, hipStream_t s, const ResizePassArgs &a, uint32_t rows, uint32_t span, const ResizeLutSrc &l) {
    if constexpr (NC8 * rz_chunk(SRC16) <= kResizeHMaxChunks * 8) { // (a thread holds at most 128 bytes of taps: larger counts are never launched, nor compiled)
        if (a.k_small) hipLaunchKernelGGL((k_resize_h_u8_dot<NC8, SRC16, false>), grid, dim3(kResizeHBlock), lds, s, a, rows, span, l);
        else hipLaunchKernelGGL((k_resize_h_u8_dot<NC8, SRC16, true>), grid, dim3(kResizeHBlock), lds, s, a, rows, span, l);
    }
}
uint32_t resize_h_dot_chunks(uint32_t window, bool src16) { return (rz_chunk(src16) - 1 + window + rz_chunk(src16) - 1) / rz_chunk(src16); }
uint32_t resize_h_dot_span(uint32_t block_span, uint32_t window, bool src16) { return (block_span + 15 + resize_h_dot_chunks(window, src16) * rz_chunk(src16) + 15) / 16 * 16; } // bytes of a row that one block's 256 outputs read
bool resize_h_dot_fits(uint32_t block_span, uint32_t window, bool src16, size_t lut_cap) {
    const uint32_t nc8 = resize_h_dot_chunks(window, src16);
    const size_t span = resize_h_dot_span(block_span, window, src16);
    return window && nc8 <= kResizeHMaxChunks * 8 / rz_chunk(src16) && span * kResizeHRows <= (size_t)kResizeHVecs * kResizeHBlock * 16 && span * kResizeHRows * 2 + lut_cap <= 64 * 1024;
}
template <bool SRC16>
static hipError_t launch_resize_h_dot(const ResizePassArgs &a, const ResizeLutSrc &l, uint32_t rows, hipStream_t s) {
    if (!a.window) return hipErrorNotSupported;
    const uint32_t nc8 = resize_h_dot_chunks(a.window, SRC16);
    const uint32_t span = resize_h_dot_span(a.block_span, a.window, SRC16);
    if (!(resize_h_dot_fits(a.block_span, a.window, SRC16, SRC16 ? l.lut_cap : 0u) && (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 && a.src_pitch % 16 == 0))
        return hipErrorNotSupported;
    const uint32_t gx = (a.out_size + kResizeHBlock - 1) / kResizeHBlock;
    const uint32_t steps = (rows + kResizeHRows - 1) / kResizeHRows;
#ifndef SARPRO_RESIZE_H_BLOCKS
#define SARPRO_RESIZE_H_BLOCKS 2048
#endif
    const uint32_t blocks = SARPRO_RESIZE_H_BLOCKS; // ~2048 blocks, each walks its share of the rows
    const uint32_t gy = std::min<uint32_t>(steps, std::max<uint32_t>(1, blocks / gx));
    const dim3 grid(gx, gy);
    const size_t lds = (size_t)span * kResizeHRows * 2 + (SRC16 ? l.lut_cap : 0u);
    switch (nc8) {
#define SARPRO_RZ_CASE(N) case N: launch_resize_h_dot_n<SRC16, N>(grid, lds, s, a, rows, span, l); break;
    SARPRO_RZ_CASE(1) SARPRO_RZ_CASE(2) SARPRO_RZ_CASE(3) SARPRO_RZ_CASE(4) SARPRO_RZ_CASE(5) SARPRO_RZ_CASE(6) SARPRO_RZ_CASE(7) SARPRO_RZ_CASE(8)
    SARPRO_RZ_CASE(9) SARPRO_RZ_CASE(10) SARPRO_RZ_CASE(11) SARPRO_RZ_CASE(12) SARPRO_RZ_CASE(13) SARPRO_RZ_CASE(14) SARPRO_RZ_CASE(15)
#undef SARPRO_RZ_CASE
    default: launch_resize_h_dot_n<SRC16, 16>(grid, lds, s, a, rows, span, l); break;
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
    const uint32_t ny = a.oy_n ? a.oy_n : a.out_size - a.oy0; // (a window of the output rows: a row stripe's share)
    if (!a.width || !a.out_size || !ny) return hipSuccess;
    dim3 grid((a.width + kBlock - 1) / kBlock, ny);
    if (elem_size == 1 && !a.generic) {
        hipLaunchKernelGGL(k_resize_v_u8_x8, dim3((a.width + kBlock * 8 - 1) / (kBlock * 8), ny), dim3(kBlock), 0, s, a);
        return hipGetLastError();
    }
    if (elem_size == 1) hipLaunchKernelGGL((k_resize_v<uint8_t, int32_t>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((k_resize_v<uint16_t, long long>), grid, dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

} // namespace sarpro

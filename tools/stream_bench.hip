// stream_bench.hip -- micro-benchmark behind the fused pass's decomposition (tools/, not part of the library):
// persistent workgroups of 1024 threads stream two u16 rasters (and optionally write an RGB raster) in pieces of a
// given shape; reports achieved TB/s.  Variants: bytes per lane per load, rows dealt contiguously or block-cyclic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args { const unsigned short *in0, *in1; unsigned char *rgb; size_t pitch; int rows, cols; int nwg; int hb; int store; unsigned *sink; int valu; };

// VPX pixels per lane (4 -> 8-B loads, 8 -> 16-B loads).  A workgroup = 16 waves; strip = GX waves across.
template <int VPX, int GX, int CYCLIC>
__global__ __launch_bounds__(1024) void k_stream(Args a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int GY = 16 / GX;
    const int wx = wave % GX, wy = wave / GX;
    const int strip_w = GX * 64 * VPX;
    const int nstrips = a.cols / strip_w;          // whole strips only
    const int wg_per_strip = a.nwg / nstrips;      // workgroups that share a strip
    const int strip = blockIdx.x / wg_per_strip, ph = blockIdx.x % wg_per_strip;
    if (strip >= nstrips) return;
    const int col = strip * strip_w + (wx * 64 + lane) * VPX;
    unsigned acc = 0;
    auto do_row = [&](int r) {
        const unsigned short *p0 = a.in0 + (size_t)r * a.pitch + col, *p1 = a.in1 + (size_t)r * a.pitch + col;
        if (VPX == 4) { uint2 x = *(const uint2 *)p0, y = *(const uint2 *)p1; acc += x.x ^ x.y ^ y.x ^ y.y; }
        else { uint4 x = *(const uint4 *)p0, y = *(const uint4 *)p1; acc += x.x ^ x.y ^ x.z ^ x.w ^ y.x ^ y.y ^ y.z ^ y.w; }
        for (int v = 0; v < a.valu; ++v) { // a.valu x 8 dependent-free vector instructions per row
            float f0 = __builtin_bit_cast(float, acc), f1 = f0 + 1.0f, f2 = f0 + 2.0f, f3 = f0 + 3.0f;
            asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %3, %3, %0, %3\n\tv_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %3, %3, %0, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
            acc += __builtin_bit_cast(unsigned, f0 + f1 + f2 + f3);
        }
        if (a.store) {
            unsigned char *o = a.rgb + ((size_t)r * a.pitch + col) * 3;
            struct __attribute__((packed, aligned(4))) U3 { unsigned x, y, z; };
            if (VPX == 4 && a.store == 2) { // the same 768 bytes of the wave as 48 lanes x 16 B (what a cross-lane transpose would feed)
                if (lane < 48) *(uint4 *)(a.rgb + ((size_t)r * a.pitch + (col - lane * VPX)) * 3 + lane * 16) = make_uint4(acc, acc + 1, acc + 2, acc + 3);
            } else if (VPX == 4) *(U3 *)o = U3{acc, acc + 1, acc + 2};
            else { *(U3 *)o = U3{acc, acc + 1, acc + 2}; *(U3 *)(o + 12) = U3{acc + 3, acc + 4, acc + 5}; }
        }
    };
    if (CYCLIC) { // blocks of hb row-steps dealt round-robin to the strip's workgroups: they sweep down together
        const int nblk = (a.rows / GY + a.hb - 1) / a.hb;
        for (int b = ph; b < nblk; b += wg_per_strip)
            for (int s = 0; s < a.hb; ++s) { const int r = (b * a.hb + s) * GY + wy; if (r < a.rows) do_row(r); }
    } else {      // contiguous share of the rows
        const int per = (a.rows + wg_per_strip - 1) / wg_per_strip;
        const int r0 = ph * per, r1 = min(a.rows, r0 + per);
        for (int r = r0 + wy; r < r1; r += GY) do_row(r);
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

template <int VPX, int GX, int CYCLIC> float run(Args a, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void *)k_stream<VPX, GX, CYCLIC>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipLaunchKernelGGL((k_stream<VPX, GX, CYCLIC>), dim3(a.nwg), dim3(1024), 163840, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<VPX, GX, CYCLIC>), dim3(a.nwg), dim3(1024), 163840, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int rows = 20000, cols = 20000; const size_t pitch = 20480;
    Args a{}; unsigned short *d0, *d1; unsigned char *rgb; unsigned *sink;
    CK(hipMalloc(&d0, pitch * rows * 2)); CK(hipMalloc(&d1, pitch * rows * 2)); CK(hipMalloc(&rgb, pitch * rows * 3)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(d0, 1, pitch * rows * 2)); CK(hipMemset(d1, 2, pitch * rows * 2));
    a.in0 = d0; a.in1 = d1; a.rgb = rgb; a.pitch = pitch; a.rows = rows; a.cols = 20480; a.sink = sink;
    const double rd = 2.0 * 20480.0 * rows * 2, wr = 3.0 * 20480.0 * rows;
    for (int store = 0; store < 2; ++store) {
        a.store = store;
        const double bytes = rd + (store ? wr : 0);
        for (int hb : {0, 4, 16, 64}) {
            a.hb = hb ? hb : 1;
            a.nwg = 250; // 10 strips of 2048 (VPX 4, GX 8) -> 25 workgroups each; VPX 8, GX 4 the same strip width
            float t;
            if (hb == 0) {
                t = run<4, 8, 0>(a, 10); printf("store %d  8B/lane gx8  contiguous      : %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
                t = run<8, 4, 0>(a, 10); printf("store %d 16B/lane gx4  contiguous      : %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
                t = run<8, 8, 0>(a, 10); printf("store %d 16B/lane gx8  contiguous      : %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
            } else {
                t = run<4, 8, 1>(a, 10); printf("store %d  8B/lane gx8  cyclic hb %-3d   : %.3f ms  %.2f TB/s\n", store, hb, t, bytes / t / 1e9);
                t = run<8, 4, 1>(a, 10); printf("store %d 16B/lane gx4  cyclic hb %-3d   : %.3f ms  %.2f TB/s\n", store, hb, t, bytes / t / 1e9);
                t = run<8, 8, 1>(a, 10); printf("store %d 16B/lane gx8  cyclic hb %-3d   : %.3f ms  %.2f TB/s\n", store, hb, t, bytes / t / 1e9);
            }
        }
    }
    for (int valu : {0, 4, 8, 16, 32}) { // does vector work per row hide under the memory time?
        a.valu = valu; a.store = 1; a.nwg = 250; a.hb = 4;
        float t = run<4, 8, 0>(a, 10); printf("store 1  8B/lane gx8 contiguous + %3d VALU/row: %.3f ms  %.2f TB/s\n", valu * 12, t, (rd + wr) / t / 1e9);
        a.store = 0; t = run<4, 8, 0>(a, 10); printf("store 0  8B/lane gx8 contiguous + %3d VALU/row: %.3f ms  %.2f TB/s\n", valu * 12, t, rd / t / 1e9);
    }
    a.valu = 0;
    { a.store = 2; a.nwg = 250; float t = run<4, 8, 0>(a, 10); printf("store x4 (48 lanes)  8B/lane gx8 contiguous: %.3f ms  %.2f TB/s\n", t, (rd + wr) / t / 1e9); a.valu = 16; t = run<4, 8, 0>(a, 10); printf("store x4 (48 lanes) +192 VALU              : %.3f ms\n", t); a.store = 1; t = run<4, 8, 0>(a, 10); printf("store x3 (64 lanes) +192 VALU              : %.3f ms\n", t); a.valu = 0; }
    // narrow strips: does a workgroup that reads 512-B / 1-KiB / 2-KiB row segments stream as fast as one that reads 4 KiB?
    for (int store = 0; store < 2; ++store) {
        a.store = store; a.hb = 4;
        const double bytes = rd + (store ? wr : 0);
        float t;
        a.nwg = 240; t = run<4, 1, 0>(a, 10); printf("store %d  8B/lane gx1 (256 px)  contiguous: %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
        a.nwg = 240; t = run<4, 2, 0>(a, 10); printf("store %d  8B/lane gx2 (512 px)  contiguous: %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
        a.nwg = 240; t = run<4, 4, 0>(a, 10); printf("store %d  8B/lane gx4 (1024 px) contiguous: %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
        a.nwg = 240; t = run<4, 4, 1>(a, 10); printf("store %d  8B/lane gx4 (1024 px) cyclic 4  : %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
        a.nwg = 240; t = run<4, 16, 0>(a, 10); printf("store %d  8B/lane gx16 (4096 px) contiguous: %.3f ms  %.2f TB/s\n", store, t, bytes / t / 1e9);
    }
    return 0;
}

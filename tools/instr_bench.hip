// instr_bench.hip -- issue cost of the vector instructions the fused pass leans on (tools/, not part of the library).
// Each kernel: 256 blocks x 1024 threads (4 waves per SIMD), a loop of 64 x 16 independent instances of ONE instruction.
// Prints cycles per wave-instruction per SIMD at the clock s_memtime/s_memrealtime report.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP> __global__ __launch_bounds__(1024) void k(unsigned *out, int iters, unsigned long long *clk) {
    unsigned v[16]; float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 2654435761u + i; f[i] = (float)(threadIdx.x + i) * 0.37f; }
    unsigned s4 = 4u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
                if (OP == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(v[i]) : "v"(f[i]));
                if (OP == 2) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v[i]) : "v"(s4));
                if (OP == 3) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
                if (OP == 7) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[i]));
                if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
                if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double *)&f[i & 14]) : "v"(*(double *)&f[(i + 2) & 14]));
                if (OP == 10) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
                if (OP == 11) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(unsigned long long *)&v[i & 14]) : "v"(*(unsigned long long *)&v[(i + 2) & 14]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i] + (unsigned)f[i];
    if (acc == 0x12345u) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int OP> int run(const char *name, unsigned *d, unsigned long long *clk) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters, clk);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, d, iters, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    // per SIMD: 4 waves x iters x 64 instructions each
    printf("%-22s %6.2f s_memtime ticks, %6.2f ns per wave-instruction per SIMD (4 waves/SIMD; x clock GHz = cycles)\n", name, (double)c / (4.0 * iters * 64), ms * 1e6 / (4.0 * iters * 64));
    return 0;
}
int main() {
    unsigned *d; unsigned long long *clk; CK(hipMalloc(&d, 64)); CK(hipMalloc(&clk, 8));
    run<0>("v_fma_f32", d, clk); run<6>("v_add_f32", d, clk); run<1>("v_cvt_pk_u8_f32", d, clk); run<2>("v_lshlrev_b32_sdwa", d, clk);
    run<3>("v_pk_min_u16", d, clk); run<4>("v_bcnt_u32_b32", d, clk); run<5>("v_and_b32", d, clk); run<7>("v_bfe_u32", d, clk);
    run<8>("v_perm_b32", d, clk); run<9>("v_pk_fma_f32", d, clk); run<10>("v_or3_b32", d, clk); run<11>("v_lshl_add_u64", d, clk);
    return 0;
}

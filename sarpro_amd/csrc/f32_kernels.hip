// f32_kernels.hip -- kernels of the f32-input flavour (pol-op results, resampled reads, any
// Array2<f32> a caller hands to process_scalar_data_pipeline).
//
// Every decision the reference takes per pixel (valid?, 4096-bin index, CLAHE bin, output level)
// is a monotone step function of the sample, so the host ships it as a sorted table of f32
// thresholds (host_logic.cpp) and the device resolves it with a branch-free binary search in
// LDS -- compares only, no device log10/pow in any integer result.  The only transcendental
// evaluated here is log10 for the dB buffer itself (a1) and for mean/std, which tolerate ulps.
#include "f32_kernels.h"

#include <algorithm>

namespace sarpro {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;

__device__ inline int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

// number of k in 1..N with v >= thr[k]  (thr sorted ascending, N = 2^m - 1, NaN compares false)
template <int N>
__device__ inline uint32_t step_search(const float *thr, float v) {
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t step = (N + 1) / 2; step; step >>= 1)
        if (v >= thr[idx + step]) idx += step;
    return idx;
}

// The same count by estimate + verification: thr[0] = -inf and thr[N + 1] = +inf must be readable.  k is right iff
// thr[k] <= v < thr[k + 1] (ascending table), whatever produced k; otherwise the binary search decides.
template <int N>
__device__ inline uint32_t est_search(const float *thr, float v, const F32StepEstimate &e) {
    float t = fminf(fmaxf(__builtin_amdgcn_logf(v * e.inv_x0) * e.scale, 0.0f), 1.0f); // log2; NaN -> 0
    if (e.gamma != 1.0f) t = __builtin_amdgcn_exp2f(e.gamma * __builtin_amdgcn_logf(t)); // t^gamma (t = 0: exp2(-inf) = 0)
    const uint32_t k = (uint32_t)fminf(fmaxf(t * e.nsteps + e.bias, 0.0f), (float)N);
    const float t0 = thr[k], t1 = thr[k + 1];
    if (t0 <= v && v < t1) return k;
    return step_search<N>(thr, v);
}

// ops.rs:4-44, IEEE f32 (hipcc's default correctly rounded divide); the same function as kernels.hip k_polop_f32
__device__ inline float pol_one(int op, float x, float y) {
    switch (op) {
    case SARPRO_OP_SUM: return x + y;
    case SARPRO_OP_DIFF: return x - y;
    case SARPRO_OP_RATIO:
    case SARPRO_OP_LOGRATIO: return fabsf(y) > 1e-10f ? x / y : 0.0f;
    default: { const float d = x + y; return fabsf(d) > 1e-10f ? (x - y) / d : 0.0f; }
    }
}

template <int VEC> struct F32Vec;
template <> struct F32Vec<4> {
    float4 v;
    __device__ static F32Vec load(const float *p) { F32Vec r; r.v = *reinterpret_cast<const float4 *>(p); return r; }
    __device__ float get(int j) const { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
    // the samples of row r, columns col .. col + 3: the raster, or the pol-op of two rasters computed here
    __device__ static F32Vec fetch(const float *in, size_t pitch, const F32Pol &p, size_t r, size_t col) {
        if (p.op < 0) return load(in + r * pitch + col);
        float4 x, y;
        if (p.u16) {
            const ushort4 xa = *reinterpret_cast<const ushort4 *>(reinterpret_cast<const uint16_t *>(p.a) + r * p.pitch + col);
            const ushort4 xb = *reinterpret_cast<const ushort4 *>(reinterpret_cast<const uint16_t *>(p.b) + r * p.pitch + col);
            x = make_float4((float)xa.x, (float)xa.y, (float)xa.z, (float)xa.w);
            y = make_float4((float)xb.x, (float)xb.y, (float)xb.z, (float)xb.w);
        } else {
            x = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.a) + r * p.pitch + col);
            y = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.b) + r * p.pitch + col);
        }
        F32Vec o;
        o.v = make_float4(pol_one(p.op, x.x, y.x), pol_one(p.op, x.y, y.y), pol_one(p.op, x.z, y.z), pol_one(p.op, x.w, y.w));
        return o;
    }
};
template <> struct F32Vec<1> {
    float v;
    __device__ static F32Vec load(const float *p) { F32Vec r; r.v = *p; return r; }
    __device__ float get(int) const { return v; }
    __device__ static F32Vec fetch(const float *in, size_t pitch, const F32Pol &p, size_t r, size_t col) {
        if (p.op < 0) return load(in + r * pitch + col);
        const size_t i = r * p.pitch + col;
        F32Vec o;
        o.v = p.u16 ? pol_one(p.op, (float)reinterpret_cast<const uint16_t *>(p.a)[i], (float)reinterpret_cast<const uint16_t *>(p.b)[i])
                    : pol_one(p.op, reinterpret_cast<const float *>(p.a)[i], reinterpret_cast<const float *>(p.b)[i]);
        return o;
    }
};

// VEC levels of one lane -> the output raster: one 4- or 8-byte store when the whole vector lies inside the row
// and the raster allows it (16-byte-aligned base is the callers' contract for VEC = 4, pitch % 4 == 0 checked here).
template <int VEC, bool OUT16>
__device__ inline void store_levels(void *out, size_t elem, const uint32_t *lv, int n, bool vec_ok) {
    if (VEC == 4 && n == 4 && vec_ok) {
        if (OUT16) *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(out) + elem) = make_uint2(lv[0] | (lv[1] << 16), lv[2 % VEC] | (lv[3 % VEC] << 16));
        else *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(out) + elem) = lv[0] | (lv[1 % VEC] << 8) | (lv[2 % VEC] << 16) | (lv[3 % VEC] << 24);
    } else {
        for (int j = 0; j < n; ++j) {
            if (OUT16) reinterpret_cast<uint16_t *>(out)[elem + j] = (uint16_t)lv[j];
            else reinterpret_cast<uint8_t *>(out)[elem + j] = (uint8_t)lv[j];
        }
    }
}

// ------------------------------------------------------------------------------------
// a. pre-pass: count / min / max of the valid samples and the dB moments (deterministic:
//    per-block partials, reduced by the host in block order).
// ------------------------------------------------------------------------------------
// 10 log10(x) of a normal, positive f32 in f64, for the dB moments (mean / std: statistics output and the two discrete
// tests of the Adaptive strategy; tolerance 1e-9).  x = m 2^e with a 24-bit m: the top 8 mantissa
// bits pick a centre c with log2(c) and 1/c from a 256-entry table (built by the block with the library log2),
// r = m / c - 1 lies within 2^-9 and log2(1 + r) is a degree-5 series: ~15 f64 operations instead of the ~100 of the
// library log10, absolute error of log2 below 1e-15.
__device__ inline double db_of_f32_fast(float x, const double *logc, const double *invc) {
    const uint32_t bits = __float_as_uint(x);
    const int e = (int)(bits >> 23) - 127;
    const uint32_t mant = bits & 0x7FFFFFu;
    const uint32_t i = mant >> 15;
    const double m = 1.0 + (double)mant * 0x1p-23;
    const double r = fma(m, invc[i], -1.0);
    const double p = r * (1.0 - r * (0.5 - r * (1.0 / 3.0 - r * (0.25 - r * 0.2))));
    const double log2x = ((double)e + logc[i]) + p * 1.4426950408889634; // 1 / ln 2
    return log2x * 3.0102999566398120;                                  // 10 log10(2)
}

template <int VEC, bool MOMENTS>
__global__ __launch_bounds__(kBlock) void k_f32_prepass(const float *__restrict__ in, size_t pitch, uint32_t rows,
                                                        uint32_t cols, float t_valid, F32Partial *__restrict__ out, F32Pol pol) {
    const uint32_t vpr = (cols + VEC - 1) / VEC;
    const uint64_t total = (uint64_t)rows * vpr;
    unsigned long long cnt = 0;
    double sum = 0.0, sumsq = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    __shared__ double logc[256], invc[256];
    if (MOMENTS) {
        static_assert(kBlock == 256, "one table entry per thread");
        const double c = 1.0 + ((double)threadIdx.x + 0.5) / 256.0;
        logc[threadIdx.x] = log2(c);
        invc[threadIdx.x] = 1.0 / c;
        __syncthreads();
    }
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(idx / vpr);
        const uint32_t col = (uint32_t)(idx - (uint64_t)r * vpr) * VEC;
        const F32Vec<VEC> v = F32Vec<VEC>::fetch(in, pitch, pol, r, col);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            if (col + j < cols && x >= t_valid) {
                ++cnt;
                mn = fminf(mn, x);
                mx = fmaxf(mx, x);
                if (MOMENTS) {
                    const double db = db_of_f32_fast(x, logc, invc);
                    sum += db;
                    sumsq += db * db;
                }
            }
        }
    }
    __shared__ F32Partial part[kBlock];
    part[threadIdx.x] = F32Partial{cnt, sum, sumsq, mn, mx};
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) { // fixed tree: deterministic
        if ((int)threadIdx.x < s) {
            F32Partial a = part[threadIdx.x], b = part[threadIdx.x + s];
            a.count += b.count; a.sum += b.sum; a.sumsq += b.sumsq;
            a.minv = fminf(a.minv, b.minv); a.maxv = fmaxf(a.maxv, b.maxv);
            part[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

// ------------------------------------------------------------------------------------
// b. 4096-bin statistics histogram (autoscale.rs:108-117) by threshold search.
// ------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_f32_hist4096(const float *__restrict__ in, size_t pitch, uint32_t rows,
                                                         uint32_t cols, float t_valid, const float *__restrict__ g_thr,
                                                         unsigned long long *__restrict__ g_hist, F32StepEstimate est, F32Pol pol) {
    __shared__ float thr[4096 + 1];
    __shared__ uint32_t hist[4096];
    for (int i = threadIdx.x; i < 4096; i += kBlock) { thr[i] = i ? g_thr[i] : -INFINITY; hist[i] = 0; }
    if (threadIdx.x == 0) thr[4096] = INFINITY;
    __syncthreads();
    const uint32_t vpr = (cols + VEC - 1) / VEC;
    const uint64_t total = (uint64_t)rows * vpr;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(idx / vpr);
        const uint32_t col = (uint32_t)(idx - (uint64_t)r * vpr) * VEC;
        const F32Vec<VEC> v = F32Vec<VEC>::fetch(in, pitch, pol, r, col);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            if (col + j < cols && x >= t_valid) atomicAdd(&hist[est.use ? est_search<4095>(thr, x, est) : step_search<4095>(thr, x)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += kBlock)
        if (hist[i]) atomicAdd(&g_hist[i], (unsigned long long)hist[i]);
}

// ------------------------------------------------------------------------------------
// c. level map for the percentile strategies (autoscale.rs:437-447 / 647-655 / 731-741).
//    u8: 255 thresholds in LDS + histogram of the levels; u16: 65535 thresholds gathered
//    from global memory (256 KiB, L2-resident).
// ------------------------------------------------------------------------------------
template <int VEC, bool OUT16>
__global__ __launch_bounds__(kBlock) void k_f32_level(F32LevelArgs a) {
    __shared__ float thr[256 + 1];
    __shared__ uint32_t hist[256];
    if (!OUT16) { thr[threadIdx.x] = threadIdx.x ? a.thr[threadIdx.x] : -INFINITY; hist[threadIdx.x] = 0; }
    if (threadIdx.x == 0) thr[256] = INFINITY;
    __shared__ double logc[256], invc[256]; // table of db_of_f32_fast (u16 levels)
    if (OUT16) {
        const double c = 1.0 + ((double)threadIdx.x + 0.5) / 256.0;
        logc[threadIdx.x] = log2(c);
        invc[threadIdx.x] = 1.0 / c;
    }
    __syncthreads();
    const uint32_t vpr = (a.cols + VEC - 1) / VEC;
    const uint64_t total = (uint64_t)a.rows * vpr;
    const bool vec_store = a.out_pitch % VEC == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
    uint32_t zeros = 0;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(idx / vpr);
        const uint32_t col = (uint32_t)(idx - (uint64_t)r * vpr) * VEC;
        const F32Vec<VEC> v = F32Vec<VEC>::fetch(a.in, a.in_pitch, a.pol, r, col);
        uint32_t lvs[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            uint32_t lv = 0;
            if (col + j < a.cols && x >= a.t_valid) {
                if (OUT16 && a.f64_levels) {
                    if (x >= a.t_last) lv = 65535u;
                    else if (x < a.t_first) lv = 0u;
                    else {
                        const double db = db_of_f32_fast(x, logc, invc); // within ~1e-14 of glibc's value: far inside the 1e-6 margin below
                        const double t = (fmin(fmax(db, a.low), a.high) - a.low) / a.range;
                        const double y = fmin(fmax((a.gamma == 1.0 ? t : pow(t, a.gamma)) * a.max_val, 0.0), a.max_val);
                        const double r = rint(y);
                        lv = (fabs(y - r) < 1e-6 || !(y == y)) ? step_search<65535>(a.thr, x) : (uint32_t)y;
                    }
                } else if (a.est.use) lv = OUT16 ? est_search<65535>(a.thr, x, a.est) : est_search<255>(thr, x, a.est);
                else lv = OUT16 ? step_search<65535>(a.thr, x) : step_search<255>(thr, x);
            }
            lvs[j] = lv;
            if (!OUT16 && col + j < a.cols) { if (lv == 0) ++zeros; else atomicAdd(&hist[lv], 1u); }
        }
        store_levels<VEC, OUT16>(a.out, (size_t)r * a.out_pitch + col, lvs, (int)min((uint32_t)VEC, a.cols - col), vec_store);
    }
    if (!OUT16) {
        if (zeros) atomicAdd(&hist[0], zeros);
        __syncthreads();
        if (hist[threadIdx.x]) atomicAdd(&a.level_hist[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------
// d. CLAHE per-tile 256-bin histograms straight from the samples (autoscale.rs:247-269).
// ------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_f32_tile_hist(F32TileHistArgs a) {
    __shared__ float thr[256 + 1];
    __shared__ uint32_t hist[256];
    thr[threadIdx.x] = threadIdx.x ? a.thr[threadIdx.x] : -INFINITY;
    if (threadIdx.x == 0) thr[256] = INFINITY;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const Rect rc = a.rects[blockIdx.x];
    const int col = rc.cstart + lane_id() * VEC;
    if (col < rc.c1 && col + VEC > rc.c0) {
        for (int r = rc.r0 + wave_id(); r < rc.r1; r += kWavesPerBlock) {
            const F32Vec<VEC> v = F32Vec<VEC>::fetch(a.in, a.pitch, a.pol, r, col);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = col + j;
                const float x = v.get(j);
                if (c >= rc.c0 && c < rc.c1 && x >= a.t_valid) atomicAdd(&hist[a.est.use ? est_search<255>(thr, x, a.est) : step_search<255>(thr, x)], 1u);
            }
        }
    }
    __syncthreads();
    if (hist[threadIdx.x])
        atomicAdd(&a.tile_bins[(size_t)rc.id[0] * 256 + threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// ------------------------------------------------------------------------------------
// e. CLAHE apply for f32 samples: same blend as k_clahe_apply_u16 (kernels.hip), bin by
//    threshold search.
// ------------------------------------------------------------------------------------
template <int VEC, bool OUT16>
__global__ __launch_bounds__(kBlock) void k_f32_clahe_apply(F32ClaheApplyArgs a) {
    __shared__ __align__(16) double cdf4[256 * 4];
    __shared__ float thr[256 + 1];
    __shared__ uint32_t hist[256];
    const Rect rc = a.rects[blockIdx.x];
    {
        const int b = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) cdf4[b * 4 + k] = a.cdfs[(size_t)rc.id[k] * 256 + b];
        thr[b] = b ? a.thr[b] : -INFINITY;
        if (b == 0) thr[256] = INFINITY;
        hist[b] = 0;
    }
    __syncthreads();
    const int col = rc.cstart + lane_id() * VEC;
    const bool lane_on = col < rc.c1 && col + VEC > rc.c0;
    const bool vec_store = a.out_pitch % VEC == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
    uint32_t zeros = 0;
    double dx[VEC], omdx[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = col + j;
        const RowWeight w = a.col_w[(c >= rc.c0 && c < rc.c1) ? c : rc.c0];
        dx[j] = w.d;
        omdx[j] = w.omd;
    }
    if (lane_on) {
        for (int r = rc.r0 + wave_id(); r < rc.r1; r += kWavesPerBlock) {
            const F32Vec<VEC> v = F32Vec<VEC>::fetch(a.in, a.in_pitch, a.pol, r, col);
            const RowWeight rw = a.row_w[r];
            uint32_t lvs[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = col + j;
                const bool own = c >= rc.c0 && c < rc.c1;
                const float x = v.get(j);
                uint32_t lv = 0;
                if (own && x >= a.t_valid) {
                    const uint32_t bin = a.est.use ? est_search<255>(thr, x, a.est) : step_search<255>(thr, x);
                    const double4 c4 = *reinterpret_cast<const double4 *>(&cdf4[bin * 4]);
                    const double top = c4.x * omdx[j] + c4.y * dx[j];
                    const double bottom = c4.z * omdx[j] + c4.w * dx[j];
                    double o = top * rw.omd + bottom * rw.d;
                    o = fmin(fmax(o, 0.0), 1.0);
                    lv = (uint32_t)(o * a.max_val);
                }
                lvs[j] = lv;
                if (!OUT16 && own) { if (lv == 0) ++zeros; else atomicAdd(&hist[lv], 1u); }
            }
            if (col >= rc.c0 && col + VEC <= rc.c1) {
                store_levels<VEC, OUT16>(a.out, (size_t)r * a.out_pitch + col, lvs, VEC, vec_store);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int c = col + j;
                    if (c < rc.c0 || c >= rc.c1) continue;
                    if (OUT16) reinterpret_cast<uint16_t *>(a.out)[(size_t)r * a.out_pitch + c] = (uint16_t)lvs[j];
                    else reinterpret_cast<uint8_t *>(a.out)[(size_t)r * a.out_pitch + c] = (uint8_t)lvs[j];
                }
            }
        }
    }
    if (!OUT16) {
        if (zeros) atomicAdd(&hist[0], zeros);
        __syncthreads();
        if (hist[threadIdx.x]) atomicAdd(&a.level_hist[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------------
// f. process_scalar_data_inplace (pipeline.rs:8-40): the dB buffer and the validity mask.
//    mask is exact (threshold compare); db is the device's f64 log10 (<= 1 ulp from glibc's).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_db_mask_f32(const float *__restrict__ in, size_t n, float t_valid,
                                                        double *__restrict__ db, uint8_t *__restrict__ mask) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float x = in[i];
        if (db) db[i] = 10.0 * log10(fmax((double)x, 1e-10));
        if (mask) mask[i] = x >= t_valid ? 1 : 0;
    }
}

inline int stream_grid(uint64_t items, int per_cu = 8) {
    const uint64_t want = (items + kBlock - 1) / kBlock, cap = 256ull * per_cu;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

} // namespace

int f32_prepass_grid(uint32_t rows, uint32_t cols, bool vec) {
    const int V = vec ? 4 : 1;
    return stream_grid((uint64_t)rows * ((cols + V - 1) / V), 4);
}

hipError_t launch_f32_prepass(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, bool moments,
                              F32Partial *d_partials, int grid, hipStream_t s, const F32Pol &pol) {
    if (vec) {
        if (moments) hipLaunchKernelGGL((k_f32_prepass<4, true>), dim3(grid), dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_partials, pol);
        else hipLaunchKernelGGL((k_f32_prepass<4, false>), dim3(grid), dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_partials, pol);
    } else {
        if (moments) hipLaunchKernelGGL((k_f32_prepass<1, true>), dim3(grid), dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_partials, pol);
        else hipLaunchKernelGGL((k_f32_prepass<1, false>), dim3(grid), dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_partials, pol);
    }
    return hipGetLastError();
}

hipError_t launch_f32_hist4096(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec,
                               const float *d_thr, unsigned long long *d_hist, F32StepEstimate est, hipStream_t s, const F32Pol &pol) {
    const int V = vec ? 4 : 1;
    dim3 grid(stream_grid((uint64_t)rows * ((cols + V - 1) / V), 4));
    if (vec) hipLaunchKernelGGL(k_f32_hist4096<4>, grid, dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_thr, d_hist, est, pol);
    else hipLaunchKernelGGL(k_f32_hist4096<1>, grid, dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_thr, d_hist, est, pol);
    return hipGetLastError();
}

hipError_t launch_f32_level(const F32LevelArgs &a, bool vec, bool out16, hipStream_t s) {
    const int V = vec ? 4 : 1;
    dim3 grid(stream_grid((uint64_t)a.rows * ((a.cols + V - 1) / V)));
    if (vec) {
        if (out16) hipLaunchKernelGGL((k_f32_level<4, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_f32_level<4, false>), grid, dim3(kBlock), 0, s, a);
    } else {
        if (out16) hipLaunchKernelGGL((k_f32_level<1, true>), grid, dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_f32_level<1, false>), grid, dim3(kBlock), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_f32_tile_hist(const F32TileHistArgs &a, int nrects, bool vec, hipStream_t s) {
    if (nrects <= 0) return hipSuccess;
    if (vec) hipLaunchKernelGGL(k_f32_tile_hist<4>, dim3(nrects), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(k_f32_tile_hist<1>, dim3(nrects), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_f32_clahe_apply(const F32ClaheApplyArgs &a, int nrects, bool vec, bool out16, hipStream_t s) {
    if (nrects <= 0) return hipSuccess;
    if (vec) {
        if (out16) hipLaunchKernelGGL((k_f32_clahe_apply<4, true>), dim3(nrects), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_f32_clahe_apply<4, false>), dim3(nrects), dim3(kBlock), 0, s, a);
    } else {
        if (out16) hipLaunchKernelGGL((k_f32_clahe_apply<1, true>), dim3(nrects), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_f32_clahe_apply<1, false>), dim3(nrects), dim3(kBlock), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_db_mask_f32(const float *in, size_t n, float t_valid, double *db, uint8_t *mask, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_db_mask_f32, dim3(stream_grid(n)), dim3(kBlock), 0, s, in, n, t_valid, db, mask);
    return hipGetLastError();
}

} // namespace sarpro

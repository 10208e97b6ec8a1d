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

// A threshold table as PAIRS (thr[k], thr[k + 1]) at [k]: the verification of an estimate reads both bounds of its step with ONE
// 8-byte LDS read.  (As two 4-byte reads -- a ds_read2_b32 -- the 32 lanes of a group pick among 32 banks twice, ~3.5 lanes on the
// busiest one each time: 14 LDS cycles per wave against 7 for the ds_read_b64 of a pair; the kernels that search per sample are
// bound by exactly these cycles.)
struct ThrPairs {
    const float2 *p;
    __device__ float operator[](uint32_t i) const { return p[i].x; }
};
__device__ inline void thr_bounds(const float *thr, uint32_t k, float *t0, float *t1) { *t0 = thr[k]; *t1 = thr[k + 1]; }
#ifdef SARPRO_ABL_F32_THR_SPLIT // timing ablation: the two bounds as two 4-byte reads (what the plain thr[] table of rounds 1-3 cost)
__device__ inline void thr_bounds(const ThrPairs &thr, uint32_t k, float *t0, float *t1) {
    *t0 = *reinterpret_cast<const volatile float *>(&thr.p[k].x);
    *t1 = *reinterpret_cast<const volatile float *>(&thr.p[k].y);
}
#else
__device__ inline void thr_bounds(const ThrPairs &thr, uint32_t k, float *t0, float *t1) { const float2 t = thr.p[k]; *t0 = t.x; *t1 = t.y; }
#endif
// fills pairs[0 .. n - 1] from thr[1 .. n - 1] (global), thr[0] = -inf, thr[n] = +inf; n threads of the workgroup call it with i = their index
__device__ inline void thr_pairs_fill(float2 *pairs, const float *g_thr, int n, int i) {
    if (i < n) pairs[i] = make_float2(i ? g_thr[i] : -INFINITY, i + 1 < n ? g_thr[i + 1] : INFINITY);
}

// number of k in 1..N with v >= thr[k]  (thr sorted ascending, N = 2^m - 1, NaN compares false)
template <int N, typename Tab>
__device__ inline uint32_t step_search(const Tab &thr, float v) {
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t step = (N + 1) / 2; step; step >>= 1)
        if (v >= thr[idx + step]) idx += step;
    return idx;
}

// The same count by estimate + verification: thr[0] = -inf and thr[N + 1] = +inf must be readable.  k is right iff
// thr[k] <= v < thr[k + 1] (ascending table), whatever produced k; otherwise the binary search decides.
template <int N, typename Tab>
__device__ inline uint32_t est_search(const Tab &thr, float v, const F32StepEstimate &e) {
    uint32_t k;
    if (e.gamma == 1.0f) { // the common case, folded: the estimate only has to be right often, the two reads below decide
        k = (uint32_t)__builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_amdgcn_logf(v), e.a_mul, e.b_add), 0.0f, (float)N); // NaN -> 0
    } else {
        float t = fminf(fmaxf(__builtin_amdgcn_logf(v * e.inv_x0) * e.scale, 0.0f), 1.0f); // log2; NaN -> 0
        t = __builtin_amdgcn_exp2f(e.gamma * __builtin_amdgcn_logf(t)); // t^gamma (t = 0: exp2(-inf) = 0)
        k = (uint32_t)fminf(fmaxf(t * e.nsteps + e.bias, 0.0f), (float)N);
    }
    float t0, t1;
    thr_bounds(thr, k, &t0, &t1);
    if (t0 <= v && v < t1) return k;
    return step_search<N>(thr, v);
}

// est_search for M samples at once: all estimates, then all table reads, then all verifications -- no branch between the samples,
// so their chains (v_log_f32, two dependent LDS reads) overlap; ONE branch for the samples whose estimate failed.
template <int N, int M, typename Tab>
__device__ inline void est_search_m(const Tab &thr, const float (&v)[M], const F32StepEstimate &e, uint32_t (&k)[M]) {
    if (e.gamma == 1.0f) {
        float t0[M], t1[M];
#pragma unroll
        for (int j = 0; j < M; ++j) k[j] = (uint32_t)__builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_amdgcn_logf(v[j]), e.a_mul, e.b_add), 0.0f, (float)N);
#pragma unroll
        for (int j = 0; j < M; ++j) thr_bounds(thr, k[j], &t0[j], &t1[j]);
        uint32_t bad = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) bad |= (t0[j] <= v[j] && v[j] < t1[j]) ? 0u : (1u << j);
        if (bad) {
#pragma unroll
            for (int j = 0; j < M; ++j)
                if ((bad >> j) & 1u) k[j] = step_search<N>(thr, v[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < M; ++j) k[j] = est_search<N>(thr, v[j], e);
    }
}

// ops.rs:4-44, IEEE f32 (hipcc's default correctly rounded divide); the same function as kernels.hip k_polop_f32
// num / den for operands that are integers of magnitude < 2^18 (u16 DN, their sums and differences), den != 0: the hardware's
// reciprocal (1 ulp), the quotient it gives, and ONE correction of that quotient by its exact residual -- 4 instructions
// against the 10 of the compiler's IEEE division (v_div_scale x2, a refined reciprocal, two corrections, v_div_fmas,
// v_div_fixup).  The frame rescales extreme exponents and patches inf / nan / zero denominators: the identity for these
// operands.  The refinement and the second correction matter only when the true quotient lies within ~1e-7 ulp of a rounding
// boundary, which a ratio of integers this small cannot (its distance from a 25-bit midpoint is at least 2^-25 / den of its
// own magnitude).  Not taken on trust: checked against `/` over ALL 2^32 pairs of u16 values, for the ratio and for the
// normalised difference, on the hardware the suite runs on (tests/test_gpu_polop_fused.py; round 2's 8-instruction form and
// the two forms in between pass the same check).
__device__ inline float div_small_ints(float num, float den) {
    const float r = __builtin_amdgcn_rcpf(den);
    const float q = num * r;
    return __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
}

// ops.rs:10-19 / 22-33 / 35-44: num / den where |den| > 1e-10, else 0
template <bool INTS>
__device__ inline float ratio_one(float num, float den) {
    const bool ok = fabsf(den) > 1e-10f;
    const float dsafe = ok ? den : 1.0f; // unconditional division: the compiler would otherwise put each behind its own branch
    const float q = INTS ? div_small_ints(num, dsafe) : num / dsafe;
    return ok ? q : 0.0f;
}

template <bool INTS = false>
__device__ inline float pol_one(int op, float x, float y) {
    // straight-line in the (uniform) operation: one guarded division whatever it is, so that the divisions of a vector's four
    // elements interleave instead of sitting behind a chain of scalar branches each
    const bool nd = op == SARPRO_OP_NDIFF;
    const float s = x + y, d = x - y;
    const float num = nd ? d : x, den = nd ? s : y;
    const bool ok = fabsf(den) > 1e-10f;
    const float dsafe = ok ? den : 1.0f;       // unconditional division: the compiler would otherwise put each behind its own branch
    const float q0 = INTS ? div_small_ints(num, dsafe) : num / dsafe;
    const float q = ok ? q0 : 0.0f;
    return op == SARPRO_OP_SUM ? s : (op == SARPRO_OP_DIFF ? d : q);
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
    // The same in two steps -- the loads alone, then the arithmetic -- so that a loop can issue the NEXT item's loads before it
    // works on the current one (one item in flight per wave leaves HBM idle: bytes in flight / latency is the bandwidth).
    struct Raw { uint4 a; uint2 b, c; };
    __device__ static Raw load_raw(const float *in, size_t pitch, const F32Pol &p, size_t r, size_t col) {
        Raw w;
        w.a = make_uint4(0, 0, 0, 0); w.b = make_uint2(0, 0); w.c = make_uint2(0, 0);
        if (p.op < 0) w.a = *reinterpret_cast<const uint4 *>(in + r * pitch + col);
        else if (p.u16) {
            w.b = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p.a) + r * p.pitch + col);
            w.c = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p.b) + r * p.pitch + col);
        } else {
            w.a = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(p.a) + r * p.pitch + col);
            const uint4 y = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(p.b) + r * p.pitch + col);
            w.b = make_uint2(y.x, y.y); w.c = make_uint2(y.z, y.w);
        }
        return w;
    }
    __device__ static F32Vec make(const Raw &w, const F32Pol &p) {
        F32Vec o;
        if (p.op < 0) { o.v = make_float4(__uint_as_float(w.a.x), __uint_as_float(w.a.y), __uint_as_float(w.a.z), __uint_as_float(w.a.w)); return o; }
        float4 x, y;
        if (p.u16) {
            x = make_float4((float)(w.b.x & 0xFFFFu), (float)(w.b.x >> 16), (float)(w.b.y & 0xFFFFu), (float)(w.b.y >> 16));
            y = make_float4((float)(w.c.x & 0xFFFFu), (float)(w.c.x >> 16), (float)(w.c.y & 0xFFFFu), (float)(w.c.y >> 16));
        } else {
            x = make_float4(__uint_as_float(w.a.x), __uint_as_float(w.a.y), __uint_as_float(w.a.z), __uint_as_float(w.a.w));
            y = make_float4(__uint_as_float(w.b.x), __uint_as_float(w.b.y), __uint_as_float(w.c.x), __uint_as_float(w.c.y));
        }
        // one uniform branch per VECTOR on the kind of operation, then four straight-line elements (pol_one's selects on the
        // operation cost six instructions per element when it is not known at compile time)
        if (p.op == SARPRO_OP_RATIO || p.op == SARPRO_OP_LOGRATIO) {
            if (p.u16) o.v = make_float4(ratio_one<true>(x.x, y.x), ratio_one<true>(x.y, y.y), ratio_one<true>(x.z, y.z), ratio_one<true>(x.w, y.w));
            else o.v = make_float4(ratio_one<false>(x.x, y.x), ratio_one<false>(x.y, y.y), ratio_one<false>(x.z, y.z), ratio_one<false>(x.w, y.w));
        } else if (p.op == SARPRO_OP_NDIFF) {
            if (p.u16) o.v = make_float4(ratio_one<true>(x.x - y.x, x.x + y.x), ratio_one<true>(x.y - y.y, x.y + y.y), ratio_one<true>(x.z - y.z, x.z + y.z), ratio_one<true>(x.w - y.w, x.w + y.w));
            else o.v = make_float4(ratio_one<false>(x.x - y.x, x.x + y.x), ratio_one<false>(x.y - y.y, x.y + y.y), ratio_one<false>(x.z - y.z, x.z + y.z), ratio_one<false>(x.w - y.w, x.w + y.w));
        } else if (p.op == SARPRO_OP_SUM) o.v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        else o.v = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        return o;
    }
};
// Eight samples per lane: one 16-byte load per band of u16 operands (the four-sample form reads 8 bytes per lane and band).
template <> struct F32Vec<8> {
    float v[8];
    __device__ float get(int j) const { return v[j]; }
    struct Raw { uint4 a, b, c, d; }; // raster: a, b | f32 operands: a, b = first, c, d = second | u16 operands: a = first, c = second
    __device__ static Raw load_raw(const float *in, size_t pitch, const F32Pol &p, size_t r, size_t col) {
        Raw w;
        w.a = w.b = w.c = w.d = make_uint4(0, 0, 0, 0);
        if (p.op < 0) {
            const uint4 *q = reinterpret_cast<const uint4 *>(in + r * pitch + col);
            w.a = q[0]; w.b = q[1];
        } else if (p.u16) {
            w.a = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p.a) + r * p.pitch + col);
            w.c = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p.b) + r * p.pitch + col);
        } else {
            const uint4 *qa = reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(p.a) + r * p.pitch + col);
            const uint4 *qb = reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(p.b) + r * p.pitch + col);
            w.a = qa[0]; w.b = qa[1]; w.c = qb[0]; w.d = qb[1];
        }
        return w;
    }
    __device__ static F32Vec make(const Raw &w, const F32Pol &p) {
        F32Vec o;
        const uint32_t wa[8] = {w.a.x, w.a.y, w.a.z, w.a.w, w.b.x, w.b.y, w.b.z, w.b.w};
        const uint32_t wc[8] = {w.c.x, w.c.y, w.c.z, w.c.w, w.d.x, w.d.y, w.d.z, w.d.w};
        if (p.op < 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = __uint_as_float(wa[j]);
            return o;
        }
        float x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (p.u16) {
                x[j] = (float)((j & 1) ? (wa[j >> 1] >> 16) : (wa[j >> 1] & 0xFFFFu));
                y[j] = (float)((j & 1) ? (wc[j >> 1] >> 16) : (wc[j >> 1] & 0xFFFFu));
            } else { x[j] = __uint_as_float(wa[j]); y[j] = __uint_as_float(wc[j]); }
        }
        if (p.op == SARPRO_OP_RATIO || p.op == SARPRO_OP_LOGRATIO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = p.u16 ? ratio_one<true>(x[j], y[j]) : ratio_one<false>(x[j], y[j]);
        } else if (p.op == SARPRO_OP_NDIFF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = p.u16 ? ratio_one<true>(x[j] - y[j], x[j] + y[j]) : ratio_one<false>(x[j] - y[j], x[j] + y[j]);
        } else if (p.op == SARPRO_OP_SUM) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = x[j] + y[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = x[j] - y[j];
        }
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
    struct Raw { float x, y; };
    __device__ static Raw load_raw(const float *in, size_t pitch, const F32Pol &p, size_t r, size_t col) {
        Raw w{0.0f, 0.0f};
        if (p.op < 0) { w.x = in[r * pitch + col]; return w; }
        const size_t i = r * p.pitch + col;
        if (p.u16) { w.x = (float)reinterpret_cast<const uint16_t *>(p.a)[i]; w.y = (float)reinterpret_cast<const uint16_t *>(p.b)[i]; }
        else { w.x = reinterpret_cast<const float *>(p.a)[i]; w.y = reinterpret_cast<const float *>(p.b)[i]; }
        return w;
    }
    __device__ static F32Vec make(const Raw &w, const F32Pol &p) {
        F32Vec o;
        o.v = p.op < 0 ? w.x : pol_one(p.op, w.x, w.y);
        return o;
    }
};

// Walk of the grid-stride kernels over (row, vector column) items: no division per item, the next item's loads issued before the
// current item is worked on.
template <int VEC>
struct StrideWalk {
    uint64_t idx, total, step;
    uint32_t r, vc, vpr, step_r, step_c;
    typename F32Vec<VEC>::Raw cur;
    // (phase, nphase): this walk takes every nphase-th item of the lane's sequence, starting with the phase-th -- a kernel that
    // runs nphase walks side by side has nphase loads in flight per lane
    __device__ StrideWalk(uint32_t rows, uint32_t cols, const float *in, size_t pitch, const F32Pol &pol, uint32_t phase = 0, uint32_t nphase = 1) {
        vpr = (cols + VEC - 1) / VEC;
        total = (uint64_t)rows * vpr;
        const uint64_t step0 = (uint64_t)gridDim.x * blockDim.x;
        step = step0 * nphase;
        step_r = (uint32_t)(step / vpr); step_c = (uint32_t)(step % vpr);
        idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + step0 * phase;
        r = (uint32_t)(idx / vpr); vc = (uint32_t)(idx % vpr);
        if (idx < total) cur = F32Vec<VEC>::load_raw(in, pitch, pol, r, (size_t)vc * VEC);
    }
    __device__ bool live() const { return idx < total; }
    // returns the current item's samples and its (row, column); advances, prefetching the item after it
    __device__ F32Vec<VEC> next(const float *in, size_t pitch, const F32Pol &pol, uint32_t *row, uint32_t *col) {
        *row = r; *col = vc * VEC;
        const typename F32Vec<VEC>::Raw mine = cur;
        idx += step; r += step_r; vc += step_c;
        if (vc >= vpr) { vc -= vpr; ++r; }
        if (idx < total) cur = F32Vec<VEC>::load_raw(in, pitch, pol, r, (size_t)vc * VEC);
        return F32Vec<VEC>::make(mine, pol);
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
    StrideWalk<VEC> walk(rows, cols, in, pitch, pol);
    while (walk.live()) {
        uint32_t r, col;
        const F32Vec<VEC> v = walk.next(in, pitch, pol, &r, &col);
        // no branch between the samples of a vector (their logarithms overlap); a sample that does not count adds exact zeros, in
        // the same order as before: the partial sums keep their bits
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            const bool ok = col + j < cols && x >= t_valid;
            cnt += ok ? 1u : 0u;
            mn = fminf(mn, ok ? x : INFINITY);
            mx = fmaxf(mx, ok ? x : -INFINITY);
            if (MOMENTS) {
                const double db = db_of_f32_fast(ok ? x : 1.0f, logc, invc);
                sum += ok ? db : 0.0;
                sumsq += ok ? db * db : 0.0;
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
// a'. zone route (f32_kernels.h): row sample of the float's leading bits; min / max pass that also counts the samples at or
//     above each zone bound and keeps the samples inside a zone; count of the kept samples against a few thresholds.
// ------------------------------------------------------------------------------------
constexpr int kSampleBlock = 1024; // 128 KiB of LDS per workgroup: one per CU, so make it a full one
template <int VEC>
__global__ __launch_bounds__(kSampleBlock) void k_f32_sample_keys(const float *__restrict__ in, size_t pitch, uint32_t rows, uint32_t cols,
                                                            float t_valid, uint32_t row_stride, float *__restrict__ sample,
                                                            uint32_t sample_pitch, uint32_t *__restrict__ g_hist, F32Pol pol) {
    extern __shared__ uint32_t keys[]; // [kSampleKeys]
    for (int i = threadIdx.x; i < kSampleKeys; i += kSampleBlock) keys[i] = 0;
    __syncthreads();
    const uint32_t vpr = (cols + VEC - 1) / VEC;
    const uint32_t nsrows = (rows + row_stride - 1) / row_stride;
    for (uint32_t sr = blockIdx.x; sr < nsrows; sr += gridDim.x) {
        const uint32_t r = min(sr * row_stride + row_stride / 2, rows - 1); // mid-phase rows
        float *dst = sample + (size_t)sr * sample_pitch;
        for (uint32_t vc = threadIdx.x; vc < vpr; vc += kSampleBlock) {
            const uint32_t col = vc * VEC;
            const F32Vec<VEC> v = F32Vec<VEC>::fetch(in, pitch, pol, r, col);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float x = v.get(j);
                if (!(col + j < cols && x >= t_valid)) x = 0.0f; // invalid: below every threshold
                else atomicAdd(&keys[__float_as_uint(x) >> 16], 1u);
                dst[col + j] = x; // sample_pitch covers vpr * VEC
            }
        }
        // the scalar form stops at cols: the pad columns up to the sample's pitch (the second pass reads whole rows of it) must
        // not keep samples of an earlier call
        for (uint32_t c = vpr * VEC + threadIdx.x; c < sample_pitch; c += kSampleBlock) dst[c] = 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSampleKeys; i += kSampleBlock)
        if (keys[i]) atomicAdd(&g_hist[i], keys[i]);
}

// one workgroup of 1024: prefix sums over the key histogram, the bucket and base count of every probe rank
__global__ __launch_bounds__(1024) void k_f32_zone_pick(F32ZoneSelectArgs a) {
    __shared__ uint32_t incl[1024];
    __shared__ uint32_t s_kmin, s_kmax, s_rank[kMaxProbes];
    const int t = threadIdx.x;
    constexpr int per = kSampleKeys / 1024;
    static_assert(per == 32, "eight 16-byte loads per thread");
    uint32_t sum = 0, lo = 0xFFFFFFFFu, hi = 0;
    uint32_t hk[per];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.key_hist + t * per);
#pragma unroll
        for (int q = 0; q < per / 4; ++q) { const uint4 v = src[q]; hk[4 * q] = v.x; hk[4 * q + 1] = v.y; hk[4 * q + 2] = v.z; hk[4 * q + 3] = v.w; }
    }
#pragma unroll
    for (int k = 0; k < per; ++k) {
        sum += hk[k];
        if (hk[k]) { lo = min(lo, (uint32_t)(t * per + k)); hi = max(hi, (uint32_t)(t * per + k)); }
    }
    if (t == 0) { s_kmin = 0xFFFFFFFFu; s_kmax = 0; }
    // inclusive prefix over the 1024 threads: inside the wave by shuffles, across the 16 waves through LDS
    __shared__ uint32_t wave_tot[16];
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d); if ((t & 63) >= d) inc += v; }
    if ((t & 63) == 63) wave_tot[t >> 6] = inc;
    __syncthreads();
    if (lo != 0xFFFFFFFFu) { atomicMin(&s_kmin, lo); atomicMax(&s_kmax, hi); }
    {
        uint32_t before = 0;
        for (int w = 0; w < (t >> 6); ++w) before += wave_tot[w];
        inc += before;
    }
    incl[t] = inc;
    __syncthreads();
    const uint32_t ns = incl[1023];
    if (t == 0) {
        // rank error of a row sample: 6 sigma of a simple random sample of that size (finite-population corrected), never
        // below 0.2 % (rows are clusters: neighbouring samples are correlated); a zone that still misses costs the 4096-bin sweep
        const double f = (double)a.sample_fraction;
        const double delta = f >= 1.0 ? 0.0 : fmax(0.002, 6.0 * sqrt((1.0 - f) / (double)max(ns, 1u)));
        for (int i = 0; i < a.npcts; ++i) {
            const double n = (double)ns;
            s_rank[2 * i] = (uint32_t)floor(fmax(a.pcts[i] - delta, 0.0) * n);
            s_rank[2 * i + 1] = (uint32_t)fmin(ceil(fmin(a.pcts[i] + delta, 1.0) * n), n - 1.0);
        }
        a.work->ns = ns; a.work->kmin = s_kmin; a.work->kmax = s_kmax; a.work->nprobe = 2 * a.npcts;
        a.work->nz = 0;
    }
    __syncthreads();
    if (ns == 0) return;
    const uint32_t excl = incl[t] - sum;
    for (int i = 0; i < 2 * a.npcts; ++i) {
        const uint32_t r = s_rank[i];
        if (excl <= r && r < incl[t]) {
            uint32_t c = excl, cbase = 0;
            int found = -1;
#pragma unroll
            for (int k = 0; k < per; ++k) { // (registers, static indices: no dependent loads, no early exit)
                if (found < 0 && r < c + hk[k]) { found = k; cbase = c; }
                c += hk[k];
            }
            if (found >= 0) { a.work->probe_key[i] = t * per + found; a.work->probe_base[i] = cbase; a.work->probe_rank[i] = r; }
        }
    }
}

// the stored sample again: sub-bucket histograms of the probed keys
__global__ __launch_bounds__(kBlock) void k_f32_sample_sub(const float *__restrict__ sample, uint64_t n, float t_valid,
                                                           const F32ZoneWork *__restrict__ work, uint32_t *__restrict__ g_sub) {
    __shared__ uint32_t sub[kMaxProbes * kSubKeys];
    __shared__ uint32_t pk[kMaxProbes];
    __shared__ uint32_t probed[kSampleKeys / 32]; // bit k: key k is probed (most samples are not in a probed bucket)
    for (int i = threadIdx.x; i < kMaxProbes * kSubKeys; i += kBlock) sub[i] = 0;
    for (int i = threadIdx.x; i < kSampleKeys / 32; i += kBlock) probed[i] = 0;
    const int np = (int)work->nprobe;
    if ((int)threadIdx.x < kMaxProbes) pk[threadIdx.x] = (int)threadIdx.x < np ? work->probe_key[threadIdx.x] : 0xFFFFFFFFu;
    __syncthreads();
    if ((int)threadIdx.x < np) atomicOr(&probed[(pk[threadIdx.x] & (kSampleKeys - 1)) >> 5], 1u << (pk[threadIdx.x] & 31u));
    __syncthreads();
    auto one = [&](float x) {
        if (!(x >= t_valid)) return;
        const uint32_t bits = __float_as_uint(x), key = bits >> 16;
        if (key >= (uint32_t)kSampleKeys || !((probed[key >> 5] >> (key & 31u)) & 1u)) return;
#pragma unroll
        for (int p = 0; p < kMaxProbes; ++p)
            if (key == pk[p]) atomicAdd(&sub[p * kSubKeys + ((bits >> 7) & (kSubKeys - 1))], 1u); // equal probe keys: each keeps its own copy
    };
    // (the stored sample's pitch is a multiple of 4 and its base an allocation: whole vectors)
    const uint64_t n4 = n / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kBlock) {
        const float4 v = reinterpret_cast<const float4 *>(sample)[i];
        one(v.x); one(v.y); one(v.z); one(v.w);
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) one(sample[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < kMaxProbes * kSubKeys; i += kBlock)
        if (sub[i]) atomicAdd(&g_sub[i], sub[i]);
}

// one wave: the value at every probe rank to 2^-16, then the zones
__global__ __launch_bounds__(64) void k_f32_zone_finalize(F32ZoneSelectArgs a) {
    __shared__ uint32_t edge[kMaxProbes]; // bit pattern: lower edge of the probe's sub-bucket (even probes), upper edge (odd probes)
    F32ZoneWork *w = a.work;
    const int t = threadIdx.x;
    const int np = (int)w->nprobe;
    if (w->ns < 20000u || w->kmax >= 0x7F80u) {
        if (t == 0) { w->nz = 0; w->kbase = 0; w->nrun = 0; } // one gap, nothing kept: the sweep still counts
        if (t < 2 * kMaxZones) w->bounds[t] = INFINITY;
        return;
    }
    __shared__ uint32_t subl[kMaxProbes * kSubKeys];
    for (int i = t; i < np * kSubKeys; i += 64) subl[i] = a.sub_hist[i];
    __syncthreads();
    // the sub-bucket holding each probe rank: the wave scans a probe's 512 counts together, eight per lane
    static_assert(kSubKeys == 512, "eight sub-buckets per lane");
    for (int p = 0; p < np; ++p) {
        uint32_t h8[8], mine = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { h8[k] = subl[p * kSubKeys + t * 8 + k]; mine += h8[k]; }
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d); if (t >= d) inc += v; }
        const uint32_t r = w->probe_rank[p];
        uint32_t c = w->probe_base[p] + inc - mine; // counts below this lane's first sub-bucket
        // the FIRST sub-bucket k with r < c_k + h_k (ranks beyond the last one: the last sub-bucket)
        const bool here = r < c + mine; // (cumulative ends do not decrease: the lanes that say yes form a suffix)
        const unsigned long long who = __ballot(here);
        if (who ? (t == (int)__builtin_ctzll(who)) : (t == 63)) {
            uint32_t sb = who ? 0u : (uint32_t)kSubKeys - 1u;
            if (who) {
                sb = t * 8 + 7;
                for (int k = 0; k < 8; ++k) {
                    if (r < c + h8[k]) { sb = t * 8 + k; break; }
                    c += h8[k];
                }
            }
            const uint32_t base = (w->probe_key[p] << 16) | (sb << 7);
            edge[p] = (p & 1) ? base + (1u << 7) : base;
        }
    }
    __syncthreads();
    __shared__ float zl[kMaxZones], zh[kMaxZones];
    __shared__ int s_m;
    if (t == 0) {
        // a 4096-bin of the (estimated) span, in relative value; the zones are widened by 2.5 of them on either side
        const double lo_all = (double)__uint_as_float(w->kmin << 16), hi_all = (double)__uint_as_float((w->kmax + 1) << 16);
        const double span_db = 10.0 * log10(hi_all / lo_all);
        const double widen = pow(10.0, 2.5 * (span_db / 4096.0) / 10.0);
        int nz = 0;
        for (int q = 0; q < a.npcts; ++q) {
            float lo = (float)((double)__uint_as_float(edge[2 * q]) / widen);
            float hi = (float)((double)__uint_as_float(edge[2 * q + 1]) * widen);
            lo = __uint_as_float(__float_as_uint(lo) - 1u); // rounding of the two conversions: one step outwards
            hi = __uint_as_float(__float_as_uint(hi) + 1u);
            if (!(lo > a.t_valid)) lo = a.t_valid;
            if (!(hi < 3.0e38f)) hi = INFINITY;
            int i = nz++;
            while (i > 0 && zl[i - 1] > lo) { zl[i] = zl[i - 1]; zh[i] = zh[i - 1]; --i; }
            zl[i] = lo; zh[i] = hi;
        }
        int m = 0;
        for (int i = 0; i < nz; ++i) {
            if (m && zl[i] <= zh[m - 1]) zh[m - 1] = fmaxf(zh[m - 1], zh[i]);
            else { zl[m] = zl[i]; zh[m] = zh[i]; ++m; }
        }
        s_m = m;
    }
    __syncthreads();
    const int m = s_m;
    // The sweep classifies by the float's leading 16 bits: a bucket that touches a zone is KEPT whole (its samples go to the side
    // buffer, the count kernel sorts them against the exact thresholds), the buckets between two runs of kept buckets form a gap
    // whose samples are only counted.  Runs in ascending order; zones whose buckets touch share a run.
    __shared__ uint32_t run_s[kMaxZones], run_e[kMaxZones];
    __shared__ int s_nrun;
    if (t == 0) {
        const uint32_t kmid = (w->kmin + w->kmax) / 2;
        const uint32_t kbase = kmid > (uint32_t)kZoneLutKeys / 2 ? kmid - (uint32_t)kZoneLutKeys / 2 : 0u;
        const uint32_t klast = kbase + (uint32_t)kZoneLutKeys - 1;
        int nr = 0;
        for (int j = 0; j < m; ++j) {
            const uint32_t b0 = __float_as_uint(zl[j]), b1 = isinf(zh[j]) ? 0x7F800000u : __float_as_uint(zh[j]);
            const uint32_t ks = min(max(b0 >> 16, kbase), klast), ke = min(max((b1 - 1u) >> 16, kbase), klast);
            if (nr && ks <= run_e[nr - 1] + 1u) run_e[nr - 1] = max(run_e[nr - 1], ke);
            else { run_s[nr] = ks; run_e[nr] = ke; ++nr; }
            w->zone_run[j] = nr - 1;
        }
        for (int j = m; j < kMaxZones; ++j) w->zone_run[j] = -1;
        s_nrun = nr;
        w->kbase = kbase;
    }
    __syncthreads();
    const int nr = s_nrun;
    double mass = 0.0; // share of the sample in kept buckets
    for (int r = 0; r < nr; ++r)
        for (uint32_t k = run_s[r] + t; k <= run_e[r]; k += 64) mass += (double)a.key_hist[k];
    for (int d = 32; d > 0; d >>= 1) mass += __shfl_down(mass, d);
    if (t != 0) return;
    mass /= (double)w->ns;
    w->mass_est = (float)mass;
    for (int k = 0; k < 2 * kMaxZones; ++k) w->bounds[k] = INFINITY;
    // (a sample whose populated keys span more than the table: nothing sensible to select)
    w->nz = ((float)(1.3 * mass + 0.004) <= a.max_mass && w->kmax - w->kmin + 2u < (uint32_t)kZoneLutKeys) ? m : 0;
    w->nrun = w->nz > 0 ? nr : 0; // (0: the min / max pass keeps nothing)
    for (int r = 0; r < nr; ++r) { w->run_s[r] = run_s[r]; w->run_e[r] = run_e[r]; }
    if (w->nz > 0)
        for (int j = 0; j < m; ++j) { w->bounds[2 * j] = zl[j]; w->bounds[2 * j + 1] = zh[j]; }
}

// the sweep's class table from the runs: four keys per thread
__global__ __launch_bounds__(kBlock) void k_f32_zone_lut(const F32ZoneWork *__restrict__ w, uint32_t *__restrict__ lut) {
    const uint32_t i4 = blockIdx.x * kBlock + threadIdx.x;
    if (i4 >= (uint32_t)kZoneLutKeys / 4) return;
    const int nr = w->nrun;
    uint32_t word = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t k = w->kbase + 4 * i4 + b;
        uint32_t cls = 0;
        for (int r = 0; r < nr; ++r) {
            if (k > w->run_e[r]) cls += 8u;
            else if (k >= w->run_s[r]) { cls = 0x80u | 56u; break; }
        }
        word |= cls << (8 * b);
    }
    lut[i4] = word;
}

// 1 << (s mod 64): the hardware shift reads six bits of the amount, the language wants them masked first
__device__ inline unsigned long long shl64_one(uint32_t s) {
    unsigned long long r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "v"(s), "v"(1ull));
    return r;
}

// The min / max pass of the zone route.  Per sample: the class byte of its leading 16 bits (LDS table), one 64-bit add into
// eight packed 8-bit counters (one per gap; unpacked every 63 turns), and -- for a kept bucket -- the append to the wave's side
// buffer.  The cost does not depend on the number of zones (round 2 compared every sample with every bound: 0.53 / 0.71 / 0.79 ms
// for 2 / 4 / 5 zones).
// NW walks of VEC samples side by side per turn: <4, 2> for f32 rasters, <8, 1> for u16 operands (one 16-byte load per band and lane)
template <int VEC, int NW>
__global__ __launch_bounds__(kBlock) void k_f32_prepass_zones(F32ZoneArgs a) {
    // Valid samples are positive floats: they order like their bit patterns read as signed integers, an invalid one is replaced
    // by -1.0f (a negative integer).  Integer compares have no NaN twin, min / max are VOP2.
    __shared__ uint32_t lut[kZoneLutKeys / 4];
    constexpr uint32_t kRing = 1024; // kept samples on their way out: flushed 64 at a time, one coalesced store (a turn adds up to 64 x 2 VEC)
    __shared__ float ring[kWavesPerBlock][kRing];
    __shared__ unsigned long long gsum[8];
    for (int i = threadIdx.x; i < kZoneLutKeys / 4; i += kBlock) lut[i] = a.lut[i];
    if (threadIdx.x < 8) gsum[threadIdx.x] = 0;
    const int kbase = (int)a.work->kbase;
    __syncthreads();
    const uint8_t *lutb = reinterpret_cast<const uint8_t *>(lut);
    int mn = 0x7F800000, mx = (int)0x80000000;
    unsigned long long acc = 0;
    uint32_t gap[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) gap[k] = 0;
    // every WAVE appends to its own quarter of the workgroup's region: the cursor is a scalar, no atomics
    const uint32_t wcap = a.cap / kWavesPerBlock;
    float *mine = a.zone_buf + (size_t)blockIdx.x * a.cap + (size_t)wave_id() * wcap;
    float *myring = ring[wave_id()];
    uint32_t cursor = 0, flushed = 0;
    StrideWalk<VEC> walk(a.rows, a.cols, a.in, a.pitch, a.pol, 0, NW), walk_b(a.rows, a.cols, a.in, a.pitch, a.pol, NW - 1, NW);
    const bool whole_rows = a.cols % VEC == 0 && a.t_valid > 0.0f;
    // A wave-uniform trip count (lane 0 holds the wave's smallest item index, so it runs longest): the cursor stays scalar
    const uint64_t first = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u);
    // (two walks side by side, both items in every turn: 2 x VEC independent chains of divide -> table read -> counter add)
    constexpr int M = NW * VEC;
    static_assert(M <= 8 && NW <= 2, "31 turns of M samples must fit an 8-bit counter; the ring takes 64 x M per turn");
    const uint32_t nit = __builtin_amdgcn_readfirstlane(first < walk.total ? (uint32_t)((walk.total - first + walk.step - 1) / walk.step) : 0u);
    for (uint32_t it = 0; it < nit; ++it) {
        uint32_t r0 = 0, col0 = 0xFFFFFFF0u, r1 = 0, col1 = 0xFFFFFFF0u; // a lane past its last item: every sample fails the column test
        F32Vec<VEC> v0{}, v1{};
        if (walk.live()) v0 = walk.next(a.in, a.pitch, a.pol, &r0, &col0);
        if (NW == 2 && walk_b.live()) v1 = walk_b.next(a.in, a.pitch, a.pol, &r1, &col1);
        float xs[M];
        bool keep[M];
        unsigned long long zm[M], any = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            xs[j] = j < VEC ? v0.get(j % VEC) : v1.get(j % VEC);
            const uint32_t colj = (j < VEC ? col0 : col1) + (uint32_t)(j % VEC);
            // (whole vectors only -- cols % VEC == 0 -- need no column test: a lane past its last item holds zeros, which are invalid)
            const bool ok = (whole_rows || colj < a.cols) && xs[j] >= a.t_valid; // NaN fails
            const int xi = ok ? __float_as_int(xs[j]) : (int)0xBF800000;
            mn = (int)min((uint32_t)mn, (uint32_t)xi); // valid samples are non-negative integers, the invalid one is a huge unsigned
            mx = max(mx, xi);
            const int idx = min(max((xi >> 16) - kbase, 0), kZoneLutKeys - 1);
            const uint32_t cls = ok ? (uint32_t)lutb[idx] : 56u; // an invalid sample: counter 7, which nobody reads
            acc += shl64_one(cls);
            keep[j] = cls > 0x7Fu;
            zm[j] = __ballot(keep[j]);
            any |= zm[j];
        }
        if (any) {
            uint32_t base = cursor;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(zm[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)zm[j], 0u));
                if (keep[j]) myring[pos & (kRing - 1u)] = xs[j];
                base += (uint32_t)__popcll(zm[j]);
            }
            cursor = base;
            while (cursor - flushed >= 64u) { // the wave's own LDS writes are visible to it in program order
                const uint32_t p = flushed + (uint32_t)lane_id();
                if (p < wcap) mine[p] = myring[p & (kRing - 1u)];
                flushed += 64u;
            }
        }
        if (it % 31u == 30u || it + 1 == nit) { // 31 turns x 2 VEC samples < 256: no counter has overflowed
#pragma unroll
            for (int k = 0; k < 7; ++k) gap[k] += (uint32_t)(acc >> (8 * k)) & 0xFFu;
            acc = 0;
        }
    }
    {
        const uint32_t p = flushed + (uint32_t)lane_id();
        if (p < cursor && p < wcap) mine[p] = myring[p & (kRing - 1u)];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mn = min(mn, __shfl_xor(mn, d)); mx = max(mx, __shfl_xor(mx, d));
#pragma unroll
        for (int k = 0; k < 7; ++k) gap[k] += __shfl_xor(gap[k], d);
    }
    __shared__ int s_mn[kWavesPerBlock], s_mx[kWavesPerBlock];
    if (lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&gsum[k], (unsigned long long)gap[k]);
        s_mn[wave_id()] = mn; s_mx[wave_id()] = mx;
        a.zone_n[blockIdx.x * kWavesPerBlock + wave_id()] = cursor;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) { mn = min(mn, s_mn[w]); mx = max(mx, s_mx[w]); }
        a.partials[blockIdx.x] = F32Partial{0, 0.0, 0.0, mn == 0x7F800000 ? INFINITY : __int_as_float(mn), mx < 0 ? -INFINITY : __int_as_float(mx)};
    }
    if (threadIdx.x < 8) a.gap_counts[(size_t)blockIdx.x * 8 + threadIdx.x] = gsum[threadIdx.x];
}

// the min / max pass's per-workgroup results reduced into the host's mailbox (one workgroup of 1024: two turns for the pass's 2048)
constexpr int kPostBlock = 1024;
__global__ __launch_bounds__(kPostBlock) void k_f32_zone_post(F32ZoneArgs a, int grid, F32ZoneMail *mail, uint32_t *flag, uint32_t seq) {
    __shared__ unsigned long long s_sum[2 * kMaxZones + 2];
    __shared__ float s_mn[kPostBlock / 64], s_mx[kPostBlock / 64];
    __shared__ uint32_t s_over;
    if (threadIdx.x < 2 * kMaxZones + 2) s_sum[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_over = 0;
    __syncthreads();
    float mn = INFINITY, mx = -INFINITY;
    unsigned long long kept = 0, ge[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) ge[z] = 0;
    uint32_t over = 0;
    const uint32_t wcap = a.cap / kWavesPerBlock;
    static_assert(kWavesPerBlock == 4 && sizeof(F32Partial) == 32, "vector loads below");
    for (int i = threadIdx.x; i < grid; i += kPostBlock) { // one workgroup of the pass per thread and turn: all its loads in flight together
        const F32Partial p = a.partials[i];
        const uint4 zn = reinterpret_cast<const uint4 *>(a.zone_n)[i];
        const ulonglong2 *gp = reinterpret_cast<const ulonglong2 *>(a.gap_counts + (size_t)i * 8);
#pragma unroll
        for (int z = 0; z < 4; ++z) { const ulonglong2 g2 = gp[z]; ge[2 * z] += g2.x; ge[2 * z + 1] += g2.y; }
        mn = fminf(mn, p.minv); mx = fmaxf(mx, p.maxv);
        kept += (unsigned long long)zn.x + zn.y + zn.z + zn.w;
        over |= (zn.x > wcap) | (zn.y > wcap) | (zn.z > wcap) | (zn.w > wcap);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d)); mx = fmaxf(mx, __shfl_xor(mx, d));
        kept += __shfl_xor(kept, d); over |= __shfl_xor(over, d);
#pragma unroll
        for (int z = 0; z < 8; ++z) ge[z] += __shfl_xor(ge[z], d);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int z = 0; z < 8; ++z)
            if (ge[z]) atomicAdd(&s_sum[z], ge[z]);
        atomicAdd(&s_sum[2 * kMaxZones + 1], kept);
        if (over) s_over = 1;
        s_mn[wave_id()] = mn; s_mx[wave_id()] = mx;
    }
    __syncthreads();
    if (threadIdx.x < 8) mail->gap[threadIdx.x] = s_sum[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + sizeof(F32ZoneWork) / 4) // (a second wave copies the zone record, word by word)
        reinterpret_cast<uint32_t *>(&mail->work)[threadIdx.x - 64] = reinterpret_cast<const uint32_t *>(a.work)[threadIdx.x - 64];
    if (threadIdx.x == 0) {
        for (int w = 1; w < kPostBlock / 64; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
        mail->kept = s_sum[2 * kMaxZones + 1];
        mail->min_v = mn; mail->max_v = mx;
        mail->overflow = s_over; mail->pad = 0;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// device words -> the host's mailbox (one workgroup; a few KiB at most)
__global__ __launch_bounds__(kPostBlock) void k_post(PostSegs g, uint32_t *flag, uint32_t seq) {
    for (int k = 0; k < g.n; ++k) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(g.src[k]);
        uint32_t *dst = reinterpret_cast<uint32_t *>(g.dst[k]);
        for (uint32_t i = threadIdx.x; i < g.bytes[k] / 4; i += kPostBlock) dst[i] = src[i];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(kBlock) void k_prep(PrepSegs g) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x, nt = gridDim.x * kBlock;
    for (int k = 0; k < g.n; ++k) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(g.src[k]);
        uint32_t *dst = reinterpret_cast<uint32_t *>(g.dst[k]);
        for (uint32_t i = t; i < g.bytes[k] / 4; i += nt) dst[i] = src[i];
    }
    for (int k = 0; k < g.nz; ++k) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(g.zero[k]);
        for (uint32_t i = t; i < g.zbytes[k] / 4; i += nt) dst[i] = 0u;
    }
}

// self-test of div_small_ints: every pair (a, b) of u16 values, the ratio a / b (b != 0) and the normalised difference
// (a - b) / (a + b) (a + b != 0) against the compiler's IEEE division; counts the pairs that differ
__global__ __launch_bounds__(256) void k_selftest_div_small_ints(unsigned long long *mismatches) {
    const uint32_t a = blockIdx.x;          // 65536 blocks
    uint32_t bad = 0;
    for (uint32_t b = threadIdx.x; b < 65536u; b += 256) {
        const float x = (float)a, y = (float)b;
        if (b != 0 && __float_as_uint(div_small_ints(x, y)) != __float_as_uint(x / y)) ++bad;
        const float s = x + y, d = x - y;
        if (s != 0.0f && __float_as_uint(div_small_ints(d, s)) != __float_as_uint(d / s)) ++bad;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

__global__ __launch_bounds__(kBlock) void k_f32_zone_count(const float *__restrict__ zone_buf, const uint32_t *__restrict__ zone_n, uint32_t cap,
                                                           int nregions, const float *__restrict__ g_thr, int nthr,
                                                           unsigned long long *__restrict__ g_counts) {
    __shared__ float thr[kZoneMaxThr + 2];
    __shared__ uint32_t hist[kZoneMaxThr + 1];
    for (int i = threadIdx.x; i <= kZoneMaxThr + 1; i += kBlock) thr[i] = (i >= 1 && i <= nthr) ? g_thr[i] : (i ? INFINITY : -INFINITY);
    for (int i = threadIdx.x; i <= kZoneMaxThr; i += kBlock) hist[i] = 0;
    __syncthreads();
    // a region (one wave's side buffer of the sweep) per WAVE and turn, four loads in flight per lane: the regions are short
    // (a few thousand samples) and a workgroup looping over one of them waited on every load
    auto count_one = [&](float v) {
        // (a handful of zones holds some tens of thresholds, not 1023: as many search steps as their number needs)
        const uint32_t k = nthr <= 63 ? step_search<63>(thr, v) : (nthr <= 255 ? step_search<255>(thr, v) : step_search<kZoneMaxThr>(thr, v));
        atomicAdd(&hist[k], 1u);
    };
    for (int w = blockIdx.x * kWavesPerBlock + wave_id(); w < nregions; w += gridDim.x * kWavesPerBlock) {
        const uint32_t n = min(zone_n[w], cap);
        const float *src = zone_buf + (size_t)w * cap;
        uint32_t i = lane_id();
        for (; i + 192 < n; i += 256) {
            const float v0 = src[i], v1 = src[i + 64], v2 = src[i + 128], v3 = src[i + 192];
            count_one(v0); count_one(v1); count_one(v2); count_one(v3);
        }
        for (; i < n; i += 64) count_one(src[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= kZoneMaxThr; i += kBlock)
        if (hist[i]) atomicAdd(&g_counts[i], (unsigned long long)hist[i]);
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
    StrideWalk<VEC> walk(rows, cols, in, pitch, pol);
    while (walk.live()) {
        uint32_t r, col;
        const F32Vec<VEC> v = walk.next(in, pitch, pol, &r, &col);
        // the VEC samples side by side: estimates, table reads and verifications of all of them together (est_search_m)
        float xs[VEC];
        bool ok[VEC];
        uint32_t bin[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            ok[j] = col + j < cols && x >= t_valid;
            xs[j] = ok[j] ? x : 1.0f; // (a sample that does not count still gets a bin: any finite value will do)
        }
        if (est.use) est_search_m<4095, VEC>(thr, xs, est, bin);
        else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) bin[j] = step_search<4095>(thr, xs[j]);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (ok[j]) atomicAdd(&hist[bin[j]], 1u);
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
// MODE 0: threshold table (estimate + verification).  1: the level in f64 with the reference's expression.  2: the same for
// gamma == 1, folded: y = A log2(x) + B with A log2(c_i) + B in the table and A in the series' coefficients (degree 4: the
// remainder r^5 / (5 ln 2) < 1e-14 of log2, 2e-9 levels at the narrowest window; the margin below is 1e-6).
// (A template parameter, not a flag: the f64 forms' registers cost the table form three waves of eight per SIMD.)
template <int VEC, bool OUT16, int MODE>
__global__ __launch_bounds__(kBlock) void k_f32_level(F32LevelArgs a) {
    constexpr bool F64 = MODE != 0;
    __shared__ __align__(8) float2 thr_p[256]; // (thr[k], thr[k + 1]) at [k]: one 8-byte read verifies an estimate
    __shared__ uint32_t hist[256];
    const ThrPairs thr{thr_p};
    if (!OUT16) {
        if (a.thr) thr_pairs_fill(thr_p, a.thr, 256, threadIdx.x);
        else thr_p[threadIdx.x] = make_float2(-INFINITY, threadIdx.x == 255 ? INFINITY : -INFINITY);
        hist[threadIdx.x] = 0;
    }
    __shared__ double logc[F64 ? 256 : 1], invc[F64 ? 256 : 1]; // table of db_of_f32_fast (f64 levels)
    if (F64) {
        const double c = 1.0 + ((double)threadIdx.x + 0.5) / 256.0;
        logc[threadIdx.x] = MODE == 2 ? fma(a.lin_a, log2(c), a.lin_b) : log2(c);
        invc[threadIdx.x] = 1.0 / c;
    }
    // log2(1 + r) = r (1 - r / 2 + r^2 / 3 - r^3 / 4) / ln 2, times A
    const double k1 = a.lin_a * 1.4426950408889634, k2 = k1 * -0.5, k3 = k1 * (1.0 / 3.0), k4 = k1 * -0.25;
    __syncthreads();
    const bool vec_store = a.out_pitch % VEC == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
    uint32_t zeros = 0;
    StrideWalk<VEC> walk(a.rows, a.cols, a.in, a.in_pitch, a.pol);
    while (walk.live()) {
        uint32_t r, col;
        const F32Vec<VEC> v = walk.next(a.in, a.in_pitch, a.pol, &r, &col);
        const uint32_t r_row = r;
        const double inv_range = 1.0 / a.range;
        uint32_t lvs[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            uint32_t lv = 0;
            if (col + j < a.cols && x >= a.t_valid) {
                if (F64) { // (u8 output takes this form on the small-scene direct route: no threshold table at all)
                    if (x >= a.t_last) lv = (uint32_t)a.max_val;
                    else if (x < a.t_first) lv = 0u;
                    else {
                        double y;
                        if (MODE == 2) {
                            const uint32_t bits = __float_as_uint(x), mant = bits & 0x7FFFFFu;
                            const int e = (int)(bits >> 23) - 127;
                            const double m = __hiloint2double((int)(0x3FF00000u | (mant >> 3)), (int)(mant << 29)); // 1.mant, exactly
                            const double rr = fma(m, invc[mant >> 15], -1.0);
                            y = fma(rr, fma(rr, fma(rr, fma(rr, k4, k3), k2), k1), fma(a.lin_a, (double)e, logc[mant >> 15]));
                            y = fmin(fmax(y, 0.0), a.lin_ymax);
                        } else {
                            const double db = db_of_f32_fast(x, logc, invc); // within ~1e-14 of glibc's value: far inside the 1e-6 margin below
                            // a multiplication where the reference divides by `range`: a relative 1e-16, 1e-11 levels, far inside the margin
                            const double t = (fmin(fmax(db, a.low), a.high) - a.low) * inv_range;
                            y = fmin(fmax((a.gamma == 1.0 ? t : pow(t, a.gamma)) * a.max_val, 0.0), a.max_val);
                        }
                        const double r = rint(y);
                        if (fabs(y - r) < 1e-6 || !(y == y)) {
                            if (a.f64_levels == 2) {
                                const uint32_t q = atomicAdd(a.uq_count, 1u);
                                if (q < a.uq_cap) a.uq_entries[q] = make_uint4(r_row, col + j, __float_as_uint(x), (uint32_t)y); // .w: the provisional level
                                lv = (uint32_t)y | 0x80000000u; // provisional: the host patches it (and counts it: bit 31 keeps it out of the histogram)
                            } else lv = OUT16 ? step_search<65535>(a.thr, x) : step_search<255>(thr, x);
                        } else lv = (uint32_t)y;
                    }
                } else if (a.est.use) lv = OUT16 ? est_search<65535>(a.thr, x, a.est) : est_search<255>(thr, x, a.est);
                else lv = OUT16 ? step_search<65535>(a.thr, x) : step_search<255>(thr, x);
            }
            const bool queued_lv = (lv & 0x80000000u) != 0u;
            lv &= 0x7FFFFFFFu;
            lvs[j] = lv;
            if (!OUT16 && col + j < a.cols && !queued_lv) { if (lv == 0) ++zeros; else atomicAdd(&hist[lv], 1u); }
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
    __shared__ __align__(8) float2 thr_p[256]; // (thr[k], thr[k + 1]) at [k]: one 8-byte read verifies an estimate
    __shared__ uint32_t hist[256];
    thr_pairs_fill(thr_p, a.thr, 256, threadIdx.x);
    const ThrPairs thr{thr_p};
    hist[threadIdx.x] = 0;
    __syncthreads();
    const Rect rc = a.rects[blockIdx.x];
    const int col = rc.cstart + lane_id() * VEC;
    if (col < rc.c1 && col + VEC > rc.c0) {
        // TWO rows per turn (rows r and r + 4 of the wave's sequence), their 2 x VEC chains of divide -> log -> table reads -> LDS add
        // side by side: the pass is bound by the latency of one such chain, not by instruction count or HBM
        constexpr int S = kWavesPerBlock;
        bool own[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) own[j] = col + j >= rc.c0 && col + j < rc.c1;
        int r = rc.r0 + wave_id();
        typename F32Vec<VEC>::Raw cur0{}, cur1{};
        if (r < rc.r1) cur0 = F32Vec<VEC>::load_raw(a.in, a.pitch, a.pol, r, col);
        if (r + S < rc.r1) cur1 = F32Vec<VEC>::load_raw(a.in, a.pitch, a.pol, r + S, col);
        for (; r < rc.r1; r += 2 * S) {
            const typename F32Vec<VEC>::Raw m0 = cur0, m1 = cur1;
            const bool second = r + S < rc.r1;
            if (r + 2 * S < rc.r1) cur0 = F32Vec<VEC>::load_raw(a.in, a.pitch, a.pol, r + 2 * S, col); // the next turn's rows in flight
            if (r + 3 * S < rc.r1) cur1 = F32Vec<VEC>::load_raw(a.in, a.pitch, a.pol, r + 3 * S, col);
            const F32Vec<VEC> v0 = F32Vec<VEC>::make(m0, a.pol), v1 = F32Vec<VEC>::make(m1, a.pol);
            uint32_t bin[2 * VEC];
            bool ok[2 * VEC];
            float xs[2 * VEC];
#pragma unroll
            for (int j = 0; j < 2 * VEC; ++j) {
                const float x = j < VEC ? v0.get(j % VEC) : v1.get(j % VEC);
                ok[j] = own[j % VEC] && x >= a.t_valid && (j < VEC || second);
                xs[j] = ok[j] ? x : 1.0f; // (a sample that does not count still gets a bin: any finite value will do)
            }
            if (a.est.use) est_search_m<255, 2 * VEC>(thr, xs, a.est, bin);
            else {
#pragma unroll
                for (int j = 0; j < 2 * VEC; ++j) bin[j] = step_search<255>(thr, xs[j]);
            }
#pragma unroll
            for (int j = 0; j < 2 * VEC; ++j)
                if (ok[j]) atomicAdd(&hist[bin[j]], 1u);
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
    __shared__ __align__(16) double2 cdf2[2 * 256]; // (c00, c01) at [bin], (c10, c11) at [256 + bin]: two 16-byte arrays, every bank in use (kernels.hip 4)
    __shared__ __align__(8) float2 thr_p[256]; // (thr[k], thr[k + 1]) at [k]: one 8-byte read verifies an estimate
    __shared__ uint32_t hist[256];
    const ThrPairs thr{thr_p};
    const Rect rc = a.rects[blockIdx.x];
    {
        const int b = threadIdx.x;
        cdf2[b] = make_double2(a.cdfs[(size_t)rc.id[0] * 256 + b], a.cdfs[(size_t)rc.id[1] * 256 + b]);
        cdf2[256 + b] = make_double2(a.cdfs[(size_t)rc.id[2] * 256 + b], a.cdfs[(size_t)rc.id[3] * 256 + b]);
        thr_pairs_fill(thr_p, a.thr, 256, b);
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
        // The VEC samples of a row side by side with no branch between them -- bins (estimates verified together), then the f64
        // blends: the pass is bound by the latency of one sample's chain (divide, log, two table reads, 32 bytes of CDF from LDS, ten
        // dependent f64 operations), not by instruction count: 0.64 -> 0.53 ms at 400 MP.  (R rows per turn: R = 2 needs the
        // registers of eight f64 chains and runs at 0.73.)
        constexpr int R = 1, S = kWavesPerBlock, M = R * VEC;
        bool own[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) own[j] = col + j >= rc.c0 && col + j < rc.c1;
        const bool whole = col >= rc.c0 && col + VEC <= rc.c1;
        int r = rc.r0 + wave_id();
        typename F32Vec<VEC>::Raw cur[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { cur[k] = typename F32Vec<VEC>::Raw{}; if (r + k * S < rc.r1) cur[k] = F32Vec<VEC>::load_raw(a.in, a.in_pitch, a.pol, r + k * S, col); }
        for (; r < rc.r1; r += R * S) {
            typename F32Vec<VEC>::Raw mine[R];
            RowWeight rw[R];
            bool rowok[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                mine[k] = cur[k];
                rowok[k] = r + k * S < rc.r1;
                if (r + (R + k) * S < rc.r1) cur[k] = F32Vec<VEC>::load_raw(a.in, a.in_pitch, a.pol, r + (R + k) * S, col); // the next turn's rows in flight
                rw[k] = a.row_w[min(r + k * S, (int)rc.r1 - 1)];
            }
            uint32_t lvs[M], bins[M];
            float xs[M];
            bool ok[M];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const F32Vec<VEC> v = F32Vec<VEC>::make(mine[k], a.pol);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float x = v.get(j);
                    ok[k * VEC + j] = rowok[k] && own[j] && x >= a.t_valid;
                    xs[k * VEC + j] = ok[k * VEC + j] ? x : 1.0f; // (a sample that does not count still gets a bin: any finite value will do)
                }
            }
            if (a.est.use) est_search_m<255, M>(thr, xs, a.est, bins);
            else {
#pragma unroll
                for (int j = 0; j < M; ++j) bins[j] = step_search<255>(thr, xs[j]);
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const int k = i / VEC, j = i % VEC;
                const double2 ct = cdf2[bins[i]], cb = cdf2[256 + bins[i]];
                const double top = ct.x * omdx[j] + ct.y * dx[j];
                const double bottom = cb.x * omdx[j] + cb.y * dx[j];
                double o = top * rw[k].omd + bottom * rw[k].d;
                o = fmin(fmax(o, 0.0), 1.0);
                lvs[i] = ok[i] ? (uint32_t)(o * a.max_val) : 0u;
                if (!OUT16 && rowok[k] && own[j]) { if (lvs[i] == 0) ++zeros; else atomicAdd(&hist[lvs[i]], 1u); }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (!rowok[k]) continue;
                const size_t row = (size_t)(r + k * S);
                if (whole) {
                    store_levels<VEC, OUT16>(a.out, row * a.out_pitch + col, &lvs[k * VEC], VEC, vec_store);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if (!own[j]) continue;
                        if (OUT16) reinterpret_cast<uint16_t *>(a.out)[row * a.out_pitch + col + j] = (uint16_t)lvs[k * VEC + j];
                        else reinterpret_cast<uint8_t *>(a.out)[row * a.out_pitch + col + j] = (uint8_t)lvs[k * VEC + j];
                    }
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

// e'. The same for u8 output, speculative: the blend in f32 from a 16-byte entry (c00, c10, c01 - c00, c11 - c10) per bin decides
//     the level unless o * 255 lies within the margin of an integer; those samples take the reference's f64 sequence.  The
//     construction, its error bound and the margins are those of the u16 flavour's kernel (kernels.hip 4b: 2^-12 in interior
//     cells, 2^-10 where the cell extrapolates; all-zero and, in interior cells, all-one bins get biased entries that never come
//     near an integer; an invalid sample reads entry 256 = "all zero").  What it saves here: 32 bytes of f64 CDFs per sample from
//     LDS (half of it bank conflicts) and ten f64 operations.  SARPRO_HIP_NO_SPEC=1: the f64 kernel above.
#ifdef SARPRO_SPEC_DELTA_R3
constexpr float kF32SpecDeltaEdge = 1.0f / 1024.0f, kF32SpecDeltaInner = 1.0f / 4096.0f;
#else
constexpr float kF32SpecDeltaEdge = 6.2e-4f, kF32SpecDeltaInner = 1.6e-4f; // (kernels.hip 4b: 1.17x what the error bound requires)
#endif
constexpr float kF32SpecDeltaEdgeY = 3.0e-4f, kF32SpecDeltaEdgeX = 2.6e-4f; // cells that extrapolate along one axis only (kernels.hip 4b)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_f32_clahe_apply_spec(F32ClaheApplyArgs a) {
    __shared__ __align__(16) double2 cdf2[2 * 256]; // (c00, c01) at [bin], (c10, c11) at [256 + bin]: two 16-byte arrays, every bank in use (kernels.hip 4)
    __shared__ __align__(16) float4 e32[256 + 1];
    __shared__ __align__(8) float2 thr_p[256]; // (thr[k], thr[k + 1]) at [k]: one 8-byte read verifies an estimate
    __shared__ uint32_t hist[256];
    const ThrPairs thr{thr_p};
    const Rect rc = a.rects[blockIdx.x];
    const bool edge = (rc.pad[0] & 1) != 0;
    {
        const int b = threadIdx.x;
        double c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = a.cdfs[(size_t)rc.id[k] * 256 + b];
        cdf2[b] = make_double2(c[0], c[1]);
        cdf2[256 + b] = make_double2(c[2], c[3]);
        const bool saturated = c[0] == 1.0 && c[1] == 1.0 && c[2] == 1.0 && c[3] == 1.0 && !edge;
        const bool zero = c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0 && c[3] == 0.0;
        const float kz = 0.5f / 255.0f;
        const float c00 = (float)c[0], c01 = (float)c[1], c10 = (float)c[2], c11 = (float)c[3];
        e32[b] = saturated ? make_float4(1.001f, 1.001f, 0.0f, 0.0f) : zero ? make_float4(kz, kz, 0.0f, 0.0f) : make_float4(c00, c10, c01 - c00, c11 - c10);
        if (b == 0) e32[256] = make_float4(kz, kz, 0.0f, 0.0f);
        thr_pairs_fill(thr_p, a.thr, 256, b);
        hist[b] = 0;
    }
    __syncthreads();
    const int col = rc.cstart + lane_id() * VEC;
    const bool lane_on = col < rc.c1 && col + VEC > rc.c0;
    const bool vec_store = a.out_pitch % VEC == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7) == 0;
#ifdef SARPRO_SPEC_DELTA_R3
    const float near_delta = edge ? kF32SpecDeltaEdge : kF32SpecDeltaInner;
#else
    const float near_delta = !edge ? kF32SpecDeltaInner : (rc.pad[0] & 6) == 2 ? kF32SpecDeltaEdgeY : (rc.pad[0] & 6) == 4 ? kF32SpecDeltaEdgeX : kF32SpecDeltaEdge;
#endif
    const float bias = -0.5f - near_delta, two_delta = 2.0f * near_delta;
    uint32_t zeros = 0;
    double dx[VEC], omdx[VEC];
    float dxf[VEC];
    bool own[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = col + j;
        own[j] = c >= rc.c0 && c < rc.c1;
        const RowWeight w = a.col_w[own[j] ? c : rc.c0];
        dx[j] = w.d; omdx[j] = w.omd; dxf[j] = (float)w.d;
    }
    if (lane_on) {
        int r = rc.r0 + wave_id();
        typename F32Vec<VEC>::Raw cur{};
        if (r < rc.r1) cur = F32Vec<VEC>::load_raw(a.in, a.in_pitch, a.pol, r, col);
        for (; r < rc.r1; r += kWavesPerBlock) {
            const typename F32Vec<VEC>::Raw mine = cur;
            if (r + kWavesPerBlock < rc.r1) cur = F32Vec<VEC>::load_raw(a.in, a.in_pitch, a.pol, r + kWavesPerBlock, col); // next row in flight
            const F32Vec<VEC> v = F32Vec<VEC>::make(mine, a.pol);
            const RowWeight rw = a.row_w[r];
            const float wy1 = (float)rw.omd * 255.0f, wy2 = (float)rw.d * 255.0f;
            uint32_t lvs[VEC], bins[VEC], flagged = 0;
            float xs[VEC];
            bool valid[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float x = v.get(j);
                valid[j] = own[j] && x >= a.t_valid;
                xs[j] = valid[j] ? x : 1.0f;
            }
            if (a.est.use) est_search_m<255, VEC>(thr, xs, a.est, bins); // estimates of all four verified together
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) bins[j] = step_search<255>(thr, xs[j]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) { // the four samples side by side: entries, f32 blends
                bins[j] = valid[j] ? bins[j] : 256u;
                const float4 e = e32[bins[j]];
                const float top = fmaf(e.z, dxf[j], e.x), bottom = fmaf(e.w, dxf[j], e.y);
                const float ya = fmaf(bottom, wy2, fmaf(top, wy1, bias));
                const uint32_t la = __builtin_amdgcn_cvt_pk_u8_f32(ya, 0, 0u), lb = __builtin_amdgcn_cvt_pk_u8_f32(ya + two_delta, 0, 0u);
                lvs[j] = la;
                flagged |= (la != lb ? 1u : 0u) << j;
            }
            if (flagged) { // within the margin of an integer: the reference's own sequence decides
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (!((flagged >> j) & 1u)) continue;
                    const double2 ct = cdf2[bins[j]], cb = cdf2[256 + bins[j]]; // (a flagged sample is valid: entry 256 is never near)
                    const double top = ct.x * omdx[j] + ct.y * dx[j];
                    const double bottom = cb.x * omdx[j] + cb.y * dx[j];
                    double o = top * rw.omd + bottom * rw.d;
                    o = fmin(fmax(o, 0.0), 1.0);
                    lvs[j] = (uint32_t)(o * 255.0);
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (own[j]) { if (lvs[j] == 0) ++zeros; else atomicAdd(&hist[lvs[j]], 1u); }
            if (col >= rc.c0 && col + VEC <= rc.c1) {
                store_levels<VEC, false>(a.out, (size_t)r * a.out_pitch + col, lvs, VEC, vec_store);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (own[j]) reinterpret_cast<uint8_t *>(a.out)[(size_t)r * a.out_pitch + col + j] = (uint8_t)lvs[j];
            }
        }
    }
    if (zeros) atomicAdd(&hist[0], zeros);
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&a.level_hist[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
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

// Kernels that end by flushing a block-private histogram with global atomics (256 or 4096 hot words shared by every block): on small
// rasters the flush, not the sweep, is the kernel -- 4096 blocks x 256 adds on 256 addresses took 0.05 ms for a 4 MP band.  At least 32
// vectors per thread before another block is worth its flush (SARPRO_HIP_HIST_GRID_VPT: tuning switch).
inline int hist_grid_vectors_per_thread() { return 8; }
inline int hist_grid(uint64_t items, int per_cu = 8) {
    const uint64_t by_work = items / ((uint64_t)kBlock * (uint64_t)hist_grid_vectors_per_thread());
    const int g = stream_grid(items, per_cu);
    return (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g, std::max<uint64_t>(by_work, 1)));
}

} // namespace

hipError_t launch_f32_sample_keys(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, uint32_t row_stride,
                                  float *d_sample, uint32_t sample_pitch, uint32_t *d_key_hist, hipStream_t s, const F32Pol &pol) {
    const size_t lds = sizeof(uint32_t) * kSampleKeys;
    const uint32_t nsrows = (rows + row_stride - 1) / row_stride;
    const int grid = (int)std::min<uint32_t>(nsrows, 256u);
    hipError_t e;
    if (vec) {
        if ((e = opt_in_dynamic_lds(reinterpret_cast<const void *>(&k_f32_sample_keys<4>))) != hipSuccess) return e;
        hipLaunchKernelGGL((k_f32_sample_keys<4>), dim3(grid), dim3(kSampleBlock), lds, s, in, pitch, rows, cols, t_valid, row_stride, d_sample, sample_pitch,
                           d_key_hist, pol);
    } else {
        if ((e = opt_in_dynamic_lds(reinterpret_cast<const void *>(&k_f32_sample_keys<1>))) != hipSuccess) return e;
        hipLaunchKernelGGL((k_f32_sample_keys<1>), dim3(grid), dim3(kSampleBlock), lds, s, in, pitch, rows, cols, t_valid, row_stride, d_sample, sample_pitch,
                           d_key_hist, pol);
    }
    return hipGetLastError();
}

hipError_t launch_selftest_div_small_ints(unsigned long long *d_mismatches, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_div_small_ints, dim3(65536), dim3(256), 0, s, d_mismatches);
    return hipGetLastError();
}

hipError_t launch_f32_zone_pick(const F32ZoneSelectArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_zone_pick, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_f32_sample_sub(const float *d_sample, uint64_t n, float t_valid, const F32ZoneWork *work, uint32_t *d_sub_hist, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_sample_sub, dim3(stream_grid(n, 4)), dim3(kBlock), 0, s, d_sample, n, t_valid, work, d_sub_hist);
    return hipGetLastError();
}

hipError_t launch_f32_zone_finalize(const F32ZoneSelectArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_zone_finalize, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_f32_zone_lut, dim3(kZoneLutKeys / 4 / kBlock), dim3(kBlock), 0, s, a.work, reinterpret_cast<uint32_t *>(a.lut));
    return hipGetLastError();
}

hipError_t launch_f32_prepass_zones(const F32ZoneArgs &a, bool vec, int grid, hipStream_t s) {
    const bool v8 = vec && a.pol.op >= 0 && a.pol.u16 && a.pol.pitch % 8 == 0 && !a.no_vec8;
    if (v8) hipLaunchKernelGGL((k_f32_prepass_zones<8, 1>), dim3(grid), dim3(kBlock), 0, s, a);
    else if (vec) hipLaunchKernelGGL((k_f32_prepass_zones<4, 2>), dim3(grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((k_f32_prepass_zones<1, 2>), dim3(grid), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_f32_zone_post(const F32ZoneArgs &a, int grid, F32ZoneMail *mail, uint32_t *flag, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_zone_post, dim3(1), dim3(kPostBlock), 0, s, a, grid, mail, flag, seq);
    return hipGetLastError();
}
hipError_t launch_prep(const PrepSegs &segs, hipStream_t s) {
    uint64_t words = 0;
    for (int k = 0; k < segs.n; ++k) words += segs.bytes[k] / 4;
    for (int k = 0; k < segs.nz; ++k) words += segs.zbytes[k] / 4;
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(k_prep, dim3((unsigned)std::min<uint64_t>(64, (words + kBlock * 4 - 1) / (kBlock * 4))), dim3(kBlock), 0, s, segs);
    return hipGetLastError();
}
hipError_t launch_post(const PostSegs &segs, uint32_t *flag, uint32_t seq, hipStream_t s) {
    hipLaunchKernelGGL(k_post, dim3(1), dim3(kPostBlock), 0, s, segs, flag, seq);
    return hipGetLastError();
}

hipError_t launch_f32_zone_count(const float *zone_buf, const uint32_t *zone_n, uint32_t cap, int nregions, const float *d_thr, int nthr,
                                 unsigned long long *d_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_zone_count, dim3(std::min(nregions, 1024)), dim3(kBlock), 0, s, zone_buf, zone_n, cap, nregions, d_thr, nthr, d_counts);
    return hipGetLastError();
}

// The zone sweep's turn is one long dependent chain (divide, table read, packed-counter add): it needs more waves per SIMD than
// the plain min / max pass to hide it -- 400 MP, log-ratio of u16 bands: 0.81 / 0.50 / 0.44 / 0.46 ms with 2 / 4 / 6 / 8 workgroups per CU
int f32_zone_grid(uint32_t rows, uint32_t cols, bool vec) {
    const int per_cu = 6; // (<= 8: 2048 partials)
    const int V = vec ? 4 : 1;
    return stream_grid((uint64_t)rows * ((cols + V - 1) / V), per_cu);
}

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
    dim3 grid(hist_grid((uint64_t)rows * ((cols + V - 1) / V), 4));
    if (vec) hipLaunchKernelGGL(k_f32_hist4096<4>, grid, dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_thr, d_hist, est, pol);
    else hipLaunchKernelGGL(k_f32_hist4096<1>, grid, dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_thr, d_hist, est, pol);
    return hipGetLastError();
}

namespace {
__global__ void k_patch_u8(uint8_t *out, size_t pitch, const uint4 *patches, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)patches[i].x * pitch + patches[i].y] = (uint8_t)patches[i].z;
}

// ------------------------------------------------------------------------------------
// b'. 4096-bin histogram WITHOUT host-built thresholds (small scenes: the 4095 thresholds cost the host 0.35 ms of glibc, the
//     zone route ~0.1 ms of single-workgroup kernels -- both more than the scene's own passes).  Every block first merges the
//     pre-pass partials (min / max sample), then bins with the reference's own expression (autoscale.rs:108-117) in f64:
//     t = clamp((dB - min_dB) * inv_span, 0, 1), idx = min(floor(t * 4096), 4095), with the device's dB (within ~1e-14 of
//     glibc's: table + series, as the moments use).  The bin is accepted when t * 4096 is farther than 1e-6 from an integer --
//     the device / glibc difference moves it by ~1e-10 -- and the sample is QUEUED otherwise (a handful per scene) for the host
//     to bin with glibc.  Samples equal to the minimum / maximum sample are bins 0 / 4095 by construction (t = 0, t = 1).
// ------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_f32_hist4096_direct(const float *__restrict__ in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid,
                                                                const F32Partial *__restrict__ partials, int nparts, unsigned long long *__restrict__ g_hist,
                                                                uint32_t *uq_count, uint4 *uq_entries, uint32_t uq_cap, F32Pol pol) {
    __shared__ uint32_t hist[4096];
    __shared__ double logc[256], invc[256];
    __shared__ float s_mn[kBlock], s_mx[kBlock];
    for (int i = threadIdx.x; i < 4096; i += kBlock) hist[i] = 0;
    {
        const double c = 1.0 + ((double)threadIdx.x + 0.5) / 256.0;
        logc[threadIdx.x] = log2(c);
        invc[threadIdx.x] = 1.0 / c;
        float mn = INFINITY, mx = -INFINITY;
        for (int i = threadIdx.x; i < nparts; i += kBlock) { mn = fminf(mn, partials[i].minv); mx = fmaxf(mx, partials[i].maxv); }
        s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
    }
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { s_mn[threadIdx.x] = fminf(s_mn[threadIdx.x], s_mn[threadIdx.x + st]); s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + st]); }
        __syncthreads();
    }
    const float xmin = s_mn[0], xmax = s_mx[0];
    if (!(xmax > xmin) || !(xmax < INFINITY)) return; // empty, single-valued or +inf: the host's degenerate arms need no bins
    const double min_db = db_of_f32_fast(xmin, logc, invc), max_db = db_of_f32_fast(xmax, logc, invc);
    const double inv_span = 1.0 / (max_db - min_db);
    StrideWalk<VEC> walk(rows, cols, in, pitch, pol);
    while (walk.live()) {
        uint32_t r, col;
        const F32Vec<VEC> v = walk.next(in, pitch, pol, &r, &col);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x = v.get(j);
            if (!(col + j < cols && x >= t_valid)) continue;
            uint32_t idx;
            if (x == xmin) idx = 0u;
            else if (x == xmax) idx = 4095u;
            else {
                const double t = fmin(fmax((db_of_f32_fast(x, logc, invc) - min_db) * inv_span, 0.0), 1.0);
                const double y = t * 4096.0;
                if (fabs(y - rint(y)) < 1e-6 || !(y == y)) {
                    const uint32_t q = atomicAdd(uq_count, 1u);
                    if (q < uq_cap) uq_entries[q] = make_uint4(r, col + j, __float_as_uint(x), 0u);
                    continue;
                }
                idx = min((uint32_t)y, 4095u);
            }
            atomicAdd(&hist[idx], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += kBlock)
        if (hist[i]) atomicAdd(&g_hist[i], (unsigned long long)hist[i]);
}

__global__ void k_patch_u16(uint16_t *out, size_t pitch, const uint4 *patches, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)patches[i].x * pitch + patches[i].y] = (uint16_t)patches[i].z;
}
} // namespace

hipError_t launch_patch_u8(uint8_t *out, size_t pitch, const uint4 *d_patches, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_patch_u8, dim3((n + 255) / 256), dim3(256), 0, s, out, pitch, d_patches, n);
    return hipGetLastError();
}

hipError_t launch_patch_u16(uint16_t *out, size_t pitch, const uint4 *d_patches, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_patch_u16, dim3((n + 255) / 256), dim3(256), 0, s, out, pitch, d_patches, n);
    return hipGetLastError();
}

hipError_t launch_f32_hist4096_direct(const float *in, size_t pitch, uint32_t rows, uint32_t cols, float t_valid, bool vec, const F32Partial *d_partials,
                                      int nparts, unsigned long long *d_hist, uint32_t *d_uq_count, uint4 *d_uq_entries, uint32_t uq_cap, hipStream_t s,
                                      const F32Pol &pol) {
    const int V = vec ? 4 : 1;
    dim3 grid(hist_grid((uint64_t)rows * ((cols + V - 1) / V), 4));
    if (vec) hipLaunchKernelGGL(k_f32_hist4096_direct<4>, grid, dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_partials, nparts, d_hist, d_uq_count, d_uq_entries, uq_cap, pol);
    else hipLaunchKernelGGL(k_f32_hist4096_direct<1>, grid, dim3(kBlock), 0, s, in, pitch, rows, cols, t_valid, d_partials, nparts, d_hist, d_uq_count, d_uq_entries, uq_cap, pol);
    return hipGetLastError();
}

hipError_t launch_f32_level(const F32LevelArgs &a, bool vec, bool out16, hipStream_t s) {
    const int V = vec ? 4 : 1;
    dim3 grid(out16 ? stream_grid((uint64_t)a.rows * ((a.cols + V - 1) / V)) : hist_grid((uint64_t)a.rows * ((a.cols + V - 1) / V)));
    const int mode = a.f64_levels == 0 ? 0 : (a.lin ? 2 : 1);
#define SARPRO_LEVEL_LAUNCH(V_, O_)                                                                              \
    do {                                                                                                        \
        if (mode == 2) hipLaunchKernelGGL((k_f32_level<V_, O_, 2>), grid, dim3(kBlock), 0, s, a);               \
        else if (mode == 1) hipLaunchKernelGGL((k_f32_level<V_, O_, 1>), grid, dim3(kBlock), 0, s, a);          \
        else hipLaunchKernelGGL((k_f32_level<V_, O_, 0>), grid, dim3(kBlock), 0, s, a);                         \
    } while (0)
    if (vec) {
        if (out16) SARPRO_LEVEL_LAUNCH(4, true);
        else SARPRO_LEVEL_LAUNCH(4, false);
    } else {
        if (out16) SARPRO_LEVEL_LAUNCH(1, true);
        else SARPRO_LEVEL_LAUNCH(1, false);
    }
#undef SARPRO_LEVEL_LAUNCH
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
    if (!out16 && a.max_val == 255.0 && !a.no_spec) {
        if (vec) hipLaunchKernelGGL((k_f32_clahe_apply_spec<4>), dim3(nrects), dim3(kBlock), 0, s, a);
        else hipLaunchKernelGGL((k_f32_clahe_apply_spec<1>), dim3(nrects), dim3(kBlock), 0, s, a);
        return hipGetLastError();
    }
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

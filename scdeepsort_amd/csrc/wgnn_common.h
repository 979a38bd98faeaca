// wgnn_common.h - device helpers shared by the aggregation kernels (row-wave and LDS-tiled).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "wgnn.h"

namespace wgnn {


constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

enum { EPI_FWD = 0, EPI_BWD_SRC = 1, EPI_BWD_ALPHA = 2 };

struct KArgs {
    const void* rowptr;                          // int32[R+1], or int64[R+1] when flags & WGNN_FLAG_ROWPTR_I64
    const int* col; const float* val;
    const float* cs1; const float* cs2;          // optional per-column scale factors (alpha[col], inv_deg[col])
    const void* src; long ld_src;                // gathered rows
    const float* alpha; int mode; int self_idx;
    const void* self; long ld_self;              // EPI_FWD: h_self; EPI_BWD_SRC: h_src rows (for dalpha); EPI_BWD_ALPHA: h_self
    const int* row_ids; const float* inv_deg; const float* bias;
    void* out; long ld_out;
    const float* g; long ld_g;                   // EPI_BWD_ALPHA: upstream gradient rows
    float* aux1; float* aux2;                    // neigh_sum (FWD, optional) / dalpha (BWD_SRC) / dalpha_row, dself_row (BWD_ALPHA)
    int D; unsigned flags; int accumulate;
    const int4* items; long n_items;
    const int4* long_rows; long n_long;
    float* partials;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __half* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&u.x);
    const __half2 b = *reinterpret_cast<const __half2*>(&u.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(__half* p, float4 v) {
    __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 u; u.x = *reinterpret_cast<unsigned*>(&a); u.y = *reinterpret_cast<unsigned*>(&b);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void fma4(float4& acc, float w, const float4& x) {
    acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y);
    acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// sum over the LPR lanes of a lane group (all 64 lanes participate)
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// in-degree of CSR row r from the caller's row-pointer array (32- or 64-bit entries)
__device__ __forceinline__ float row_degree(const KArgs& a, long r) {
    if (a.flags & WGNN_FLAG_ROWPTR_I64) {
        const long long* p = reinterpret_cast<const long long*>(a.rowptr);
        return (float)(p[r + 1] - p[r]);
    }
    const int* p = reinterpret_cast<const int*>(a.rowptr);
    return (float)(p[r + 1] - p[r]);
}

// Shared epilogue: turns the accumulated neighbour sum of one row into the kernel's outputs.
template <int LPR, int NV, typename TIn, typename TOut, int EPI>
__device__ __forceinline__ void epilogue(const KArgs& a, float4 (&acc)[NV], int slot, int l, bool writer) {
    const int r = a.row_ids ? a.row_ids[slot] : slot;
    float invd = 1.0f;
    if (!(a.flags & WGNN_FLAG_NO_MEAN)) {
        invd = a.inv_deg ? a.inv_deg[r] : 1.0f / (row_degree(a, r) + 1.0f);
    }
    if constexpr (EPI == EPI_FWD) {
        const float rs = invd * (a.mode == WGNN_DST_IS_GENE ? a.alpha[r] : 1.0f);
        const bool has_self = !(a.flags & WGNN_FLAG_NO_SELF) && a.self != nullptr;
        const float sc = has_self ? invd * (a.mode == WGNN_NO_ALPHA ? 1.0f : a.alpha[a.self_idx]) : 0.0f;
        const float post = (a.flags & WGNN_FLAG_OUT_SCALE_ALPHA) && a.mode == WGNN_DST_IS_GENE ? a.alpha[r] : 1.0f;
        const TIn* selfp = reinterpret_cast<const TIn*>(a.self) +
                           (size_t)((a.flags & WGNN_FLAG_SELF_COMPACT) ? slot : r) * a.ld_self;
        TOut* outp = reinterpret_cast<TOut*>(a.out) + (size_t)slot * a.ld_out;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c0 = (k * LPR + l) * 4;
            if (writer && c0 < a.D) {
                float4 o = acc[k];
                if (a.aux1) st4(a.aux1 + (size_t)slot * a.D + c0, o);           // raw neighbour sum, saved for dalpha
                o.x *= rs; o.y *= rs; o.z *= rs; o.w *= rs;
                if (has_self) fma4(o, sc, ld4(selfp + c0));
                if (a.bias) { const float4 b = ld4(a.bias + c0); o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
                if (a.flags & WGNN_FLAG_RELU) {
                    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                }
                if (a.flags & WGNN_FLAG_OUT_SCALE_ALPHA) { o.x *= post; o.y *= post; o.z *= post; o.w *= post; }
                st4(outp + c0, o);
            }
        }
    } else if constexpr (EPI == EPI_BWD_SRC) {
        // slot == source row s.  acc = T[s] = sum_r t_val * colscale[r] * g[r]
        const float rs = (a.mode == WGNN_SRC_IS_GENE) ? a.alpha[r] : 1.0f;
        float* outp = reinterpret_cast<float*>(a.out) + (size_t)slot * a.ld_out;
        float dot = 0.f;
        const bool want_dalpha = (a.mode == WGNN_SRC_IS_GENE) && a.aux1 && a.self;
        const float* hp = reinterpret_cast<const float*>(a.self) + (size_t)r * a.ld_self;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c0 = (k * LPR + l) * 4;
            if (c0 < a.D) {
                if (want_dalpha && writer) dot += dot4(acc[k], ld4(hp + c0));
                if (writer) {
                    float4 o = acc[k];
                    o.x *= rs; o.y *= rs; o.z *= rs; o.w *= rs;
                    if (a.accumulate) { const float4 p = ld4(outp + c0); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
                    st4(outp + c0, o);
                }
            }
        }
        if (want_dalpha) {                         // wave-uniform branch
            dot = group_sum<LPR>(dot);
            if (writer && l == 0) a.aux1[r] = (a.accumulate ? a.aux1[r] : 0.f) + dot;
        }
    } else {                                       // EPI_BWD_ALPHA: acc = S[i] = sum_j val_j h_src[col_j]
        const float* gp = a.g + (size_t)slot * a.ld_g;
        const float* sp = reinterpret_cast<const float*>(a.self) +
                          (size_t)((a.flags & WGNN_FLAG_SELF_COMPACT) ? slot : r) * a.ld_self;
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c0 = (k * LPR + l) * 4;
            if (writer && c0 < a.D) {
                const float4 gv = ld4(gp + c0);
                d1 += dot4(gv, acc[k]);
                if (a.self) d2 += dot4(gv, ld4(sp + c0));
            }
        }
        d1 = group_sum<LPR>(d1); d2 = group_sum<LPR>(d2);
        if (writer && l == 0) {
            if (a.aux1) a.aux1[slot] = invd * d1;
            if (a.aux2) a.aux2[slot] = invd * d2;
        }
    }
}


inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace wgnn

// wgnn_linear.hip - the dense half of a layer on the fp32 matrix cores (gfx950):
//
//     out = act( X . W^T + bias )          NodeUpdate.forward = relu(fc_neigh(neigh))   (reference models/gnn.py:18-25)
//                                          classifier head    = linear(h)               (reference models/gnn.py:66-67)
//
// Why it exists: (1) the C ABI (include/wgnn.h) is otherwise not self-sufficient for one whole layer - a non-torch
// caller would have to bring its own GEMM; (2) SURVEY 8b lists `wgnn_agg_linear_relu_fwd` (composed here: aggregation launch + this GEMM); (3) north_star
// allows MFMA where the dense feature x weight projection matters: at BASELINE cfg3 the projections are 40 GFLOP =
// ~11 % of a forward.  v_mfma_f32_32x32x2_f32 is EXACT fp32 (bitwise an fmaf chain in k order, MI355X_MICROARCH.md) at
// the fp32 vector rate (157 TF chip peak), so parity with the oracle's fp32 Linear needs no tolerance beyond summation
// order.
//
// Kernel: 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA blocks of
// 32 x 32, 64 accumulator VGPRs), K walked in slabs of 16 staged in LDS (row stride 17 floats: the per-lane fragment
// reads `As[(row)*17 + k]` hit 32 distinct banks), next slab's global loads issued before the current slab's MFMAs,
// two LDS buffers, one barrier per slab.  M, N, K need only be multiples of 4 (K) / 1 (M, N): edges are guarded.
#include <algorithm>
#include "wgnn_common.h"

namespace {
using namespace wgnn;

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef WGNN_LIN_BK
#define WGNN_LIN_BK 16
#endif
constexpr int kBM = 128, kBN = 128, kBK = WGNN_LIN_BK, kLd = kBK + 1;      // (the forward kernel's tile height is a template parameter)
constexpr unsigned WGNN_LIN_FORCE_64 = 1u << 16, WGNN_LIN_FORCE_128 = 1u << 17;   // timing switches of wgnn_linear_fwd_ex
constexpr int kLinThreads = 256;

struct LinArgs {
    const void* x; long ld_x;            // f32 or f16 rows (TX); f16 is widened in registers on the way into LDS
    const float* w; long ld_w;
    const float* bias;
    float* out; long ld_out;             // may be null when only the scaled copy is wanted
    const float* row_scale;              // optional [M]: out2[m, :] = row_scale[m] * act(x W^T + bias)[m, :]
    float* out2; long ld_out2;
    long M; int N; int K; unsigned flags;
};

// TX = float | __half: storage type of X.  fp16-stored node features (BASELINE cfg5) are read as 8-byte groups of four
// halves and converted in registers (fp16-rounded inputs, fp32 multiply-accumulate) - no fp32 copy of [C, 400] in HBM.
// out2: a second, row-scaled copy of the result written from the same accumulators - the projected gene table P_g and
// its alpha-folded form alpha[g] * P_g[g] (the source table of the LDS-streamed cells<-genes pass, models/gnn.py:54's
// (h * alpha) factor) come out of ONE kernel instead of GEMM + scale_rows.
// MI = 32-row MFMA blocks per wave along M: 2 -> 128 x 128 tiles (wave 64 x 64), 1 -> 64 x 128 tiles (wave 32 x 64).  The
// smaller tile is for operands whose 128-row tiling leaves the last round of workgroups thin (20k rows = 157 x 2 tiles on
// 256 CUs): twice the tiles, up to 6 workgroups per CU.
template <typename TX, bool DUAL, int MI>
__global__ void __launch_bounds__(kLinThreads, 4) linear_mfma_f32(const LinArgs a) {
    constexpr int kBM = 64 * MI;
    __shared__ float As[2][kBM * kLd];
    __shared__ float Ws[2][kBN * kLd];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // column tiles vary fastest: the workgroups that share a row tile of X run together
    const long m0 = (long)(blockIdx.x / ((a.N + kBN - 1) / kBN)) * kBM;
    const int n0 = (int)(blockIdx.x % ((a.N + kBN - 1) / kBN)) * kBN;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // global -> registers: thread t moves 2 float4 of X and 2 of W per slab (rows t/4 and t/4 + 64, k offset (t%4)*4)
    constexpr int kTPR = kBK / 4;                              // threads per tile row (one float4 each)
    constexpr int kRPP = kLinThreads / kTPR;                   // tile rows covered per pass
    constexpr int kNPA = kBM / kRPP, kNPW = kBN / kRPP;        // passes over the X tile / the W tile
    const int lr = t / kTPR, lk = (t % kTPR) * 4;
    float4 xa[kNPA], wa[kNPW];
    auto gload = [&](int k0) {
        const bool kin = k0 + lk < a.K;                        // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
        for (int i = 0; i < kNPA; ++i) {
            const long row = m0 + lr + kRPP * i;
            xa[i] = (row < a.M && kin) ? ld4(reinterpret_cast<const TX*>(a.x) + row * a.ld_x + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < kNPW; ++i) {
            const int col = n0 + lr + kRPP * i;
            wa[i] = (col < a.N && kin) ? ld4(a.w + (long)col * a.ld_w + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kNPA; ++i) {
            float* pa = &As[buf][(lr + kRPP * i) * kLd + lk];
            pa[0] = xa[i].x; pa[1] = xa[i].y; pa[2] = xa[i].z; pa[3] = xa[i].w;
        }
#pragma unroll
        for (int i = 0; i < kNPW; ++i) {
            float* pw = &Ws[buf][(lr + kRPP * i) * kLd + lk];
            pw[0] = wa[i].x; pw[1] = wa[i].y; pw[2] = wa[i].z; pw[3] = wa[i].w;
        }
    };

    const int nslab = (a.K + kBK - 1) / kBK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;                  // fragment: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload((s + 1) * kBK);               // in flight during this slab's MFMAs
        const float* pa = &As[buf][(wm * 32 * MI + fr) * kLd + fk];
        const float* pw = &Ws[buf][(wn * 64 + fr) * kLd + fk];
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const float b0 = pw[kk], b1 = pw[32 * kLd + kk];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float ai = pa[i * 32 * kLd + kk];
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, b1, acc[i][1], 0, 0, 0);
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);                    // the other buffer: last read two slabs ago
        __syncthreads();
    }

    // C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool relu = a.flags & WGNN_FLAG_RELU;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + fr;
            if (col >= a.N) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < a.M) {
                    float v = acc[i][j][r] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    if constexpr (DUAL) {
                        if (a.out) a.out[row * a.ld_out + col] = v;
                        a.out2[row * a.ld_out2 + col] = a.row_scale[row] * v;
                    } else {
                        a.out[row * a.ld_out + col] = v;
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the dense half:  dW[N, K] = sum_m g[m, N] * x[m, K]   (autograd of fc_neigh, train.py:84).
// The reduction runs over the NODE axis (1e5 rows at cfg3) and the output is small (256 x 400), so the M axis is cut into
// `n_slabs` slabs: workgroup (tile, slab) accumulates one 128 x 128 output tile over its slab on the fp32 matrix cores and
// writes a partial; `wgrad_reduce` folds the slabs in fixed order (deterministic, no atomics).  The library GEMM the
// framework picks for this shape (MT32x256x64, no split along M) takes 0.9 ms at cfg3 = 22 TF.
// ---------------------------------------------------------------------------------------------
struct WgArgs {
    const float* g; long ld_g;
    const float* x; long ld_x;
    float* partial;                // [n_slabs, N, K]
    long M; int N; int K; int n_slabs; long slab_rows;
};

__global__ void __launch_bounds__(kLinThreads) wgrad_mfma_f32(const WgArgs a) {
    __shared__ float As[2][kBM * kLd];           // g tile, stored [n][m]
    __shared__ float Bs[2][kBN * kLd];           // x tile, stored [k][m]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = (a.K + kBN - 1) / kBN, tiles_n = (a.N + kBM - 1) / kBM;
    const int tile = blockIdx.x % (tiles_k * tiles_n), slab = blockIdx.x / (tiles_k * tiles_n);
    const int n0 = (tile / tiles_k) * kBM, k0 = (tile % tiles_k) * kBN;
    const long m_begin = (long)slab * a.slab_rows, m_end = min(a.M, m_begin + a.slab_rows);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // thread t moves 2 float4 of g and 2 of x per step of 16 rows: row t/32 (+8), columns (t%32)*4 .. +3
    const int lm = t >> 5, lc = (t & 31) * 4;
    float4 ga[2], xa[2];
    auto gload = [&](long m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long m = m0 + lm + 8 * i;
            const bool in = m < m_end;
            ga[i] = (in && n0 + lc < a.N) ? ld4(a.g + m * a.ld_g + n0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);   // N, K % 4 == 0
            xa[i] = (in && k0 + lc < a.K) ? ld4(a.x + m * a.ld_x + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* pa = &As[buf][lc * kLd + lm + 8 * i];
            pa[0] = ga[i].x; pa[kLd] = ga[i].y; pa[2 * kLd] = ga[i].z; pa[3 * kLd] = ga[i].w;
            float* pb = &Bs[buf][lc * kLd + lm + 8 * i];
            pb[0] = xa[i].x; pb[kLd] = xa[i].y; pb[2 * kLd] = xa[i].z; pb[3 * kLd] = xa[i].w;
        }
    };
    const long nstep = (m_end - m_begin + kBK - 1) / kBK;
    const int fr = lane & 31, fk = lane >> 5;
    if (nstep > 0) {
        gload(m_begin);
        lstore(0);
        __syncthreads();
        for (long s = 0; s < nstep; ++s) {
            const int buf = (int)(s & 1);
            if (s + 1 < nstep) gload(m_begin + (s + 1) * kBK);
            const float* pa = &As[buf][(wm * 64 + fr) * kLd + fk];
            const float* pb = &Bs[buf][(wn * 64 + fr) * kLd + fk];
#pragma unroll
            for (int kk = 0; kk < kBK; kk += 2) {
                const float a0 = pa[kk], a1 = pa[32 * kLd + kk];
                const float b0 = pb[kk], b1 = pb[32 * kLd + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (s + 1 < nstep) lstore(buf ^ 1);
            __syncthreads();
        }
    }
    float* out = a.partial + (size_t)slab * a.N * a.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = k0 + wn * 64 + j * 32 + fr;
            if (col >= a.K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < a.N) out[(size_t)row * a.K + col] = acc[i][j][r];
            }
        }
}

__global__ void __launch_bounds__(256) wgrad_reduce(const float* __restrict__ partial, float* __restrict__ out, long ld_out,
                                                    int N, int K, int n_slabs, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * K) return;
    float s = 0.f;
    for (int p = 0; p < n_slabs; ++p) s += partial[(size_t)p * N * K + i];
    float* o = out + (i / K) * ld_out + (i % K);
    *o = accumulate ? *o + s : s;
}

}  // namespace

extern "C" int wgnn_linear_wgrad_workspace(int64_t M, int32_t N, int32_t K, int64_t* n_slabs, int64_t* bytes) {
    if (M < 0 || N <= 0 || K <= 0 || !n_slabs || !bytes) return WGNN_ERR_BAD_ARG;
    const long tiles = ((N + kBM - 1) / kBM) * ((K + kBN - 1) / kBN);
    long s = (4L * 256 + tiles - 1) / tiles;                 // ~4 workgroups per CU
    s = std::max(1L, std::min(s, (long)((M + 511) / 512)));      // but at least 512 rows per slab
    // whole rounds of workgroups over the 256 CUs (round 6): tiles x slabs that is not a multiple of 256 leaves a thin last
    // round - 4 tiles x 196 slabs = 784 workgroups ran as 3 full rounds + 16 stragglers (the 256 x 256 gradient at 0.42 of
    // the matrix peak), 8 x 40 = 320 as one round + a quarter (20 000 rows: 0.28)
    long g = tiles, r = 256;
    while (r) { const long t = g % r; g = r; r = t; }            // gcd(tiles, 256)
    const long q = 256 / g;                                      // slabs per whole round
    if (s >= q) s = (s / q) * q;
    *n_slabs = s;
    *bytes = s * (int64_t)N * K * 4;
    return WGNN_OK;
}

extern "C" int wgnn_linear_wgrad(const float* g, int64_t ld_g, const float* x, int64_t ld_x, float* dW, int64_t ld_dw,
                                 int64_t M, int32_t N, int32_t K, int accumulate, float* workspace, int64_t n_slabs,
                                 void* stream) {
    if (!g || !x || !dW || M < 0 || N <= 0 || K <= 0) return WGNN_ERR_BAD_ARG;
    if (N % 4 || K % 4 || ld_g % 4 || ld_x % 4 || !aligned16(g) || !aligned16(x)) return WGNN_ERR_ALIGNMENT;
    if (ld_g < N || ld_x < K || ld_dw < K) return WGNN_ERR_BAD_ARG;
    if (!workspace || n_slabs <= 0) return WGNN_ERR_WORKSPACE;
    const long tiles = ((N + kBM - 1) / kBM) * ((K + kBN - 1) / kBN);
    const long slab_rows = std::max(1L, (long)((M + n_slabs - 1) / n_slabs));
    hipStream_t st = static_cast<hipStream_t>(stream);
    WgArgs a{g, (long)ld_g, x, (long)ld_x, workspace, (long)M, N, K, (int)n_slabs, ((slab_rows + kBK - 1) / kBK) * kBK};
    hipLaunchKernelGGL(wgrad_mfma_f32, dim3((unsigned)(tiles * n_slabs)), dim3(kLinThreads), 0, st, a);
    const long n = (long)N * K;
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, dW, (long)ld_dw, N, K,
                       (int)n_slabs, accumulate);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

extern "C" int wgnn_linear_fwd_ex(const void* x, int x_dtype, int64_t ld_x, const float* w, int64_t ld_w, const float* bias,
                                  float* out, int64_t ld_out, const float* row_scale, float* out_scaled, int64_t ld_out_scaled,
                                  int64_t M, int32_t N, int32_t K, uint32_t flags, void* stream) {
    if (!x || !w || (!out && !out_scaled) || M < 0 || N <= 0 || K <= 0) return WGNN_ERR_BAD_ARG;
    if (flags & ~(WGNN_FLAG_RELU | WGNN_LIN_FORCE_64 | WGNN_LIN_FORCE_128)) return WGNN_ERR_BAD_ARG;
    if (x_dtype != WGNN_F32 && x_dtype != WGNN_F16) return WGNN_ERR_BAD_ARG;
    if ((out_scaled != nullptr) != (row_scale != nullptr)) return WGNN_ERR_BAD_ARG;
    if (K % 4 || ld_x % 4 || ld_w % 4 || !aligned16(w)) return WGNN_ERR_ALIGNMENT;
    if (x_dtype == WGNN_F32 ? !aligned16(x) : !aligned8(x)) return WGNN_ERR_ALIGNMENT;
    if (ld_x < K || ld_w < K || (out && ld_out < N) || (out_scaled && ld_out_scaled < N)) return WGNN_ERR_BAD_ARG;
    if (M == 0) return WGNN_OK;
    // tile height: 128 rows, or 64 when that fills the chip's last round of workgroups better (4 resident workgroups per CU
    // at 128 rows, 6 at 64)
    const long col_tiles = (N + kBN - 1) / kBN;
    const long t128 = ((M + 127) / 128) * col_tiles, t64 = ((M + 63) / 64) * col_tiles;
    auto round_fill = [](long tiles, long slots) { const long r = (tiles + slots - 1) / slots; return (double)tiles / (double)(r * slots); };
    const bool small = (flags & WGNN_LIN_FORCE_64) || (!(flags & WGNN_LIN_FORCE_128) && round_fill(t64, 256 * 6) > round_fill(t128, 256 * 4) + 0.05);
    const long tiles = small ? t64 : t128;
    if (tiles > 0x7FFFFFFFL) return WGNN_ERR_UNSUPPORTED;
    LinArgs a{x, (long)ld_x, w, (long)ld_w, bias, out, (long)ld_out, row_scale, out_scaled, (long)ld_out_scaled, (long)M, N, K,
              flags & WGNN_FLAG_RELU};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)tiles), block(kLinThreads);
#define WGNN_LIN_LAUNCH(TX, DUAL)                                                                              \
    do {                                                                                                       \
        if (small) hipLaunchKernelGGL((linear_mfma_f32<TX, DUAL, 1>), grid, block, 0, st, a);                  \
        else hipLaunchKernelGGL((linear_mfma_f32<TX, DUAL, 2>), grid, block, 0, st, a);                        \
    } while (0)
    if (x_dtype == WGNN_F16) {
        if (out_scaled) WGNN_LIN_LAUNCH(__half, true); else WGNN_LIN_LAUNCH(__half, false);
    } else {
        if (out_scaled) WGNN_LIN_LAUNCH(float, true); else WGNN_LIN_LAUNCH(float, false);
    }
#undef WGNN_LIN_LAUNCH
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

extern "C" int wgnn_linear_fwd(const float* x, int64_t ld_x, const float* w, int64_t ld_w, const float* bias,
                               float* out, int64_t ld_out, int64_t M, int32_t N, int32_t K, uint32_t flags, void* stream) {
    if (!out) return WGNN_ERR_BAD_ARG;
    return wgnn_linear_fwd_ex(x, WGNN_F32, ld_x, w, ld_w, bias, out, ld_out, nullptr, nullptr, 0, M, N, K, flags, stream);
}

// One whole reference layer on a block, in the reference's literal order (aggregate, then NodeUpdate):
//     neigh = block_compute(message_func, fn.mean)        (gnn.py:47-56,65)    -> K1 into `neigh_scratch`
//     out   = relu(fc_neigh(neigh))                        (gnn.py:18-25)       -> wgnn_linear_fwd
extern "C" int wgnn_agg_linear_relu_fwd(const void* rowptr, const int32_t* col, const float* val,
                                        const float* alpha, int alpha_mode, int32_t self_idx,
                                        const float* h_src, int64_t ld_src, const float* h_self, int64_t ld_self,
                                        const int32_t* row_ids, const float* inv_deg,
                                        int64_t n_out, int32_t D, uint32_t agg_flags,
                                        const int32_t* items, int64_t n_items, const int32_t* long_rows, int64_t n_long,
                                        float* partials, int64_t n_partials,
                                        float* neigh_scratch,
                                        const float* W, int64_t ld_w, const float* bias, int32_t H, uint32_t lin_flags,
                                        float* out, int64_t ld_out, void* stream) {
    if (!neigh_scratch || !aligned16(neigh_scratch)) return WGNN_ERR_WORKSPACE;
    if (agg_flags & WGNN_FLAG_RELU) return WGNN_ERR_BAD_ARG;           // the activation belongs to the dense half here
    int rc = wgnn_agg_fwd(rowptr, col, val, alpha, alpha_mode, self_idx, h_src, ld_src, h_self, ld_self, row_ids, inv_deg,
                          nullptr, neigh_scratch, D, nullptr, n_out, D, WGNN_F32, WGNN_F32, agg_flags, items, n_items, long_rows,
                          n_long, partials, n_partials, stream);
    if (rc) return rc;
    return wgnn_linear_fwd(neigh_scratch, D, W, ld_w, bias, out, ld_out, n_out, H, D, lin_flags, stream);
}

// wgnn_train.hip - fused glue kernels of the training step (reference train.py:80-87) for CDNA4 (gfx950).
//
// A full-batch cfg3 training step is 6 aggregation passes + 7 GEMMs; what sits between them in the autograd graph -
// ReLU mask, per-row factors, bias gradient, self-loop gradient, the alpha row dots, the loss - was ~25 framework
// elementwise / reduce launches, each a full pass over a [rows, D] tensor (profiles/r04_train_kernel_stats.csv).
// Two kernels replace them:
//
//   agg_bwd_prepare   everything the backward of ONE aggregation pass needs from the upstream gradient, in one read:
//                       g        = gout * (out > 0)                                   (ReLU of NodeUpdate, gnn.py:21-22)
//                       g_scaled = f[r] * g ,  f[r] = inv_deg[r] (* alpha[r] for cell->gene edges)   -> source table of K2t
//                       dh_self  = alpha[self] * inv_deg[r] * g                                      -> gradient of the self rows
//                       dalpha_row[r] = inv_deg[r] * <g[r], neigh_sum[r]>   (cell->gene edges: dalpha[gene r])
//                       dself_row[r]  = inv_deg[r] * <g[r], h_self[r]>      (-> dalpha[self], summed by the caller)
//                       dbias[c] = sum_r g[r, c]                            (two-stage, fixed order)
//   ce_sum_fwd_bwd    CrossEntropyLoss(reduction='sum') (train.py:36): loss and dloss/dlogits = softmax - onehot, one
//                     read of the logits; block partials folded in fixed order.
// All outputs are deterministic (no atomics).  HBM-bound elementwise work: float4 per lane, one wave per row.
#include "wgnn_common.h"

namespace {
using namespace wgnn;

constexpr int kPrepWaves = 4;                  // waves per block
constexpr int kPrepMaxBlocks = 2048;           // dbias partial rows (workspace: kPrepMaxBlocks * D floats)

struct PrepArgs {
    const float* gout; long ld_gout;
    const float* out; long ld_out;               // saved forward output (ReLU mask) or nullptr
    const float* inv_deg; const float* alpha; int mode; int self_idx;
    float* g_scaled;                             // [R, D] contiguous, or nullptr
    const float* h_self; long ld_self; float* dh_self; long ld_dh;
    const float* neigh_sum;                      // [R, D] contiguous, or nullptr
    float* dalpha_row; float* dself_row; float* dbias_part;   // dbias_part: [gridDim.x, D]
    long n_rows; int D;
};

// one wave per row (grid-stride over rows); lane l owns columns 4l .. 4l+3 of every 256-column slab
__global__ void __launch_bounds__(kPrepWaves * 64) agg_bwd_prepare(const PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [kPrepWaves][D] (dbias only)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wave_id = (long)blockIdx.x * kPrepWaves + wave, n_waves = (long)gridDim.x * kPrepWaves;
    const float a_self = (a.dh_self && a.mode != WGNN_NO_ALPHA) ? a.alpha[a.self_idx] : 1.0f;
    const int n_slab = (a.D + 255) / 256;
    float4 colsum[4];                                                   // D <= 1024
#pragma unroll
    for (int s = 0; s < 4; ++s) colsum[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long r = wave_id; r < a.n_rows; r += n_waves) {
        const float invd = a.inv_deg ? a.inv_deg[r] : 1.0f;
        const float f_src = invd * (a.mode == WGNN_DST_IS_GENE ? a.alpha[r] : 1.0f);
        const float f_self = a_self * invd;
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s >= n_slab) break;
            const int c0 = s * 256 + lane * 4;
            if (c0 >= a.D) continue;
            float4 g = ld4(a.gout + (size_t)r * a.ld_gout + c0);
            if (a.out) {
                const float4 o = ld4(a.out + (size_t)r * a.ld_out + c0);
                g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
            }
            if (a.g_scaled) st4(a.g_scaled + (size_t)r * a.D + c0, make_float4(f_src * g.x, f_src * g.y, f_src * g.z, f_src * g.w));
            if (a.dh_self) st4(a.dh_self + (size_t)r * a.ld_dh + c0, make_float4(f_self * g.x, f_self * g.y, f_self * g.z, f_self * g.w));
            if (a.dalpha_row) d1 += dot4(g, ld4(a.neigh_sum + (size_t)r * a.D + c0));
            if (a.dself_row) d2 += dot4(g, ld4(a.h_self + (size_t)r * a.ld_self + c0));
            colsum[s].x += g.x; colsum[s].y += g.y; colsum[s].z += g.z; colsum[s].w += g.w;
        }
        if (a.dalpha_row) { d1 = group_sum<64>(d1); if (lane == 0) a.dalpha_row[r] = invd * d1; }
        if (a.dself_row) { d2 = group_sum<64>(d2); if (lane == 0) a.dself_row[r] = invd * d2; }
    }
    if (!a.dbias_part) return;
    // column sums: the block's waves fold through LDS in wave order, one partial row per block
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c0 = s * 256 + lane * 4;
        if (s < n_slab && c0 < a.D) *reinterpret_cast<float4*>(red + (size_t)wave * a.D + c0) = colsum[s];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.D; c += kPrepWaves * 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kPrepWaves; ++w) t += red[(size_t)w * a.D + c];
        a.dbias_part[(size_t)blockIdx.x * a.D + c] = t;
    }
}

// out[c] = sum_b part[b, c], in a fixed order (deterministic): a block owns 16 columns, thread (j, c) sums the rows
// b = j (mod 16) with four independent accumulators (the loads of one thread do not wait for each other), the 16
// per-column partials are folded through LDS in j order.  (A single thread per column walking 2048 rows serially took
// 0.47 ms - more than everything the fused kernel saves.)
__global__ void __launch_bounds__(256) fold_rows(const float* __restrict__ part, float* __restrict__ out, int n_part, int D) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (c < D) {
        int b = j;
        for (; b + 48 < n_part; b += 64) {
            t0 += part[(size_t)b * D + c]; t1 += part[(size_t)(b + 16) * D + c];
            t2 += part[(size_t)(b + 32) * D + c]; t3 += part[(size_t)(b + 48) * D + c];
        }
        for (; b < n_part; b += 16) t0 += part[(size_t)b * D + c];
    }
    red[j][cl] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (j == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[c] = t;
    }
}

// ---- CrossEntropyLoss(reduction='sum') + its gradient w.r.t. the logits --------------------------------------------
// one thread per row (n_classes is small: 16 at the BASELINE configs); block partial of the loss -> part[blockIdx.x]
__global__ void __launch_bounds__(256) ce_sum_rows(const float* __restrict__ logits, long ld, const long long* __restrict__ labels,
                                                   long n_rows, int C, float* __restrict__ dlogits, long ld_d,
                                                   float* __restrict__ part) {
    __shared__ float wsum[4];
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (r < n_rows) {
        const float* x = logits + (size_t)r * ld;
        const long long y64 = labels[r];                       // range-checked on the 64-bit value (never truncated first)
        const bool ignored = y64 == -100;                      // torch's default ignore_index: loss 0, gradient row 0
        const bool in_range = y64 >= 0 && y64 < (long long)C;  // any other label outside [0, C) never indexes the row: NaN, loudly
        const int y = in_range ? (int)y64 : 0;
        float m = x[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(x[c] - m);       // expf / logf, not the fast intrinsics: the kernel is HBM-bound and the
        const float lse = m + logf(s);                         // __logf error (~4e-7 per row) is systematic over 1e5..1e6 summed rows
        loss = ignored ? 0.f : (in_range ? lse - x[y] : __builtin_nanf(""));     // -log softmax(x)[y]
        if (dlogits) {
            float* d = dlogits + (size_t)r * ld_d;
            const float inv = 1.0f / s;
            for (int c = 0; c < C; ++c)
                d[c] = ignored ? 0.f : (in_range ? expf(x[c] - m) * inv - (c == y ? 1.0f : 0.0f) : __builtin_nanf(""));
        }
    }
    loss = group_sum<64>(loss);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// loss = sum of the block partials, folded by ONE wave in a fixed order (lane-strided, then a shuffle tree)
__global__ void __launch_bounds__(64) fold_scalar(const float* __restrict__ part, long n, float* __restrict__ out) {
    float t = 0.f;
    for (long i = threadIdx.x; i < n; i += 64) t += part[i];
    t = group_sum<64>(t);
    if (threadIdx.x == 0) out[0] = t;
}

inline int prep_blocks(long n_rows) {
    const long want = (n_rows + kPrepWaves - 1) / kPrepWaves;
    return (int)(want < kPrepMaxBlocks ? (want > 0 ? want : 1) : kPrepMaxBlocks);
}

}  // namespace

extern "C" int wgnn_agg_bwd_prepare_workspace(int64_t n_rows, int32_t D, int64_t* floats) {
    if (!floats || n_rows < 0 || D <= 0) return WGNN_ERR_BAD_ARG;
    *floats = (int64_t)prep_blocks(n_rows) * D;
    return WGNN_OK;
}

extern "C" int wgnn_agg_bwd_prepare(const float* gout, int64_t ld_gout, const float* out, int64_t ld_out,
                                    const float* inv_deg, const float* alpha, int alpha_mode, int32_t self_idx,
                                    float* g_scaled, const float* h_self, int64_t ld_self, float* dh_self, int64_t ld_dh,
                                    const float* neigh_sum, float* dalpha_row, float* dself_row, float* dbias,
                                    int64_t n_rows, int32_t D, float* workspace, int64_t workspace_floats, void* stream) {
    if (!gout || n_rows < 0) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (D <= 0 || D % 4 || ld_gout % 4 || (out && ld_out % 4) || (h_self && ld_self % 4) || (dh_self && ld_dh % 4)) return WGNN_ERR_ALIGNMENT;
    if (D > 1024) return WGNN_ERR_UNSUPPORTED;
    if ((dalpha_row && !neigh_sum) || (dself_row && !h_self)) return WGNN_ERR_BAD_ARG;
    if (!aligned16(gout) || (out && !aligned16(out)) || (g_scaled && !aligned16(g_scaled)) || (h_self && !aligned16(h_self)) ||
        (dh_self && !aligned16(dh_self)) || (neigh_sum && !aligned16(neigh_sum)))
        return WGNN_ERR_ALIGNMENT;
    const int nb = prep_blocks(n_rows);
    if (dbias && (!workspace || workspace_floats < (int64_t)nb * D)) return WGNN_ERR_WORKSPACE;
    if (n_rows == 0) {
        if (dbias && hipMemsetAsync(dbias, 0, sizeof(float) * D, static_cast<hipStream_t>(stream)) != hipSuccess) return WGNN_ERR_LAUNCH;
        return WGNN_OK;
    }
    PrepArgs a{gout, ld_gout, out, ld_out, inv_deg, alpha, alpha_mode, self_idx, g_scaled, h_self, ld_self, dh_self, ld_dh,
               neigh_sum, dalpha_row, dself_row, dbias ? workspace : nullptr, n_rows, D};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = dbias ? sizeof(float) * kPrepWaves * D : 0;
    hipLaunchKernelGGL(agg_bwd_prepare, dim3(nb), dim3(kPrepWaves * 64), lds, st, a);
    if (dbias) hipLaunchKernelGGL(fold_rows, dim3((D + 15) / 16), dim3(256), 0, st, workspace, dbias, nb, D);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

extern "C" int wgnn_ce_sum_workspace(int64_t n_rows, int64_t* floats) {
    if (!floats || n_rows < 0) return WGNN_ERR_BAD_ARG;
    *floats = (n_rows + 255) / 256 + 1;
    return WGNN_OK;
}

extern "C" int wgnn_ce_sum_fwd_bwd(const float* logits, int64_t ld_logits, const int64_t* labels, int64_t n_rows, int32_t n_classes,
                                   float* loss_sum, float* dlogits, int64_t ld_d, float* workspace, int64_t workspace_floats,
                                   void* stream) {
    if (!logits || !labels || !loss_sum || n_rows < 0 || n_classes <= 0) return WGNN_ERR_BAD_ARG;
    if (ld_logits < n_classes || (dlogits && ld_d < n_classes)) return WGNN_ERR_BAD_ARG;
    const long nb = (n_rows + 255) / 256;
    if (!workspace || workspace_floats < nb + 1) return WGNN_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_rows == 0) return hipMemsetAsync(loss_sum, 0, sizeof(float), st) == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
    hipLaunchKernelGGL(ce_sum_rows, dim3((unsigned)nb), dim3(256), 0, st, logits, (long)ld_logits,
                       reinterpret_cast<const long long*>(labels), (long)n_rows, (int)n_classes, dlogits, (long)ld_d, workspace);
    hipLaunchKernelGGL(fold_scalar, dim3(1), dim3(64), 0, st, workspace, nb, loss_sum);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

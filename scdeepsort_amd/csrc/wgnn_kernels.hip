// wgnn_kernels.hip - CDNA4 (gfx950) kernels + C ABI for scDeepSort's weighted-mean aggregation.
//
// Replaces, for the hot path of the reference:
//   GNN.message_func            models/gnn.py:47-56   (per-edge h[src]*alpha[k]*w)
//   fn.mean('m','neigh')        models/gnn.py:65      (sum over in-edges / in-degree, DGL 0.4.3)
//   NodeUpdate bias + ReLU      models/gnn.py:20-22   (fused epilogue; the GEMM is applied project-first)
//   autograd backward of those  train.py:84
//   normalize_weight            utils/preprocess_internal.py:15-23
//
// Design: destination-major CSR; one 64-lane wavefront owns one
// row chunk ("item").  A feature row of D floats is covered by LPR lanes x float4
// (16-byte, fully coalesced 1 KiB row reads at D=256); when D < 256 the 64/LPR
// lane groups take different non-zeros of the same row and are folded with
// xor-shuffles at the end.  Column index / weight of 64 non-zeros are fetched
// with one coalesced load each and broadcast from registers (v_readlane /
// ds_bpermute), alpha[k(e)] is folded into the weight once per edge, 1/(deg+1),
// the implicit self-loop, bias and ReLU are fused into the epilogue.  Rows longer
// than the plan's chunk size are split; their partial sums go to a caller-owned
// scratch and are folded in a fixed order by a finalize kernel (deterministic,
// no atomics).

#include <limits.h>
#include "wgnn_common.h"

namespace {
using namespace wgnn;

// ---------------------------------------------------------------------------------------------
// main kernel: one wave per item = (row slot, nnz range [begin,end), partial slot or -1)
// ---------------------------------------------------------------------------------------------
template <int LPR, int NV, typename TIn, typename TOut, int EPI>
__global__ void __launch_bounds__(kBlock) agg_main(const KArgs a) {
    constexpr int G = 64 / LPR;                   // non-zeros of one row processed side by side
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (item >= a.n_items) return;                // whole wave exits together
    const int4 it = a.items[item];
    const int slot = it.x, begin = it.y, end = it.z, pslot = it.w;
    if (slot < 0) return;                         // an unused item of a static-shape (device-built) plan: wave-uniform exit
    const int sub = lane / LPR, l = lane % LPR;
    const TIn* __restrict__ src = reinterpret_cast<const TIn*>(a.src);

    float4 acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    // coalesced fetch of 64 (col, weight) pairs; alpha / inv_deg folded into the weight once per edge
    auto fetch = [&](int base, int& c, float& w) {
        const int n = min(64, end - base);
        const int idx = base + min(lane, n - 1);
        c = a.col[idx];
        float ww = a.val[idx];
        if (a.cs1) ww *= a.cs1[c];
        if (a.cs2) ww *= a.cs2[c];
        w = lane < n ? ww : 0.f;
    };

    int c_cur = 0; float w_cur = 0.f;
    if (begin < end) fetch(begin, c_cur, w_cur);
    for (int base = begin; base < end; base += 64) {
        int c_nxt = 0; float w_nxt = 0.f;
        if (base + 64 < end) fetch(base + 64, c_nxt, w_nxt);          // software prefetch of the next tile
        const int n = min(64, end - base);
        const int steps = (n + G - 1) / G;
        // U gathers in flight per lane before the first FMA (the loop is otherwise bound by one L2 round trip per
        // non-zero); steps past the end re-read the last row with weight 0.
        constexpr int U = NV >= 4 ? 2 : (NV >= 2 ? 4 : 8);
        for (int j0 = 0; j0 < steps; j0 += U) {
            float4 x[U][NV];
            float wu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(j0 + u, steps - 1);
                int c; float w;
                if constexpr (G == 1) {
                    c = __builtin_amdgcn_readlane(c_cur, j);
                    w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_cur), j));
                } else {
                    c = __shfl(c_cur, j * G + sub, 64);
                    w = __shfl(w_cur, j * G + sub, 64);
                }
                wu[u] = j0 + u < steps ? w : 0.f;
                const TIn* p = src + (size_t)c * a.ld_src;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int c0 = (k * LPR + l) * 4;
                    x[u][k] = c0 < a.D ? ld4(p + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < NV; ++k) fma4(acc[k], wu[u], x[u][k]);
        }
        c_cur = c_nxt; w_cur = w_nxt;
    }
    // fold the G lane groups (each holds a partial sum over its share of the non-zeros)
    if constexpr (G > 1) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                acc[k].x += __shfl_xor(acc[k].x, off, 64); acc[k].y += __shfl_xor(acc[k].y, off, 64);
                acc[k].z += __shfl_xor(acc[k].z, off, 64); acc[k].w += __shfl_xor(acc[k].w, off, 64);
            }
        }
    }
    const bool writer = (sub == 0);
    if (pslot >= 0) {
        float* pp = a.partials + (size_t)pslot * a.D;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c0 = (k * LPR + l) * 4;
            if (writer && c0 < a.D) st4(pp + c0, acc[k]);
        }
    } else {
        epilogue<LPR, NV, TIn, TOut, EPI>(a, acc, slot, l, writer);
    }
}

// finalize: one wave per long row; folds its partial sums in slot order, then the same epilogue
template <int LPR, int NV, typename TIn, typename TOut, int EPI>
__global__ void __launch_bounds__(kBlock) agg_finalize(const KArgs a) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (idx >= a.n_long) return;
    const int4 lr = a.long_rows[idx];
    if (lr.x < 0) return;                         // unused long-row record of a static-shape plan
    const int sub = lane / LPR, l = lane % LPR;
    float4 acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub == 0) {
        // groups of four partial rows: all four requested before any is added (the loop is a chain of L2 round trips
        // otherwise - one wave per row, 3 .. 48 parts); rows past the end re-read the last one and are dropped by a select.
        // The ADDITION order stays p = 0, 1, 2, ...
        const float* base = a.partials + (size_t)lr.y * a.D;
        const int n = lr.z;
        for (int p = 0; p < n; p += 4) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c0 = (k * LPR + l) * 4;
                if (c0 < a.D) {
                    float4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = ld4(base + (size_t)min(p + i, n - 1) * a.D + c0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool on = p + i < n;                       // wave-uniform
                        acc[k].x += on ? v[i].x : 0.f; acc[k].y += on ? v[i].y : 0.f;
                        acc[k].z += on ? v[i].z : 0.f; acc[k].w += on ? v[i].w : 0.f;
                    }
                }
            }
        }
    }
    epilogue<LPR, NV, TIn, TOut, EPI>(a, acc, lr.x, l, sub == 0);
}

// K4: normalize_weight - one wave per row
template <typename TPtr>
__global__ void __launch_bounds__(kBlock) normalize_rows(const TPtr* __restrict__ rowptr, const float* __restrict__ vin,
                                                         float* __restrict__ vout, float* __restrict__ inv_deg, long n_rows) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const TPtr b = rowptr[r], e = rowptr[r + 1];
    // Four loads in flight per lane and trip (round 6): one 256-byte wave load per round trip made the kernel latency-bound
    // (640 MB in 0.57 ms = 1.1 TB/s at cfg3).  The additions keep their order (a lane still sums its elements j, j + 64, ...
    // one after the other), so the result is bit-identical to the one-load-per-trip loop.
    float s = 0.f;
    for (TPtr j = b + lane; j < e; j += 256) {
        const float a0 = vin[j];
        const float a1 = j + 64 < e ? vin[j + 64] : 0.f, a2 = j + 128 < e ? vin[j + 128] : 0.f, a3 = j + 192 < e ? vin[j + 192] : 0.f;
        s += a0;
        if (j + 64 < e) s += a1;
        if (j + 128 < e) s += a2;
        if (j + 192 < e) s += a3;
    }
    s = group_sum<64>(s);
    const float deg = (float)(e - b);
    for (TPtr j = b + lane; j < e; j += 256) {                              // (deg*w)/sum, preprocess_internal.py:23
        const float a0 = vin[j];
        const float a1 = j + 64 < e ? vin[j + 64] : 0.f, a2 = j + 128 < e ? vin[j + 128] : 0.f, a3 = j + 192 < e ? vin[j + 192] : 0.f;
        vout[j] = deg * a0 / s;
        if (j + 64 < e) vout[j + 64] = deg * a1 / s;
        if (j + 128 < e) vout[j + 128] = deg * a2 / s;
        if (j + 192 < e) vout[j + 192] = deg * a3 / s;
    }
    if (inv_deg && lane == 0) inv_deg[r] = 1.0f / (deg + 1.0f);
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
template <int LPR, int NV, typename TIn, typename TOut, int EPI>
int launch(const KArgs& a, hipStream_t st) {
    if (a.n_items > 0) {
        const long nb = (a.n_items + kWavesPerBlock - 1) / kWavesPerBlock;
        hipLaunchKernelGGL((agg_main<LPR, NV, TIn, TOut, EPI>), dim3((unsigned)nb), dim3(kBlock), 0, st, a);
    }
    if (a.n_long > 0) {
        const long nb = (a.n_long + kWavesPerBlock - 1) / kWavesPerBlock;
        hipLaunchKernelGGL((agg_finalize<LPR, NV, TIn, TOut, EPI>), dim3((unsigned)nb), dim3(kBlock), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

template <typename TIn, typename TOut, int EPI>
int dispatch_width(const KArgs& a, hipStream_t st) {
    const int q = a.D / 4;
    if (q <= 8)   return launch<8, 1, TIn, TOut, EPI>(a, st);
    if (q <= 16)  return launch<16, 1, TIn, TOut, EPI>(a, st);
    if (q <= 32)  return launch<32, 1, TIn, TOut, EPI>(a, st);
    if (q <= 64)  return launch<64, 1, TIn, TOut, EPI>(a, st);
    if (q <= 128) return launch<64, 2, TIn, TOut, EPI>(a, st);
    if (q <= 256) return launch<64, 4, TIn, TOut, EPI>(a, st);
    return WGNN_ERR_UNSUPPORTED;
}

}  // namespace

namespace wgnn {
// used by wgnn_tiled.hip: fold the partial sums of column-split tiles (same finalize kernel, same epilogue)
int launch_finalize_f32(const KArgs& a, int epi, hipStream_t st) {
    KArgs b = a;
    b.n_items = 0;
    if (epi == EPI_FWD) return dispatch_width<float, float, EPI_FWD>(b, st);
    if (epi == EPI_BWD_SRC) return dispatch_width<float, float, EPI_BWD_SRC>(b, st);
    return dispatch_width<float, float, EPI_BWD_ALPHA>(b, st);
}
}  // namespace wgnn

template <typename TPtr>
static int plan_build(const TPtr* rowptr, const int32_t* row_ids, int64_t n_rows, int32_t chunk,
                      int32_t* items, int32_t* long_rows, int64_t* n_items, int64_t* n_long, int64_t* n_partials) {
    if (!rowptr || n_rows < 0 || chunk <= 0 || !n_items || !n_long || !n_partials) return WGNN_ERR_BAD_ARG;
    int64_t ni = 0, nl = 0, np = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        const int64_t r = row_ids ? row_ids[i] : i;
        const int64_t b64 = rowptr[r], e64 = rowptr[r + 1];
        if (e64 < b64 || b64 < 0) return WGNN_ERR_PLAN;
        if (e64 > INT32_MAX) return WGNN_ERR_UNSUPPORTED;         // 32-bit nnz offsets in items / kernels: shard the cell axis
        const int32_t b = (int32_t)b64, e = (int32_t)e64;
        const int32_t nnz = e - b;
        if (nnz <= chunk) {
            if (items) { int32_t* it = items + 4 * ni; it[0] = (int32_t)i; it[1] = b; it[2] = e; it[3] = -1; }
            ++ni;
        } else {
            const int32_t nc = (nnz + chunk - 1) / chunk;
            if (long_rows) { int32_t* lr = long_rows + 4 * nl; lr[0] = (int32_t)i; lr[1] = (int32_t)np; lr[2] = nc; lr[3] = 0; }
            for (int32_t c = 0; c < nc; ++c) {
                if (items) {
                    int32_t* it = items + 4 * ni;
                    const int64_t ce = (int64_t)b + (int64_t)(c + 1) * chunk;
                    it[0] = (int32_t)i; it[1] = b + c * chunk; it[2] = ce < e ? (int32_t)ce : e;
                    it[3] = (int32_t)(np + c);
                }
                ++ni;
            }
            np += nc; ++nl;
        }
    }
    *n_items = ni; *n_long = nl; *n_partials = np;
    return WGNN_OK;
}

template <typename TPtr>
static int normalize_rows_launch(const TPtr* rowptr, const float* val_in, float* val_out, float* inv_deg, int64_t n_rows, void* stream) {
    if (!rowptr || !val_in || !val_out || n_rows < 0) return WGNN_ERR_BAD_ARG;
    if (n_rows == 0) return WGNN_OK;
    const long nb = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(normalize_rows<TPtr>, dim3((unsigned)nb), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       rowptr, val_in, val_out, inv_deg, (long)n_rows);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

extern "C" {

int wgnn_version(void) { return WGNN_VERSION; }

int wgnn_agg_workspace_bytes(int64_t n_partials, int64_t n_src, int32_t D, int alpha_mode, int tiled,
                             int64_t* partials_bytes, int64_t* src_scratch_bytes) {
    if (n_partials < 0 || n_src < 0 || D <= 0 || !partials_bytes || !src_scratch_bytes) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (D % 4) return WGNN_ERR_ALIGNMENT;
    *partials_bytes = n_partials * (int64_t)D * 4;
    *src_scratch_bytes = (tiled && alpha_mode == WGNN_SRC_IS_GENE) ? n_src * (int64_t)D * 4 : 0;
    return WGNN_OK;
}

const char* wgnn_last_error_string(int code) {
    switch (code) {
        case WGNN_OK: return "ok";
        case WGNN_ERR_BAD_ARG: return "bad argument (null pointer, negative size or bad enum)";
        case WGNN_ERR_ALIGNMENT: return "D / leading dimension not a multiple of 4 or pointer not 16-byte aligned";
        case WGNN_ERR_UNSUPPORTED: return "unsupported dtype, feature width (D <= 1024 required) or nnz >= 2^31 (shard the cell axis)";
        case WGNN_ERR_WORKSPACE: return "workspace (partials) too small or missing";
        case WGNN_ERR_LAUNCH: return "HIP launch failed";
        case WGNN_ERR_PLAN: return "plan malformed";
        default: return "unknown error";
    }
}

int wgnn_plan_build_host(const int32_t* rowptr, const int32_t* row_ids, int64_t n_rows, int32_t chunk,
                         int32_t* items, int32_t* long_rows, int64_t* n_items, int64_t* n_long, int64_t* n_partials) {
    return plan_build(rowptr, row_ids, n_rows, chunk, items, long_rows, n_items, n_long, n_partials);
}

int wgnn_plan_build_host_i64(const int64_t* rowptr, const int32_t* row_ids, int64_t n_rows, int32_t chunk,
                             int32_t* items, int32_t* long_rows, int64_t* n_items, int64_t* n_long, int64_t* n_partials) {
    return plan_build(rowptr, row_ids, n_rows, chunk, items, long_rows, n_items, n_long, n_partials);
}

static int check_common(int32_t D, int64_t n, const void* items, int64_t n_items, const void* long_rows,
                        int64_t n_long, const float* partials, int64_t n_partials) {
    if (D <= 0 || n < 0 || n_items < 0 || n_long < 0) return WGNN_ERR_BAD_ARG;
    if (D % 4) return WGNN_ERR_ALIGNMENT;
    if (D > 1024) return WGNN_ERR_UNSUPPORTED;
    if (n_items > 0 && !items) return WGNN_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !partials || n_partials <= 0)) return WGNN_ERR_WORKSPACE;
    return WGNN_OK;
}

int wgnn_agg_fwd(const void* rowptr, const int32_t* col, const float* val,
                 const float* alpha, int alpha_mode, int32_t self_idx,
                 const void* h_src, int64_t ld_src, const void* h_self, int64_t ld_self,
                 const int32_t* row_ids, const float* inv_deg, const float* bias,
                 void* out, int64_t ld_out, float* neigh_sum, int64_t n_out, int32_t D, int dtype_in, int dtype_out, uint32_t flags,
                 const int32_t* items, int64_t n_items, const int32_t* long_rows, int64_t n_long,
                 float* partials, int64_t n_partials, void* stream) {
    int rc = check_common(D, n_out, items, n_items, long_rows, n_long, partials, n_partials);
    if (rc) return rc;
    if (!rowptr || !col || !val || !h_src || !out) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (ld_src % 4 || ld_out % 4 || (h_self && ld_self % 4)) return WGNN_ERR_ALIGNMENT;
    const bool in16 = dtype_in == WGNN_F16, out16 = dtype_out == WGNN_F16;
    if ((dtype_in != WGNN_F32 && !in16) || (dtype_out != WGNN_F32 && !out16)) return WGNN_ERR_UNSUPPORTED;
    if (!(in16 ? aligned8(h_src) : aligned16(h_src)) || !(out16 ? aligned8(out) : aligned16(out)) ||
        (h_self && !(in16 ? aligned8(h_self) : aligned16(h_self))) || (bias && !aligned16(bias)))
        return WGNN_ERR_ALIGNMENT;
    if (n_out == 0) return WGNN_OK;
    KArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val;
    a.cs1 = (alpha_mode == WGNN_SRC_IS_GENE) ? alpha : nullptr; a.cs2 = nullptr;
    a.src = h_src; a.ld_src = ld_src; a.alpha = alpha; a.mode = alpha_mode; a.self_idx = self_idx;
    a.self = h_self; a.ld_self = ld_self; a.row_ids = row_ids; a.inv_deg = inv_deg; a.bias = bias;
    a.out = out; a.ld_out = ld_out; a.D = D; a.flags = flags; a.aux1 = neigh_sum;
    a.items = reinterpret_cast<const int4*>(items); a.n_items = n_items;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    if (neigh_sum && !aligned16(neigh_sum)) return WGNN_ERR_ALIGNMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!in16 && !out16) return dispatch_width<float, float, EPI_FWD>(a, st);
    if (in16 && !out16)  return dispatch_width<__half, float, EPI_FWD>(a, st);
    if (in16 && out16)   return dispatch_width<__half, __half, EPI_FWD>(a, st);
    return WGNN_ERR_UNSUPPORTED;
}

int wgnn_agg_bwd_src(const int32_t* t_rowptr, const int32_t* t_col, const float* t_val,
                     const float* alpha, int alpha_mode, const float* inv_deg_dst,
                     const float* g, int64_t ld_g, const float* h_src, int64_t ld_src,
                     float* dh_src, int64_t ld_dh, float* dalpha, int accumulate,
                     int64_t n_src, int32_t D,
                     const int32_t* items, int64_t n_items, const int32_t* long_rows, int64_t n_long,
                     float* partials, int64_t n_partials, void* stream) {
    int rc = check_common(D, n_src, items, n_items, long_rows, n_long, partials, n_partials);
    if (rc) return rc;
    if (!t_rowptr || !t_col || !t_val || !g || !dh_src) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (ld_g % 4 || ld_dh % 4 || (h_src && ld_src % 4)) return WGNN_ERR_ALIGNMENT;
    if (!aligned16(g) || !aligned16(dh_src) || (h_src && !aligned16(h_src))) return WGNN_ERR_ALIGNMENT;
    if (n_src == 0) return WGNN_OK;
    KArgs a{};
    a.rowptr = t_rowptr; a.col = t_col; a.val = t_val;
    a.cs1 = inv_deg_dst; a.cs2 = (alpha_mode == WGNN_DST_IS_GENE) ? alpha : nullptr;
    a.src = g; a.ld_src = ld_g; a.alpha = alpha; a.mode = alpha_mode;
    a.self = h_src; a.ld_self = ld_src; a.out = dh_src; a.ld_out = ld_dh; a.aux1 = dalpha;
    a.D = D; a.flags = WGNN_FLAG_NO_MEAN; a.accumulate = accumulate;
    a.items = reinterpret_cast<const int4*>(items); a.n_items = n_items;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    return dispatch_width<float, float, EPI_BWD_SRC>(a, static_cast<hipStream_t>(stream));
}

int wgnn_agg_bwd_alpha(const int32_t* rowptr, const int32_t* col, const float* val,
                       const float* inv_deg, const int32_t* row_ids,
                       const float* g, int64_t ld_g, const float* h_src, int64_t ld_src,
                       const float* h_self, int64_t ld_self, float* dalpha_row, float* dself_row,
                       int64_t n_out, int32_t D, uint32_t flags,
                       const int32_t* items, int64_t n_items, const int32_t* long_rows, int64_t n_long,
                       float* partials, int64_t n_partials, void* stream) {
    int rc = check_common(D, n_out, items, n_items, long_rows, n_long, partials, n_partials);
    if (rc) return rc;
    if (!rowptr || !col || !val || !g || !h_src) return WGNN_ERR_BAD_ARG;
    if (ld_g % 4 || ld_src % 4 || (h_self && ld_self % 4)) return WGNN_ERR_ALIGNMENT;
    if (!aligned16(g) || !aligned16(h_src) || (h_self && !aligned16(h_self))) return WGNN_ERR_ALIGNMENT;
    if (n_out == 0) return WGNN_OK;
    KArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val;
    a.src = h_src; a.ld_src = ld_src; a.mode = WGNN_NO_ALPHA;
    a.self = h_self; a.ld_self = ld_self; a.row_ids = row_ids; a.inv_deg = inv_deg;
    a.g = g; a.ld_g = ld_g; a.aux1 = dalpha_row; a.aux2 = dself_row;
    a.D = D; a.flags = flags & WGNN_FLAG_SELF_COMPACT;
    a.items = reinterpret_cast<const int4*>(items); a.n_items = n_items;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    return dispatch_width<float, float, EPI_BWD_ALPHA>(a, static_cast<hipStream_t>(stream));
}

int wgnn_normalize_rows(const int32_t* rowptr, const float* val_in, float* val_out, float* inv_deg,
                        int64_t n_rows, void* stream) {
    return normalize_rows_launch<int>(rowptr, val_in, val_out, inv_deg, n_rows, stream);
}

int wgnn_normalize_rows_i64(const int64_t* rowptr, const float* val_in, float* val_out, float* inv_deg,
                            int64_t n_rows, void* stream) {
    return normalize_rows_launch<long long>(reinterpret_cast<const long long*>(rowptr), val_in, val_out, inv_deg, n_rows, stream);
}

}  // extern "C"

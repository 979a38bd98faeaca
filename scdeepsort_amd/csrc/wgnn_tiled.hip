// wgnn_tiled.hip - LDS-streamed aggregation kernel for CDNA4 (gfx950).
//
// Same arithmetic as agg_main (reference models/gnn.py:47-56,65 + fused bias/ReLU, gnn.py:20-22),
// different data movement.  The row-wave kernel re-gathers every source row from L2/MALL once per
// non-zero (nnz*D*4 B = 82 GB per pass at BASELINE cfg3) and is bound by the ~64 B/clk/CU vector
// memory path.  Here one 1024-thread workgroup (16 waves, one per CU) owns a TILE of up to 256
// destination rows - 16 per wave, accumulators resident in VGPRs (16 x float4 per lane) - and the
// SOURCE table is streamed through LDS in blocks of 64 rows (64 KiB at D=256) with the async
// global->LDS DMA (global_load_lds_dwordx4), double-buffered.  Every staged source row is consumed
// by all rows of the tile that reference it, so the per-non-zero gather becomes a conflict-free
// ds_read_b128 (256 B/clk/CU) and global traffic drops to (rows/256) x |source table|.
//
// Per row the wave keeps a 64-entry window of (col, alpha-folded weight) in two VGPRs; the entries
// that fall into the current source block are found with one compare + ballot (columns are sorted)
// and broadcast with v_readlane.  Tiles carry a column range so that hub rows (genes expressed in
// ~every cell) are split across workgroups; their partial sums are folded by agg_finalize in a
// fixed order (deterministic, no atomics).
#include "wgnn_common.h"
#include <limits.h>

namespace {
using namespace wgnn;

constexpr int kTW = 16;                       // waves per tile workgroup
constexpr int kRPW = 16;                      // destination rows per wave
constexpr int kKB = 64;                       // source rows per LDS block
constexpr int kTileRows = kTW * kRPW;         // 256
constexpr int kRefill = 32;                   // eager window reload once this many entries are consumed
// ablation switches (timing experiments only; results are wrong when set)
constexpr unsigned kDbgNoFill = 1u << 16, kDbgNoCompute = 1u << 17, kDbgNoRefill = 1u << 18;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename TOut, int EPI>
__global__ void __launch_bounds__(kTW * 64) agg_tiled(const KArgs a, const int4* __restrict__ tile_items,
                                                      const int2* __restrict__ tile_hdr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // the ONLY LDS object: 2 x kKB x row_bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x;
    const int2 hdr = tile_hdr[tile];
    const int cb = __builtin_amdgcn_readfirstlane(hdr.x), ce = __builtin_amdgcn_readfirstlane(hdr.y);
    const int row_bytes = a.D * (int)sizeof(float);
    const int buf_bytes = kKB * row_bytes;
    const int4* __restrict__ items = tile_items + (size_t)tile * kTileRows + wave * kRPW;
    const bool active = lane * 16 < row_bytes;

    float4 acc[kRPW];
    int ccol[kRPW];
    float cval[kRPW];
    int base[kRPW], rend[kRPW], pos[kRPW];

    auto load_window = [&](int i) {
        const int idx = base[i] + lane;
        const bool ok = idx < rend[i];
        int c = INT_MAX;
        float v = 0.f;
        if (ok) {
            c = a.col[idx];
            v = a.val[idx];
            if (a.cs1) v *= a.cs1[c];
            if (a.cs2) v *= a.cs2[c];
        }
        ccol[i] = c; cval[i] = v;
    };

#pragma unroll
    for (int i = 0; i < kRPW; ++i) {
        const int4 it = items[i];
        base[i] = __builtin_amdgcn_readfirstlane(it.y);
        rend[i] = __builtin_amdgcn_readfirstlane(it.z);
        pos[i] = 0;
        acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        load_window(i);
    }

    auto fill = [&](int b, int buf) {
        const int r0 = cb + b * kKB;
        const int nbytes = min(kKB, ce - r0) * row_bytes;
        const char* g = reinterpret_cast<const char*>(a.src) + (size_t)r0 * row_bytes;
        char* l = smem + buf * buf_bytes;
        for (int p = wave; p * 1024 < nbytes; p += kTW) {
            const int off = p * 1024 + lane * 16;
            if (off < nbytes)
                __builtin_amdgcn_global_load_lds((gptr_t)(g + off), (lptr_t)(l + p * 1024), 16, 0, 0);
        }
    };

    const int nblk = (ce - cb + kKB - 1) / kKB;
    const bool do_fill = !(a.flags & kDbgNoFill), do_comp = !(a.flags & kDbgNoCompute), do_refill = !(a.flags & kDbgNoRefill);
    if (nblk > 0 && do_fill) fill(0, 0);
    for (int b = 0; b < nblk; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my pieces of block b have landed
        __syncthreads();                                      // everyone's have; everyone is done with block b-1
        if (b + 1 < nblk && do_fill) fill(b + 1, (b + 1) & 1);           // DMA of the next block overlaps this block's FMAs
        const int b0 = cb + b * kKB;
        const int b1 = min(ce, b0 + kKB);
        const char* lbuf = smem + (b & 1) * buf_bytes + lane * 16;
        if (!do_comp) continue;
#pragma unroll
        for (int i = 0; i < kRPW; ++i) {
            while (true) {
                const unsigned long long m = __ballot(lane >= pos[i] && ccol[i] < b1);
                const int e = pos[i] + __popcll(m);
                int j = pos[i];
                for (; j + 1 < e; j += 2) {                   // two independent LDS reads in flight
                    const int c0 = __builtin_amdgcn_readlane(ccol[i], j), c1 = __builtin_amdgcn_readlane(ccol[i], j + 1);
                    const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cval[i]), j));
                    const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cval[i]), j + 1));
                    if (active) {
                        const float4 x0 = *reinterpret_cast<const float4*>(lbuf + (c0 - b0) * row_bytes);
                        const float4 x1 = *reinterpret_cast<const float4*>(lbuf + (c1 - b0) * row_bytes);
                        fma4(acc[i], w0, x0);
                        fma4(acc[i], w1, x1);
                    }
                }
                if (j < e) {
                    const int c0 = __builtin_amdgcn_readlane(ccol[i], j);
                    const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cval[i]), j));
                    if (active) fma4(acc[i], w0, *reinterpret_cast<const float4*>(lbuf + (c0 - b0) * row_bytes));
                }
                pos[i] = e;
                if (e < 64 || base[i] + 64 >= rend[i] || !do_refill) break;
                base[i] += 64; pos[i] = 0;                    // window exhausted inside this block: reload, go on
                load_window(i);
            }
            if (do_refill && pos[i] >= kRefill && base[i] + pos[i] < rend[i]) {   // reload early; consumed in a later block
                base[i] += pos[i]; pos[i] = 0;
                load_window(i);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < kRPW; ++i) {
        const int4 it = items[i];
        const int slot = __builtin_amdgcn_readfirstlane(it.x), pslot = __builtin_amdgcn_readfirstlane(it.w);
        if (slot < 0) continue;
        if (pslot >= 0) {
            if (active) st4(a.partials + (size_t)pslot * a.D + lane * 4, acc[i]);
        } else {
            float4 one[1] = {acc[i]};
            epilogue<64, 1, float, TOut, EPI>(a, one, slot, lane, true);
        }
    }
}

template <typename TOut, int EPI>
int launch_tiled(const KArgs& a, const int4* items, const int2* hdr, long n_tiles, hipStream_t st) {
    const int lds = 2 * kKB * a.D * (int)sizeof(float);
    static int configured = 0;                       // per instantiation
    if (configured < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_tiled<TOut, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return WGNN_ERR_LAUNCH;
        configured = lds;
    }
    hipLaunchKernelGGL((agg_tiled<TOut, EPI>), dim3((unsigned)n_tiles), dim3(kTW * 64), lds, st, a, items, hdr);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

}  // namespace

namespace wgnn {
int launch_finalize_fwd_f32(const KArgs& a, hipStream_t st);     // defined in wgnn_kernels.hip
}

extern "C" int wgnn_agg_fwd_tiled(const int32_t* rowptr, const int32_t* col, const float* val,
                                  const float* alpha, int alpha_mode, int32_t self_idx,
                                  const float* h_src, const float* h_self, int64_t ld_self,
                                  const int32_t* row_ids, const float* inv_deg, const float* bias,
                                  float* out, int64_t ld_out, int64_t n_out, int32_t D, uint32_t flags,
                                  const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                                  const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                                  void* stream) {
    if (!rowptr || !col || !val || !h_src || !out || n_out < 0 || n_tiles < 0) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (D <= 0 || D % 4 || ld_out % 4 || (h_self && ld_self % 4)) return WGNN_ERR_ALIGNMENT;
    if (D > 256) return WGNN_ERR_UNSUPPORTED;                     // one float4 per lane; 2 x 64 x D x 4 B of LDS
    if (!aligned16(h_src) || !aligned16(out) || (h_self && !aligned16(h_self)) || (bias && !aligned16(bias)))
        return WGNN_ERR_ALIGNMENT;
    if (n_tiles > 0 && (!tile_items || !tile_hdr)) return WGNN_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !partials || n_partials <= 0)) return WGNN_ERR_WORKSPACE;
    if (n_out == 0 || n_tiles == 0) return WGNN_OK;
    KArgs a{};
    a.rowptr = rowptr; a.col = col; a.val = val;
    a.cs1 = (alpha_mode == WGNN_SRC_IS_GENE) ? alpha : nullptr;
    a.src = h_src; a.ld_src = D; a.alpha = alpha; a.mode = alpha_mode; a.self_idx = self_idx;
    a.self = h_self; a.ld_self = ld_self; a.row_ids = row_ids; a.inv_deg = inv_deg; a.bias = bias;
    a.out = out; a.ld_out = ld_out; a.D = D; a.flags = flags;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = launch_tiled<float, EPI_FWD>(a, reinterpret_cast<const int4*>(tile_items),
                                          reinterpret_cast<const int2*>(tile_hdr), n_tiles, st);
    if (rc) return rc;
    if (n_long > 0) return launch_finalize_fwd_f32(a, st);
    return WGNN_OK;
}

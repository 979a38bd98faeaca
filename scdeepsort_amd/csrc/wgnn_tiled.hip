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
// The edges of a tile are pre-sorted at plan time into (block, wave, destination row) order
// ("entries": {dst_slot<<8 | src_row_in_block, weight}), so that in every block a wave fetches its
// share with ONE coalesced 8-byte load per lane, issued a full block ahead (its latency hides behind
// the previous block's FMAs), finds each row's run with one compare + ballot, and broadcasts
// (source row, weight) with v_readlane.  alpha[k(e)] for gene->cell edges is a per-source-row factor,
// so it is applied to the source table once (scale_rows) instead of per edge - which is also the
// reference's own multiply order, (h*alpha)*w.  Tiles carry a column range so that hub rows (genes
// expressed in ~every cell) are split across workgroups; their partial sums are folded by
// agg_finalize in a fixed order (deterministic, no atomics).
#include "wgnn_common.h"

namespace {
using namespace wgnn;

constexpr int kTW = 16;                       // waves per tile workgroup
constexpr int kRPW = 16;                      // destination rows per wave
constexpr int kKB = 64;                       // source rows per LDS block
constexpr int kTileRows = kTW * kRPW;         // 256
// ablation switches (timing experiments only; results are wrong when set)
constexpr unsigned kDbgNoFill = 1u << 16, kDbgNoCompute = 1u << 17;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(4))) int* cptr_t;      // immutable plan data -> scalar (SMEM) loads

struct TArgs {
    const int2* entries;      // {dst_slot<<8 | src_local, weight bits}, sorted by (tile, block, wave, dst_slot)
    const int* seg_ptr;       // [(n_tiles*nblk_max*16) + 1] entry offsets per (tile, block, wave)
    const int4* tile_items;   // [n_tiles*256] {row_slot|-1, -, -, partial_slot|-1}
    const int2* tile_hdr;     // [n_tiles] {col_begin, col_end}
    int nblk_max;
};

// out[r] = scale[r] * in[r]   (alpha folded into the source table; tiny: |table| bytes)
__global__ void __launch_bounds__(256) scale_rows(const float* __restrict__ in, const float* __restrict__ scale,
                                                  float* __restrict__ out, long n_rows, int D4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * D4) return;
    const float s = scale[i / D4];
    float4 v = reinterpret_cast<const float4*>(in)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    reinterpret_cast<float4*>(out)[i] = v;
}

template <typename TOut, int EPI>
__global__ void __launch_bounds__(kTW * 64) agg_tiled(const KArgs a, const TArgs t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // the ONLY LDS object: 2 x kKB x row_bytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x;
    const int2 hdr = t.tile_hdr[tile];
    const int cb = __builtin_amdgcn_readfirstlane(hdr.x), ce = __builtin_amdgcn_readfirstlane(hdr.y);
    const int row_bytes = a.D * (int)sizeof(float);
    const int buf_bytes = kKB * row_bytes;
    const bool active = lane * 16 < row_bytes;
    const int nblk = (ce - cb + kKB - 1) / kKB;
    cptr_t seg = (cptr_t)(t.seg_ptr + ((size_t)tile * t.nblk_max) * kTW + wave);   // seg[b*16], seg[b*16+1]
    const bool do_fill = !(a.flags & kDbgNoFill), do_comp = !(a.flags & kDbgNoCompute);

    float4 acc[kRPW];
#pragma unroll
    for (int i = 0; i < kRPW; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

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
    // entry chunk: lane j <- entry s+j (sentinel row 255 beyond the segment end)
    auto load_chunk = [&](int s, int e, int2& ent) {
        const int idx = s + lane;
        ent = make_int2(0xFF00, 0);
        if (idx < e) ent = t.entries[idx];
    };
    // consume one chunk of <= 64 entries (sorted by destination slot) against LDS buffer `lbuf`
    auto consume = [&](const int2& ent, const char* lbuf) {
        const int rowl = ent.x >> 8;
        int e_prev = 0;
#pragma unroll
        for (int i = 0; i < kRPW; ++i) {
            const int e = __popcll(__ballot(rowl <= i));
            int j = e_prev;
            for (; j + 1 < e; j += 2) {                       // two independent LDS reads in flight
                const int c0 = __builtin_amdgcn_readlane(ent.x, j) & 0xFF, c1 = __builtin_amdgcn_readlane(ent.x, j + 1) & 0xFF;
                const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ent.y, j));
                const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ent.y, j + 1));
                if (active) {
                    const float4 x0 = *reinterpret_cast<const float4*>(lbuf + c0 * row_bytes);
                    const float4 x1 = *reinterpret_cast<const float4*>(lbuf + c1 * row_bytes);
                    fma4(acc[i], w0, x0);
                    fma4(acc[i], w1, x1);
                }
            }
            if (j < e) {
                const int c0 = __builtin_amdgcn_readlane(ent.x, j) & 0xFF;
                const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ent.y, j));
                if (active) fma4(acc[i], w0, *reinterpret_cast<const float4*>(lbuf + c0 * row_bytes));
            }
            e_prev = e;
        }
    };
    // one source block: `cur*` were loaded a block ago; `nxt*` are loaded now for block b+1
    auto block = [&](int b, int& cs, int& ce0, const int2& cur0, const int2& cur1, int ns, int ne, int2& nxt0, int2& nxt1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA pieces of block b and my entry chunks have landed
        __syncthreads();                                      // everyone's have; everyone is done with block b-1
        if (b + 1 < nblk) {
            load_chunk(ns, ne, nxt0);                         // (ns, ne) were fetched a block ago
            load_chunk(ns + 64, ne, nxt1);
            if (do_fill) fill(b + 1, (b + 1) & 1);            // DMA of the next block overlaps this block's FMAs
        }
        const int cs_ = cs, ce_ = ce0;                        // this block's segment (copied: cs/ce0 are reloaded below)
        if (b + 2 < nblk) { cs = seg[(b + 2) * kTW]; ce0 = seg[(b + 2) * kTW + 1]; }
        if (!do_comp) return;
        const char* lbuf = smem + (b & 1) * buf_bytes + lane * 16;
        int q = 0;
        for (int s = cs_; s < ce_ || q == 0; s += 64, ++q) {   // usually one chunk; > 128 entries per wave-block is rare
            int2 ent = cur0;
            if (q == 1) ent = cur1;
            if (q >= 2) load_chunk(s, ce_, ent);
            consume(ent, lbuf);
        }
    };

    if (nblk > 0) {
        int sA = seg[0], eA = seg[1], sB = 0, eB = 0;
        if (nblk > 1) { sB = seg[kTW]; eB = seg[kTW + 1]; }
        int2 a0, a1, b0, b1;
        load_chunk(sA, eA, a0);
        load_chunk(sA + 64, eA, a1);
        b0 = b1 = make_int2(0xFF00, 0);
        if (do_fill) fill(0, 0);
        for (int b = 0; b < nblk; b += 2) {                   // unrolled by two so the chunk registers swap roles statically
            block(b, sA, eA, a0, a1, sB, eB, b0, b1);
            if (b + 1 < nblk) block(b + 1, sB, eB, b0, b1, sA, eA, a0, a1);
        }
    }

    const int4* __restrict__ items = t.tile_items + (size_t)tile * kTileRows + wave * kRPW;
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

// ---------------------------------------------------------------------------------------------
// D == 256 specialisation: FLAT entry loop.  The 16 x float4 accumulators of a wave are pinned to
// v[64:127] and updated through GPR-index mode (s_set_gpr_idx_on: VGPR number += M0[7:0]) so the
// destination row of an entry can be a run-time value: no per-row control flow, every entry costs
// 2 v_readlane + 1 ds_read_b128 + 2 v_pk_fma_f32, and LDS waits are static counts.
// ---------------------------------------------------------------------------------------------
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WGNN_FMA_IDX(R4_, W_, X_)                                                                \
    do {                                                                                         \
        const float4 xx_ = (X_);                                                                 \
        const f32x2 lo_ = {xx_.x, xx_.y}, hi_ = {xx_.z, xx_.w};                                  \
        const unsigned long long ww_ = (unsigned)(W_);                                           \
        asm volatile("s_set_gpr_idx_on %[ri], gpr_idx(SRC2,DST)\n\t"                            \
                     "v_pk_fma_f32 v[64:65], %[ww], %[lo], v[64:65] op_sel_hi:[0,1,1]\n\t"       \
                     "v_pk_fma_f32 v[66:67], %[ww], %[hi], v[66:67] op_sel_hi:[0,1,1]\n\t"       \
                     "s_set_gpr_idx_off"                                                         \
                     : "+{v[64:95]}"(accA), "+{v[96:127]}"(accB)                                 \
                     : [ri] "s"(R4_), [ww] "s"(ww_), [lo] "v"(lo_), [hi] "v"(hi_)                \
                     : "m0");                                                                    \
    } while (0)

template <typename TOut, int EPI>
__global__ void __launch_bounds__(kTW * 64) agg_tiled_flat(const KArgs a, const TArgs t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int row_bytes = 1024, buf_bytes = kKB * row_bytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x;
    const int2 hdr = t.tile_hdr[tile];
    const int cb = __builtin_amdgcn_readfirstlane(hdr.x), ce = __builtin_amdgcn_readfirstlane(hdr.y);
    const int nblk = (ce - cb + kKB - 1) / kKB;
    cptr_t seg = (cptr_t)(t.seg_ptr + ((size_t)tile * t.nblk_max) * kTW + wave);
    const bool do_fill = !(a.flags & kDbgNoFill), do_comp = !(a.flags & kDbgNoCompute);

    f32x32 accA = 0.f, accB = 0.f;                   // rows 0..7 -> v[64:95], rows 8..15 -> v[96:127]

    auto fill = [&](int b, int buf) {
        const int r0 = cb + b * kKB;
        const int nbytes = min(kKB, ce - r0) * row_bytes;
        const char* g = reinterpret_cast<const char*>(a.src) + (size_t)r0 * row_bytes;
        char* l = smem + buf * buf_bytes;
        for (int p = wave; p * 1024 < nbytes; p += kTW)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + p * 1024 + lane * 16), (lptr_t)(l + p * 1024), 16, 0, 0);
    };
    auto load_chunk = [&](int s, int e, int2& ent) {
        const int idx = s + lane;
        ent = make_int2(0, 0);
        if (idx < e) ent = t.entries[idx];
    };
    // n (<= 64) entries of one chunk; meta = dst_slot<<8 | src_local
    auto consume = [&](const int2& ent, int n, const char* lbuf) {
        const int pk = ((ent.x >> 8) << 18) | ((ent.x & 0xFF) << 10);      // (4*slot)<<16 | src_local*1024
        int j = 0;
        for (; j + 3 < n; j += 4) {
            const int m0 = __builtin_amdgcn_readlane(pk, j), m1 = __builtin_amdgcn_readlane(pk, j + 1);
            const int m2 = __builtin_amdgcn_readlane(pk, j + 2), m3 = __builtin_amdgcn_readlane(pk, j + 3);
            const float4 x0 = *reinterpret_cast<const float4*>(lbuf + (m0 & 0xFFFF));
            const float4 x1 = *reinterpret_cast<const float4*>(lbuf + (m1 & 0xFFFF));
            const float4 x2 = *reinterpret_cast<const float4*>(lbuf + (m2 & 0xFFFF));
            const float4 x3 = *reinterpret_cast<const float4*>(lbuf + (m3 & 0xFFFF));
            const int w0 = __builtin_amdgcn_readlane(ent.y, j), w1 = __builtin_amdgcn_readlane(ent.y, j + 1);
            const int w2 = __builtin_amdgcn_readlane(ent.y, j + 2), w3 = __builtin_amdgcn_readlane(ent.y, j + 3);
            WGNN_FMA_IDX(m0 >> 16, w0, x0);
            WGNN_FMA_IDX(m1 >> 16, w1, x1);
            WGNN_FMA_IDX(m2 >> 16, w2, x2);
            WGNN_FMA_IDX(m3 >> 16, w3, x3);
        }
        for (; j < n; ++j) {
            const int m0 = __builtin_amdgcn_readlane(pk, j);
            const float4 x0 = *reinterpret_cast<const float4*>(lbuf + (m0 & 0xFFFF));
            const int w0 = __builtin_amdgcn_readlane(ent.y, j);
            WGNN_FMA_IDX(m0 >> 16, w0, x0);
        }
    };
    auto block = [&](int b, int& cs, int& ce0, const int2& cur0, const int2& cur1, int ns, int ne, int2& nxt0, int2& nxt1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (b + 1 < nblk) {
            load_chunk(ns, ne, nxt0);                         // (ns, ne) were fetched a block ago
            load_chunk(ns + 64, ne, nxt1);
            if (do_fill) fill(b + 1, (b + 1) & 1);
        }
        const int cs_ = cs, ce_ = ce0;
        if (b + 2 < nblk) { cs = seg[(b + 2) * kTW]; ce0 = seg[(b + 2) * kTW + 1]; }
        if (!do_comp) return;
        const char* lbuf = smem + (b & 1) * buf_bytes + lane * 16;
        int q = 0;
        for (int s = cs_; s < ce_; s += 64, ++q) {
            int2 ent = cur0;
            if (q == 1) ent = cur1;
            if (q >= 2) load_chunk(s, ce_, ent);
            consume(ent, min(64, ce_ - s), lbuf);
        }
    };

    if (nblk > 0) {
        int sA = seg[0], eA = seg[1], sB = 0, eB = 0;
        if (nblk > 1) { sB = seg[kTW]; eB = seg[kTW + 1]; }
        int2 a0, a1, b0, b1;
        load_chunk(sA, eA, a0);
        load_chunk(sA + 64, eA, a1);
        b0 = b1 = make_int2(0, 0);
        if (do_fill) fill(0, 0);
        for (int b = 0; b < nblk; b += 2) {
            block(b, sA, eA, a0, a1, sB, eB, b0, b1);
            if (b + 1 < nblk) block(b + 1, sB, eB, b0, b1, sA, eA, a0, a1);
        }
    }

    const int4* __restrict__ items = t.tile_items + (size_t)tile * kTileRows + wave * kRPW;
#pragma unroll
    for (int i = 0; i < kRPW; ++i) {
        const int4 it = items[i];
        const int slot = __builtin_amdgcn_readfirstlane(it.x), pslot = __builtin_amdgcn_readfirstlane(it.w);
        if (slot < 0) continue;
        float4 v;
        if (i < 8) v = make_float4(accA[4 * (i & 7)], accA[4 * (i & 7) + 1], accA[4 * (i & 7) + 2], accA[4 * (i & 7) + 3]);
        else       v = make_float4(accB[4 * (i & 7)], accB[4 * (i & 7) + 1], accB[4 * (i & 7) + 2], accB[4 * (i & 7) + 3]);
        if (pslot >= 0) {
            st4(a.partials + (size_t)pslot * a.D + lane * 4, v);
        } else {
            float4 one[1] = {v};
            epilogue<64, 1, float, TOut, EPI>(a, one, slot, lane, true);
        }
    }
}

template <typename TOut, int EPI>
int launch_tiled(const KArgs& a, const TArgs& t, long n_tiles, hipStream_t st) {
    const int lds = 2 * kKB * a.D * (int)sizeof(float);
    static int configured = 0;                       // per instantiation
    if (configured < lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_tiled<TOut, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return WGNN_ERR_LAUNCH;
        configured = lds;
    }
    if (a.D == 256 && !(a.flags & (1u << 19))) {       // bit 19: force the generic (row-visit) kernel, for A/B timing
        static bool flat_configured = false;
        if (!flat_configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_tiled_flat<TOut, EPI>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return WGNN_ERR_LAUNCH;
            flat_configured = true;
        }
        hipLaunchKernelGGL((agg_tiled_flat<TOut, EPI>), dim3((unsigned)n_tiles), dim3(kTW * 64), lds, st, a, t);
    } else {
        hipLaunchKernelGGL((agg_tiled<TOut, EPI>), dim3((unsigned)n_tiles), dim3(kTW * 64), lds, st, a, t);
    }
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

}  // namespace

namespace wgnn {
int launch_finalize_fwd_f32(const KArgs& a, hipStream_t st);     // defined in wgnn_kernels.hip
}

extern "C" int wgnn_agg_fwd_tiled(const int32_t* rowptr, const float* alpha, int alpha_mode, int32_t self_idx,
                                  const float* h_src, int64_t n_src, float* src_scratch,
                                  const float* h_self, int64_t ld_self,
                                  const int32_t* row_ids, const float* inv_deg, const float* bias,
                                  float* out, int64_t ld_out, int64_t n_out, int32_t D, uint32_t flags,
                                  const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max,
                                  const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                                  const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                                  void* stream) {
    if (!h_src || !out || n_out < 0 || n_tiles < 0 || n_src < 0 || nblk_max < 0) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (!inv_deg && !rowptr && !(flags & WGNN_FLAG_NO_MEAN)) return WGNN_ERR_BAD_ARG;
    if (D <= 0 || D % 4 || ld_out % 4 || (h_self && ld_self % 4)) return WGNN_ERR_ALIGNMENT;
    if (D > 256) return WGNN_ERR_UNSUPPORTED;                     // one float4 per lane; 2 x 64 x D x 4 B of LDS
    if (!aligned16(h_src) || !aligned16(out) || (h_self && !aligned16(h_self)) || (bias && !aligned16(bias)))
        return WGNN_ERR_ALIGNMENT;
    if (n_tiles > 0 && (!tile_items || !tile_hdr || !entries || !seg_ptr)) return WGNN_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !partials || n_partials <= 0)) return WGNN_ERR_WORKSPACE;
    if (alpha_mode == WGNN_SRC_IS_GENE && (!src_scratch || !aligned16(src_scratch))) return WGNN_ERR_WORKSPACE;
    if (n_out == 0 || n_tiles == 0) return WGNN_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* src = h_src;
    if (alpha_mode == WGNN_SRC_IS_GENE) {                         // (h*alpha) once per source row, gnn.py:54
        const long n4 = (long)n_src * (D / 4);
        hipLaunchKernelGGL(scale_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, h_src, alpha, src_scratch,
                           (long)n_src, D / 4);
        src = src_scratch;
    }
    KArgs a{};
    a.rowptr = rowptr;
    a.src = src; a.ld_src = D; a.alpha = alpha; a.mode = alpha_mode; a.self_idx = self_idx;
    a.self = h_self; a.ld_self = ld_self; a.row_ids = row_ids; a.inv_deg = inv_deg; a.bias = bias;
    a.out = out; a.ld_out = ld_out; a.D = D; a.flags = flags;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    TArgs t{reinterpret_cast<const int2*>(entries), seg_ptr, reinterpret_cast<const int4*>(tile_items),
            reinterpret_cast<const int2*>(tile_hdr), nblk_max};
    int rc = launch_tiled<float, EPI_FWD>(a, t, n_tiles, st);
    if (rc) return rc;
    if (n_long > 0) return launch_finalize_fwd_f32(a, st);
    return WGNN_OK;
}

// wgnn_tiled.hip - LDS-streamed aggregation kernels for CDNA4 (gfx950).
//
// Same arithmetic as agg_main (reference models/gnn.py:47-56,65 + fused bias/ReLU, gnn.py:20-22),
// different data movement.  The row-wave kernel re-gathers every source row from L2/MALL once per
// non-zero (nnz*D*4 B = 82 GB per pass at BASELINE cfg3) and is bound by the ~64 B/clk/CU vector
// memory path.  Here one 1024-thread workgroup (16 waves, one per CU) owns a TILE of up to 256
// destination rows - 16 per wave, accumulators resident in VGPRs (16 x float4 per lane) - and the
// SOURCE table is streamed through LDS in blocks of `kb` rows (78 KiB at D=256) with the async
// global->LDS DMA (global_load_lds_dwordx4), double-buffered.  Every staged source row is consumed
// by all rows of the tile that reference it, so the per-non-zero gather becomes a conflict-free
// ds_read_b128 (256 B/clk/CU) and global traffic drops to (rows/256) x |source table|.
//
// The edges of a tile are pre-sorted at plan time into (block, wave, destination row) order
// ("entries": {dst_slot<<8 | src_row_in_block, weight}), so that in every block a wave fetches its
// share ("chunk") with ONE coalesced 8-byte load per lane, issued ahead of time so that its latency
// hides behind earlier blocks' FMAs, and broadcasts (source row, weight) with v_readlane / an LDS
// broadcast read.  alpha[k(e)] for gene->cell edges is a per-source-row factor, so it is applied to
// the source table once (scale_rows) instead of per edge - which is also the reference's own
// multiply order, (h*alpha)*w.  Tiles carry a column range so that hub rows (genes expressed in
// ~every cell) are split across workgroups; their partial sums are folded by agg_finalize in a
// fixed order (deterministic, no atomics).
//
// Two kernels: agg_tiled (any D <= 256: per-row ballot visits, compiler-scheduled) and
// agg_tiled_flat4 (D == 256: generated straight-line ISA, see below and gen_flat_asm.py).
#include <atomic>
#include <type_traits>
#include "wgnn_common.h"
#include "wgnn_flat_asm.inc"

namespace {
using namespace wgnn;

constexpr int kTW = 16;                       // waves per tile workgroup
constexpr int kRPW = 16;                      // destination rows per wave
constexpr int kKBDefault = 64;                // source rows per LDS block (TArgs::kb; 2*kb*D*4 B of LDS)
constexpr int kTileRows = kTW * kRPW;         // 256
constexpr int kWStripBytes = kTW * 256;       // flat kernel: one 64-entry weight strip per wave
// ablation switches (timing experiments only; results are wrong when set)
constexpr unsigned kDbgNoFill = 1u << 16, kDbgNoCompute = 1u << 17, kDbgNoBarrier = 1u << 18;
constexpr unsigned kDbgSkipHubShift = 21;          // bits 21..23 = n: skip the entry pipeline of the first 2n LDS blocks (prices a dense treatment of hub sources)
constexpr unsigned kDbgFillToVgpr = 1u << 20;      // the fill's loads go to scratch VGPRs instead of LDS: same VMEM issue / L2 traffic, no LDS writes

// LDS row stride of agg_tiled_flat4 (host and device): 256 B (D <= 64), 512 B (D <= 128) or 1 KiB.  A power of two that
// covers the row: the per-lane LDS address is {row address} OR {16 * lane} (one v_and_or_b32 per entry), which only equals
// the sum while the lanes that carry data (16 * lane < D * 4 <= stride) stay below the row address's lowest set bit - a
// 768-byte stride for D <= 192 would pack closer but breaks exactly that (built and caught by the parity tests).
__host__ __device__ inline int flat_lds_row_bytes(int D) { return D <= 64 ? 256 : (D <= 128 ? 512 : 1024); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(4))) int* cptr_t;      // immutable plan data -> scalar (SMEM) loads

struct TArgs {
    const int2* entries;      // {dst_slot<<8 | src_local, weight bits}, sorted by (tile, block, wave, dst_slot)
    const int* seg_ptr;       // [(n_tiles*nblk_max*16) + 1] entry offsets per (tile, block, wave)
    const int4* tile_items;   // [n_tiles*256] {row_slot|-1, -, -, partial_slot|-1}
    const int2* tile_hdr;     // [n_tiles] {col_begin, col_end}
    int nblk_max;
    int kb;                   // source rows per LDS block
    int tall;                 // plan geometry: 0 = 16 waves x 16 rows per tile, 1 = 8 waves x 49 rows (WGNN_PLAN_TALL)
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
    const int kKB = t.kb;
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
    // consume one chunk of <= 64 entries against LDS buffer `lbuf`, in entry order (any order of slots: the plan
    // groups a segment's entries for the flat kernel's shared pairs, whose marks in the meta word's high half are
    // ignored here)
    auto consume = [&](const int2& ent, const char* lbuf) {
        const int n = __popcll(__ballot((ent.x & 0xFF00) != 0xFF00));
        for (int j = 0; j < n; ++j) {
            const int x = __builtin_amdgcn_readlane(ent.x, j);
            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ent.y, j));
            const int slot = (x >> 8) & 0xF;
            if (active) {
                const float4 xv = *reinterpret_cast<const float4*>(lbuf + (x & 0xFF) * row_bytes);
#pragma unroll
                for (int i = 0; i < kRPW; ++i)
                    if (slot == i) fma4(acc[i], w, xv);
            }
        }
    };
    // one source block: `cur*` were loaded a block ago; `nxt*` are loaded now for block b+1
    auto block = [&](int b, int& cs, int& ce0, const int2& cur0, const int2& cur1, int ns, int ne, int2& nxt0, int2& nxt1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA pieces of block b and my entry chunks have landed
        __syncthreads();                                      // everyone's have; everyone is done with block b-1
        // (ns, ne) were fetched with scalar loads at the END of the previous block; retire them here so that no SMEM
        // is outstanding during the FMA loop (a pending SMEM forces every LDS wait to lgkmcnt(0))
        asm volatile("" ::"s"(ns), "s"(ne));
        if (b + 1 < nblk) {
            load_chunk(ns, ne, nxt0);
            load_chunk(ns + 64, ne, nxt1);
            if (do_fill) fill(b + 1, (b + 1) & 1);            // DMA of the next block overlaps this block's FMAs
        }
        const int cs_ = cs, ce_ = ce0;                        // this block's segment
        const char* lbuf = smem + (b & 1) * buf_bytes + lane * 16;
        int q = 0;
        for (int s = cs_; (s < ce_ || q == 0) && do_comp; s += 64, ++q) {   // usually one chunk; > 128 per wave-block is rare
            int2 ent = cur0;
            if (q == 1) ent = cur1;
            if (q >= 2) load_chunk(s, ce_, ent);
            consume(ent, lbuf);
        }
        asm volatile("" ::: "memory");
        if (b + 2 < nblk) { cs = seg[(b + 2) * kTW]; ce0 = seg[(b + 2) * kTW + 1]; }   // block b+2's segment, used in block b+1
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

// Epilogue shared by the two geometries of the flat tile kernel (16 waves x 16 rows, 8 waves x 49 rows): turns the wave's
// accumulator rows into the kernel's outputs.  `items_base` = first item of this wave (tile * rows per tile + wave * RPW),
// `acc_row(i)` copies accumulator row i of the wave into compiler registers (literal-register asm of the caller).
template <typename TOut, int EPI, int RPW, typename AccRow>
__device__ __forceinline__ void tile_epilogue(const TArgs& t, size_t items_base, int lane, AccRow acc_row) {
    // The epilogue re-reads its arguments from the kernarg segment through a pointer the optimiser cannot see through:
    // otherwise every epilogue-only field of `a` (output / self / bias / alpha / inv_deg pointers, strides, partial-sum base)
    // is loaded in the prologue and kept in SGPRs across the whole block loop, which overflows the s[0:79] the compiler
    // owns here and spills into VGPR lanes (round 3: 3 .. 20 spilled SGPRs per instantiation, parked in a VGPR outside
    // the declared budget).  `a` is the kernel's first parameter, i.e. offset 0 of the kernarg segment.
    typedef const __attribute__((address_space(4))) KArgs* kargs_cptr_t;
    kargs_cptr_t a_late = (kargs_cptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a_late));
    {                                                    // ---- epilogue scope
    const KArgs a = *(const KArgs*)a_late;                             // shadows the parameter from here on (dead fields are never loaded)
    const int4* __restrict__ items = t.tile_items + items_base;
    if constexpr (EPI == EPI_FWD && std::is_same<TOut, float>::value) {
        if (!a.row_ids) {
            // Forward epilogue, two rows per trip: the rows' item words come through the scalar cache and the self rows of
            // BOTH rows are requested before either is used.  The row-at-a-time form below chains an item load, an
            // inv_deg / alpha load and a self-row load per row - ~2 k clk each, 14 rows per wave, ~10 us per tile of pure
            // latency (the computing waves of a tile all sit in it at the same time).  Same arithmetic, same order.
            cptr_t sitems = (cptr_t)items;                      // int4 items as dwords: .x at 4 i, .w at 4 i + 3
            const bool lane_on = lane * 4 < a.D;
            const bool no_mean = a.flags & WGNN_FLAG_NO_MEAN, relu = a.flags & WGNN_FLAG_RELU;
            const bool has_self = !(a.flags & WGNN_FLAG_NO_SELF) && a.self != nullptr;
            const bool out_scale = (a.flags & WGNN_FLAG_OUT_SCALE_ALPHA) && a.mode == WGNN_DST_IS_GENE;
            // per-row factors (inv_deg, alpha: inputs nobody writes during the launch) come through the scalar cache like
            // the item words: the rows are wave-uniform, and as SGPR values they cost no vector registers - the compiler
            // owns v[0:31] only and this epilogue is the place where it needs them all
            typedef const __attribute__((address_space(4))) float* cfptr_t;
            cfptr_t s_alpha = (cfptr_t)a.alpha, s_inv_deg = (cfptr_t)a.inv_deg;
            const float a_self = has_self ? (a.mode == WGNN_NO_ALPHA ? 1.0f : s_alpha[a.self_idx]) : 0.0f;
            const float* selfp = reinterpret_cast<const float*>(a.self);
            float* outp = reinterpret_cast<float*>(a.out);
            for (int i0 = 0; i0 < RPW; i0 += 2) {
                int slot[2], pslot[2];
                float4 sf[2];
                float invd[2], rs[2], post[2];
                // the lane's byte offset inside a row, opaque per trip: every access below is then "uniform row base
                // (SGPR pair) + 32-bit lane offset" - left visible, the offset is folded into each of the four base
                // pointers ahead of the loop, i.e. four 64-bit per-lane addresses (8 VGPRs) held across it
                unsigned lo = (unsigned)lane * 16u;
                asm volatile("" : "+v"(lo));
                typedef float f4v __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(1))) f4v* gf4_t;                             // global_load / global_store, saddr form
                auto at = [&](const float* base, size_t row, long ld) {
                    __attribute__((address_space(1))) char* rb =                                  // wave-uniform row base
                        (__attribute__((address_space(1))) char*)const_cast<float*>(base + row * ld);
                    asm volatile("" : "+s"(rb));                                                  // ... kept in an SGPR pair
                    return (gf4_t)(rb + lo);
                };
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    slot[k] = -1; pslot[k] = -1;                  // (an odd RPW - the tall tile's 49 - leaves the last trip one row)
                    if (i0 + k < RPW) { slot[k] = sitems[4 * (i0 + k)]; pslot[k] = sitems[4 * (i0 + k) + 3]; }
                }
                float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);     // requested per trip, together with the self rows (1 KiB, L1-resident)
                if (a.bias && lane_on) { const f4v t = *at(a.bias, 0, 0); bias4 = make_float4(t.x, t.y, t.z, t.w); }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    sf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (slot[k] >= 0 && pslot[k] < 0 && has_self && lane_on)
                        { const f4v t = *at(selfp, slot[k], a.ld_self); sf[k] = make_float4(t.x, t.y, t.z, t.w); }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    invd[k] = 1.0f; rs[k] = 1.0f; post[k] = 1.0f;
                    if (slot[k] >= 0 && pslot[k] < 0) {
                        if (!no_mean) invd[k] = a.inv_deg ? s_inv_deg[slot[k]] : 1.0f / (row_degree(a, slot[k]) + 1.0f);
                        const float a_row = a.mode == WGNN_DST_IS_GENE ? s_alpha[slot[k]] : 1.0f;   // one load serves both uses
                        rs[k] = invd[k] * a_row;
                        post[k] = a_row;
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (slot[k] < 0) continue;
                    float4 o = acc_row(i0 + k);
                    if (pslot[k] >= 0) {
                        if (lane_on) *at(a.partials, pslot[k], a.D) = f4v{o.x, o.y, o.z, o.w};
                        continue;
                    }
                    if (a.aux1 && lane_on) *at(a.aux1, slot[k], a.D) = f4v{o.x, o.y, o.z, o.w};                      // raw neighbour sum
                    o.x *= rs[k]; o.y *= rs[k]; o.z *= rs[k]; o.w *= rs[k];
                    if (has_self) fma4(o, invd[k] * a_self, sf[k]);
                    if (a.bias) { o.x += bias4.x; o.y += bias4.y; o.z += bias4.z; o.w += bias4.w; }
                    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    if (out_scale) { o.x *= post[k]; o.y *= post[k]; o.z *= post[k]; o.w *= post[k]; }
                    if (lane_on) *at(outp, slot[k], a.ld_out) = f4v{o.x, o.y, o.z, o.w};
                }
            }
            return;
        }
    }
    for (int i = 0; i < RPW; ++i) {
        const int4 it = items[i];
        const int slot = __builtin_amdgcn_readfirstlane(it.x), pslot = __builtin_amdgcn_readfirstlane(it.w);
        if (slot < 0) continue;
        const float4 v = acc_row(i);
        if (pslot >= 0) {
            if (lane * 4 < a.D) st4(a.partials + (size_t)pslot * a.D + lane * 4, v);
        } else {
            float4 one[1] = {v};
            epilogue<64, 1, float, TOut, EPI>(a, one, slot, lane, true);
        }
    }
    }                                                    // ---- epilogue scope
}

// clobber list of a hand-written statement: EVERY register of the hand-owned file by name (gen_flat_asm.py spells them out)
#define WGNN_CLOB "m0", "memory", "scc", WGNN_HAND_VGPRS, WGNN_HAND_SGPRS

// ---------------------------------------------------------------------------------------------
// agg_tiled_flat4 - the D == 256 specialisation.  Same tile / block structure as agg_tiled, but
//  * the VECTOR register file is split: the compiler may allocate v[0:31] only.  amdgpu_num_vgpr(16), not 32: on
//    gfx90a+ the attribute counts the unified VGPR + AGPR file and the allocator takes up to TWICE the figure as plain
//    VGPRs (probe kernel: 16 -> v0..v31, 24 -> v0..v47, 32 -> v0..v63) - rounds 1-3 declared 32 and the compiler in fact
//    owned v0..v63, overlapping the hand-owned registers (its epilogue used v32..v35; harmless there, unguarded anywhere);
//    v[64:127] are the wave's 16 x float4 accumulators, v[48:63] two LDS staging buffers, v[32:46] chunk / segment /
//    weight / address registers.  They carry state ACROSS statements and are touched exclusively by literal-register
//    inline asm, so the compiler never copies or spills them.  The contract is ENFORCED at build time
//    (build.audit_flat4: 0 spilled SGPRs / VGPRs, no scratch, and no compiler-emitted instruction outside ;;#ASMSTART ..
//    ;;#ASMEND names v32..v127 - a spill lane parked in the hand-owned file would corrupt results silently) and every
//    hand-written statement names the whole hand-owned file in its clobber list.  The literal SCALAR registers s[80:95]
//    (per-entry scalars, DMA addresses) are statement-local - written and read inside ONE asm statement that lists them
//    as clobbers - so the compiler may use them between statements and no SGPR budget is imposed (rounds 1-3 capped it
//    at s[0:79], which cost 3 .. 20 spilled SGPRs);
//  * the destination row of an entry is a run-time value: the accumulators are addressed through GPR-index mode
//    (s_set_gpr_idx_on: VGPR number += M0[7:0]), so there is no per-row control flow;
//  * the per-entry loop is the generated straight-line pipeline of gen_flat_asm.py (4 VALU per entry);
//  * EVERY vector-memory operation of the block loop - entry-chunk loads, segment loads, the global->LDS DMA - is
//    issued from inline asm into literal registers, with hand-counted s_waitcnt.  The compiler otherwise protects
//    each VGPR load result / LDS access with `s_waitcnt vmcnt(0)` placed AFTER the next block's DMA has been issued,
//    which serialises fill and compute.
//   v[32:33] segment {begin, end} of block b+3 ; v[34:35], v[36:37], v[38:39] entry chunks of blocks b, b+1, b+2 (mod 3)
// VMEM issue order inside block b:  S(b+3), DMA pieces of block b+1, E(b+2)   =>  at the top of block b+1 everything
// but E(b+2) must have landed: s_waitcnt vmcnt(1).  An entry chunk therefore has two block times to arrive.
// ---------------------------------------------------------------------------------------------
template <int SET> __device__ __forceinline__ int2 chunk_get() {
    int2 e;
    if constexpr (SET == 0) asm volatile("v_mov_b32 %0, v34\n\tv_mov_b32 %1, v35" : "=v"(e.x), "=v"(e.y)::"memory");
    if constexpr (SET == 1) asm volatile("v_mov_b32 %0, v36\n\tv_mov_b32 %1, v37" : "=v"(e.x), "=v"(e.y)::"memory");
    if constexpr (SET == 2) asm volatile("v_mov_b32 %0, v38\n\tv_mov_b32 %1, v39" : "=v"(e.x), "=v"(e.y)::"memory");
    return e;
}

template <typename TOut, int EPI, bool DBG>
__global__ void __launch_bounds__(kTW * 64) __attribute__((amdgpu_num_vgpr(16)))
agg_tiled_flat4(const KArgs a, const TArgs t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS row stride: 256 B / 512 B / 1 KiB, the power of two that covers the row (flat_lds_row_bytes), so that an LDS row
    // address has zero low byte and {row address (bits 8..17) | 4*slot (bits 2..5)} packs into one dword.  Rounds 2-3 kept
    // 1 KiB whatever D was: at D = 128 half of every LDS block was dead (78 rows per block where 156 fit).  Packing halves
    // the barriers per edge - and measured neutral (round 4, scratch/narrow_rows.py): the kernel is not bound by them.
    const int row_bytes = flat_lds_row_bytes(a.D);
    const int g_row = a.D * (int)sizeof(float);          // global row stride: D <= 256 floats (lanes >= D/4 carry nothing)
    const int n4 = a.D >> 2;                             // lanes that move 16 B of a row in the global->LDS DMA
    const int kKB = t.kb, buf_bytes = kKB * row_bytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x;
    const int2 hdr = t.tile_hdr[tile];
    const int cb = __builtin_amdgcn_readfirstlane(hdr.x), ce = __builtin_amdgcn_readfirstlane(hdr.y);
    const int nblk = (ce - cb + kKB - 1) / kKB;
    const int* seg = t.seg_ptr + ((size_t)tile * t.nblk_max) * kTW + wave;     // seg[b*16], seg[b*16+1]
    const unsigned dbg = DBG ? a.flags : 0u;             // ablation switches exist in the DBG instantiation only
    const bool do_fill = !(dbg & kDbgNoFill), do_comp = !(dbg & kDbgNoCompute), do_barrier = !(dbg & kDbgNoBarrier);
    int hub_left = DBG ? 2 * (int)((dbg >> kDbgSkipHubShift) & 7u) : 0;

    for (int r = 0; r < kRPW; ++r)                       // zero the accumulators v[64:127]
        asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\t"
                     "v_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\ts_set_gpr_idx_off" ::"s"(r * 4) : WGNN_CLOB);

    const int lane16 = lane * 16, row_mask = 0x3FF00;     // LDS row address bits of a packed entry (rows start at multiples of 256)
    // global -> LDS DMA of `rows` source rows starting at global row r0 into LDS buffer `buf`: one 1 KiB row per
    // wave-instruction; the first `nw` waves take part, wave w takes rows w, w + nw, ... (nw = 16: every wave, <= 5 pieces
    // per block; nw = 2 dedicated loader waves: 39 each).  Scalar base + lane offset addressing: no VALU, 6 SALU per row.
    auto fill_rows = [&](int r0, int rows, int buf, int nw) {
        // which rows this wave takes: interleaved (w, w + nw, ...) in the general form; a contiguous run when the rows are
        // equally spaced in global memory and in LDS (the grouped form below rides the instruction's immediate offset)
        const bool contiguous = g_row == row_bytes && !(DBG && (dbg & kDbgFillToVgpr));
        const int per = (rows + nw - 1) / nw;
        const int first = contiguous ? wave * per : wave;
        const int np = contiguous ? max(0, min(per, rows - first)) : (rows > wave ? (rows - wave + nw - 1) / nw : 0);
        if (np == 0) return;
        const char* g = reinterpret_cast<const char*>(a.src) + ((size_t)r0 + first) * g_row;
        const int l = (int)(size_t)smem + buf * buf_bytes + first * row_bytes;
        if (DBG && (dbg & kDbgFillToVgpr)) {                  // ablation: identical loads, no LDS writes (16-wave form only)
#define WGNN_FILLV_NEXT(K)                                                                                  \
        "s_cmp_lt_u32 %[np], " #K "\n\ts_cbranch_scc1 .Lw4_fv_%=\n\t"                                        \
        "s_add_u32 s94, s94, 0x4000\n\ts_addc_u32 s95, s95, 0\n\t"                                          \
        "global_load_dwordx4 v[40:43], %[vo], s[94:95]\n\t"
            asm volatile("s_mov_b64 s[94:95], %[g]\n\t"
                         "global_load_dwordx4 v[40:43], %[vo], s[94:95]\n\t"
                         WGNN_FILLV_NEXT(2) WGNN_FILLV_NEXT(3) WGNN_FILLV_NEXT(4) WGNN_FILLV_NEXT(5)
                         ".Lw4_fv_%=:"
                         ::[g] "s"(g), [vo] "v"(lane16), [np] "s"(np)
                         : "memory", "scc", "s94", "s95", "v40", "v41", "v42", "v43");
#undef WGNN_FILLV_NEXT
            return;
        }
        // EXEC is narrowed to the D/4 lanes that carry data for the duration of the burst (a D < 256 row is shorter than
        // its 1 KiB LDS slot; the slot's tail is never read by a lane that is stored).
        int left = np;                                        // pieces still to issue (>= 1)
        if (g_row == row_bytes) {
            // Round 5 (scratch/stream_bench2.hip, profiles/r05_stream_bench2.txt).  A wave that advances its address registers
            // after EVERY piece issues one 1-KiB piece per ~52-63 clk: the update of the lane offset / of M0 waits for the
            // DMA in front of it to have read them.  The instruction's immediate offset moves BOTH the global and the LDS
            // address, so when the rows are equally spaced on both sides (D = 64 / 128 / 256: global row = LDS row) four
            // consecutive rows go out on ONE address setting: 33.6 clk per piece - a lone loader wave streams the cfg3 table
            // (10.5 GB per pass) in 0.56 ms instead of 1.04 (this loop form; 0.875 with the old loop unrolled by six), and the
            // advance is scalar (base + M0): the loader wave issues no VALU at all.  The wave takes a CONTIGUOUS run of rows.
#define WGNN_GROUP4(O1, O2, O3, STEP4, STEP1)                                                                                 \
            asm volatile("s_mov_b64 s[92:93], exec\n\ts_sub_u32 s91, 64, %[n4]\n\ts_lshr_b64 exec, s[92:93], s91\n\t"           \
                         "s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\t"                                                \
                         "s_cmp_lt_u32 %[left], 4\n\ts_cbranch_scc1 .Lw4_g1_%=\n\t"                                          \
                         ".Lw4_g4_%=:\n\t"                                                                                   \
                         "global_load_lds_dwordx4 %[vo], s[94:95]\n\t"                                                        \
                         "global_load_lds_dwordx4 %[vo], s[94:95] offset:" #O1 "\n\t"                                         \
                         "global_load_lds_dwordx4 %[vo], s[94:95] offset:" #O2 "\n\t"                                         \
                         "global_load_lds_dwordx4 %[vo], s[94:95] offset:" #O3 "\n\t"                                         \
                         "s_add_u32 m0, m0, " #STEP4 "\n\ts_add_u32 s94, s94, " #STEP4 "\n\ts_addc_u32 s95, s95, 0\n\t"        \
                         "s_sub_u32 %[left], %[left], 4\n\ts_cmp_ge_u32 %[left], 4\n\ts_cbranch_scc1 .Lw4_g4_%=\n\t"         \
                         "s_cmp_eq_u32 %[left], 0\n\ts_cbranch_scc1 .Lw4_ge_%=\n\t"                                          \
                         ".Lw4_g1_%=:\n\t"                                                                                   \
                         "global_load_lds_dwordx4 %[vo], s[94:95]\n\t"                                                        \
                         "s_add_u32 m0, m0, " #STEP1 "\n\ts_add_u32 s94, s94, " #STEP1 "\n\ts_addc_u32 s95, s95, 0\n\t"        \
                         "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lw4_g1_%=\n\t"         \
                         ".Lw4_ge_%=:\n\t"                                                                                   \
                         "s_mov_b64 exec, s[92:93]"                                                                          \
                         : [left] "+s"(left)                                                                                 \
                         : [g] "s"(g), [l] "s"(l), [vo] "v"(lane16), [n4] "s"(n4)                                            \
                         : "m0", "memory", "scc", "s91", "s92", "s93", "s94", "s95")
            if (row_bytes == 1024) { WGNN_GROUP4(1024, 2048, 3072, 4096, 1024); }
            else if (row_bytes == 512) { WGNN_GROUP4(512, 1024, 1536, 2048, 512); }
            else { WGNN_GROUP4(256, 512, 768, 1024, 256); }
#undef WGNN_GROUP4
            return;
        }
        // Rows spaced differently in global memory and in LDS (D not a power of two): one piece per address setting - the DMA,
        // M0 += nw LDS rows, lane offset += nw source rows (v46: a running 32-bit byte offset from the scalar base - a block's
        // rows span < 4 GiB).  Six pieces per trip while six are left, then one at a time.
#define WGNN_PIECE "global_load_lds_dwordx4 v46, s[94:95]\n\ts_add_u32 m0, m0, %[ms]\n\tv_add_u32 v46, s90, v46\n\t"
        asm volatile("s_mov_b64 s[92:93], exec\n\ts_sub_u32 s91, 64, %[n4]\n\ts_mul_i32 s90, %[n4], %[gs]\n\t"   // s90 = nw rows x D*4 B
                     "s_lshr_b64 exec, s[92:93], s91\n\t"                                                       // lanes 0 .. D/4-1
                     "s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\tv_mov_b32 v46, %[vo]\n\t"
                     "s_cmp_lt_u32 %[left], 6\n\ts_cbranch_scc1 .Lw4_f1_%=\n\t"
                     ".Lw4_f6_%=:\n\t"
                     WGNN_PIECE WGNN_PIECE WGNN_PIECE WGNN_PIECE WGNN_PIECE WGNN_PIECE
                     "s_sub_u32 %[left], %[left], 6\n\ts_cmp_ge_u32 %[left], 6\n\ts_cbranch_scc1 .Lw4_f6_%=\n\t"
                     "s_cmp_eq_u32 %[left], 0\n\ts_cbranch_scc1 .Lw4_fe_%=\n\t"
                     ".Lw4_f1_%=:\n\t"
                     WGNN_PIECE
                     "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lw4_f1_%=\n\t"
                     ".Lw4_fe_%=:\n\t"
                     "s_mov_b64 exec, s[92:93]"
                     : [left] "+s"(left)
                     : [g] "s"(g), [l] "s"(l), [vo] "v"(lane16), [n4] "s"(n4), [ms] "s"(nw * row_bytes), [gs] "s"(nw * 16)
                     : "m0", "memory", "scc", "s90", "s91", "s92", "s93", "s94", "s95", "v46");
#undef WGNN_PIECE
    };
    auto fill = [&](int b) { fill_rows(cb + b * kKB, min(kKB, ce - (cb + b * kKB)), b & 1, kTW); };   // every wave its share
    // RIGHT-aligned entry chunk of segment [s, e): with n = min(64, e - s) entries, lane j <- entry s + j - (64 - n);
    // the lanes in front of the chunk replicate its first entry (consume() zeroes their weight).  Always issues exactly
    // one load (the vmcnt bookkeeping depends on it), also for an empty segment (then: any valid entry).
    auto chunk_issue = [&](auto set, int s, int e) {
        constexpr int SET = decltype(set)::value;
        const int n = min(64, e - s);
        const int2* base = t.entries + (n > 0 ? s : max(e - 1, 0));
        const int off = max(lane - (64 - n), 0) * 8;                          // n == 0: 0
        if constexpr (SET == 0) asm volatile("global_load_dwordx2 v[34:35], %0, %1" ::"v"(off), "s"(base) : "memory", "v34", "v35");
        if constexpr (SET == 1) asm volatile("global_load_dwordx2 v[36:37], %0, %1" ::"v"(off), "s"(base) : "memory", "v36", "v37");
        if constexpr (SET == 2) asm volatile("global_load_dwordx2 v[38:39], %0, %1" ::"v"(off), "s"(base) : "memory", "v38", "v39");
    };
    auto seg_load = [&](const int* p) {                   // {begin, end} of one (block, wave) -> v[32:33]
        asm volatile("global_load_dwordx2 v[32:33], %0, %1" ::"v"(0), "s"(p) : "memory", "v32", "v33");
    };
    // n (1..64) entries of one chunk -> the generated straight-line pipeline.  The weights go through the wave's
    // 256-byte LDS strip (behind the two row buffers) and come back as broadcast reads.
    const int wstrip_addr = (int)(size_t)smem + 2 * buf_bytes + wave * 256;
    const int wlane_addr = wstrip_addr + lane * 4;
    auto consume = [&](const int2& ent, int n, int buf_addr) {
        // packed entry: LDS address of the source row (bits 8..17) | 4*slot (bits 0..7) | 4*slot of the pair's second
        // entry (bits 18..25; only the first entry of a shared pair carries one)
        const int pk = (buf_addr + (ent.x & 0xFF) * row_bytes) | (((ent.x >> 8) & 0xF) << 2) | (((ent.x >> 16) & 0xF) << 20);
        const bool mine = lane >= 64 - n;
        const int wv = mine ? ent.y : 0;                                   // padding lanes: weight 0
        const int m = (n + 1) >> 1;
        // Shared pairs (two entries of one source row; the plan puts them LAST in a segment, at even offsets) occupy the
        // last n_s/2 pair steps: from step `sw` on the pipeline continues in its shared-pair stream.  The first pair of a
        // chunk always runs unshared (the warm-up is that stream's), 32 = never switch.
        const int n_s = __popcll(__ballot(mine && ent.x < 0));
        const int sw = max(32 - (n_s >> 1), 33 - m);
        asm volatile("ds_write_b32 %[wa], %[wv]\n\t" WGNN_FLAT4_ASM
                     ::[pk] "v"(pk), [wa] "v"(wlane_addr), [wv] "v"(wv), [wb] "v"(wstrip_addr), [lb] "v"(lane16),
                       [mk] "v"(row_mask), [m] "s"(m), [sw] "s"(sw)
                     : WGNN_CLOB);
    };
    auto compute = [&](auto cur_set, int cs, int ce0, int buf_addr) {
        constexpr int CUR = decltype(cur_set)::value;
        if (cs >= ce0) return;
        consume(chunk_get<CUR>(), min(64, ce0 - cs), buf_addr);
        for (int s = cs + 64; s < ce0; s += 64) {            // rare: more than 64 entries for this wave in one block
            chunk_issue(cur_set, s, ce0);                   // (E(b+2) is in flight behind it: wait for both)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            consume(chunk_get<CUR>(), min(64, ce0 - s), buf_addr);
        }
    };
    // One source block, general form.  CUR / NXT = register sets of blocks b / b+2; (cs, ce0) = this block's segment;
    // (ns, ne) receive the segment of block b+2 (kept in SGPRs until that block is consumed).
    auto block = [&](auto cur_set, auto nxt_set, int b, int cs, int ce0, int& ns, int& ne) {
        if (b + 1 < nblk) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");    // all but E(b+1): DMA of block b, E(b), S(b+2)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (do_barrier) __builtin_amdgcn_s_barrier();       // everyone's DMA pieces landed; everyone is done with block b-1
        if (b + 2 < nblk) {
            asm volatile("v_readfirstlane_b32 %0, v32\n\tv_readfirstlane_b32 %1, v33" : "=s"(ns), "=s"(ne)::"memory");
            if (b + 3 < nblk) seg_load(seg + (b + 3) * kTW);
        }
        if (b + 1 < nblk && do_fill) fill(b + 1);
        if (b + 2 < nblk) chunk_issue(nxt_set, ns, ne);
        const bool hub = DBG && hub_left > 0;
        if (DBG) hub_left -= hub ? 1 : 0;
        if (do_comp && !hub) compute(cur_set, cs, ce0, (int)(size_t)smem + (b & 1) * buf_bytes);   // smem: the only LDS object
    };
    // Dedicated loader waves (round 3).  A property of the PLAN: the leading waves of a tile that own no destination rows
    // (graph.build_tile_plan(n_loaders=L) deals them none; a wave's row slots fill from slot 0, so "slot 0 empty" = "no
    // rows") issue the tile's WHOLE global->LDS stream in the steady-state blocks, the other waves only compute.  When
    // every wave owns rows (nld == 0) or the tile is empty (nld == 16) all 16 waves stream their share as before.
    int nld = 0;
    {
        cptr_t slot0 = (cptr_t)(t.tile_items + (size_t)tile * kTileRows);        // int4 items: .x of item i at dword 4*i
        bool leading = true;
        for (int w = 0; w < kTW; ++w) {
            const int x = slot0[w * kRPW * 4];
            leading = leading && x < 0;
            nld += leading ? 1 : 0;
        }
    }
    // Steady state (b + 3 < nblk, so blocks b+1, b+2, b+3 exist and block b+1 is a full one): no range checks, the
    // segment pointer / DMA source row / buffer parity advance incrementally.
    const int* segp = seg + 3 * kTW;                         // -> segment of block b+3
    int fill_row = cb + kKB;                                 // first source row of block b+1
    auto fast_block = [&](auto cur_set, auto nxt_set, int par, int cs, int ce0, int& ns, int& ne) {
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (do_barrier) __builtin_amdgcn_s_barrier();
        asm volatile("v_readfirstlane_b32 %0, v32\n\tv_readfirstlane_b32 %1, v33" : "=s"(ns), "=s"(ne)::"memory");
        seg_load(segp);
        segp += kTW;
        if (do_fill && (nld == 0 || wave < nld)) fill_rows(fill_row, kKB, par ^ 1, nld ? nld : kTW);
        fill_row += kKB;
        chunk_issue(nxt_set, ns, ne);
        const bool hub = DBG && hub_left > 0;
        if (DBG) hub_left -= hub ? 1 : 0;
        if (do_comp && !hub) compute(cur_set, cs, ce0, (int)(size_t)smem + par * buf_bytes);
    };

    if (nblk > 0) {
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
        cptr_t sseg = (cptr_t)seg;                        // the first two segments through the scalar cache
        int sA = sseg[0], eA = sseg[1], sB = 0, eB = 0, sC = 0, eC = 0;
        if (nblk > 1) { sB = sseg[kTW]; eB = sseg[kTW + 1]; }
        if (nblk > 2) seg_load(seg + 2 * kTW);
        if (do_fill) fill(0);
        chunk_issue(S0{}, sA, eA);
        if (nblk > 1) chunk_issue(S1{}, sB, eB);
        int b = 0, par = 0;
        for (; b + 5 < nblk; b += 3) {                       // three blocks per trip (the last one prefetches up to block b+5):
            fast_block(S0{}, S2{}, par, sA, eA, sC, eC);     // the register sets rotate statically, the buffer parity
            fast_block(S1{}, S0{}, par ^ 1, sB, eB, sA, eA); // is a scalar
            fast_block(S2{}, S1{}, par, sC, eC, sB, eB);
            par ^= 1;
        }
        for (; b < nblk; b += 3) {                           // the last 3..8 blocks: general form
            block(S0{}, S2{}, b, sA, eA, sC, eC);
            if (b + 1 < nblk) block(S1{}, S0{}, b + 1, sB, eB, sA, eA);
            if (b + 2 < nblk) block(S2{}, S1{}, b + 2, sC, eC, sB, eB);
        }
    }

    tile_epilogue<TOut, EPI, kRPW>(t, (size_t)tile * kTileRows + wave * kRPW, lane, [](int i) {
        float4 v;                                        // accumulator row i of this wave -> compiler registers
        asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\tv_mov_b32 %0, v64\n\tv_mov_b32 %1, v65\n\t"
                     "v_mov_b32 %2, v66\n\tv_mov_b32 %3, v67\n\ts_set_gpr_idx_off"
                     : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "s"(i * 4) : "m0");
        return v;
    });
}

// ---------------------------------------------------------------------------------------------
// agg_tiled_tall (round 5) - the same entry pipeline on a TALL tile: 8 waves x 256 VGPRs, 49 destination rows per wave,
// 392 per tile.  Why: scratch/pipe_bench (profiles/r05_pipe_bench.txt) prices an entry of the pipeline at 2.73 / 1.88 ns per
// CU (unshared / shared pair) with 16 waves and 2.86 / 1.94 ns with 8 - two waves per SIMD cost 5 %, not the 30 % of the
// round-1 pipeline - while the 16-wave kernel spends ~35 % of a cfg3 pass outside that steady state: per-(wave, block)
// prologues, warm-ups and drains over ~47 entries, one barrier per block, and TWO tiles per CU (512 tiles of 195 rows), each
// streaming the whole source table.  A 392-row tile covers cfg3's 100 000 rows in ONE round of 256 tiles: the table is
// streamed once per CU, a (wave, block) segment holds ~150 entries (3 chunks), and 49 rows per wave share LDS rows more often
// (79 % of the entries pair up, 55 % at 13 rows).
//   Register file: the compiler may allocate v[0:41] (amdgpu_num_vgpr(21)).  v[20:41] are the pipeline's staging / weight /
//   address registers: STATEMENT-LOCAL (dead between asm statements, named as clobbers), so the compiler may use them
//   between statements and in the epilogue but never across one (an asm operand is never given a clobbered register).
//   v[42:43] segment bounds, v[44:59] entry chunks (two block sets x four chunks x {meta, weight}) and the accumulators
//   v[60:255] carry state across statements and lie beyond the compiler's cap (build.audit enforces it).
//   Entry chunks: at the top of block b the (up to) four chunks of block b+1 are requested into the other set; one
//   `s_waitcnt vmcnt(0)` at the next top covers them, the DMA pieces of the block and the segment bounds of block b+2.
//   Every wave streams its contiguous share of a block (grouped immediate-offset form: no dedicated loader wave - with
//   8 waves there is none to spare, and ~10 pieces per wave and block are ~350 clk of ~7 k).
// ---------------------------------------------------------------------------------------------
constexpr int kTallTW = 8, kTallRPW = 49, kTallRows = kTallTW * kTallRPW, kTallNCH = 4;
#define WGNN_TALL_CLOB "m0", "memory", "scc", WGNN_TALL_VGPRS, WGNN_HAND_SGPRS

template <int SET, int J> __device__ __forceinline__ int2 tall_chunk_get() {
    int2 e;
#define WGNN_TCG(S_, J_, A_, B_) \
    if constexpr (SET == S_ && J == J_) asm volatile("v_mov_b32 %0, " A_ "\n\tv_mov_b32 %1, " B_ : "=v"(e.x), "=v"(e.y)::"memory");
    WGNN_TCG(0, 0, "v44", "v45") WGNN_TCG(0, 1, "v46", "v47") WGNN_TCG(0, 2, "v48", "v49") WGNN_TCG(0, 3, "v50", "v51")
    WGNN_TCG(1, 0, "v52", "v53") WGNN_TCG(1, 1, "v54", "v55") WGNN_TCG(1, 2, "v56", "v57") WGNN_TCG(1, 3, "v58", "v59")
#undef WGNN_TCG
    return e;
}
template <int SET, int J> __device__ __forceinline__ void tall_chunk_load(int off, const int2* base) {
#define WGNN_TCL(S_, J_, R_, A_, B_) \
    if constexpr (SET == S_ && J == J_) asm volatile("global_load_dwordx2 " R_ ", %0, %1" ::"v"(off), "s"(base) : "memory", A_, B_);
    WGNN_TCL(0, 0, "v[44:45]", "v44", "v45") WGNN_TCL(0, 1, "v[46:47]", "v46", "v47") WGNN_TCL(0, 2, "v[48:49]", "v48", "v49")
    WGNN_TCL(0, 3, "v[50:51]", "v50", "v51") WGNN_TCL(1, 0, "v[52:53]", "v52", "v53") WGNN_TCL(1, 1, "v[54:55]", "v54", "v55")
    WGNN_TCL(1, 2, "v[56:57]", "v56", "v57") WGNN_TCL(1, 3, "v[58:59]", "v58", "v59")
#undef WGNN_TCL
}

template <typename TOut, int EPI, bool DBG>
__global__ void __launch_bounds__(kTallTW * 64) __attribute__((amdgpu_num_vgpr(21)))
agg_tiled_tall(const KArgs a, const TArgs t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int row_bytes = flat_lds_row_bytes(a.D);
    const int g_row = a.D * (int)sizeof(float);
    const int n4 = a.D >> 2;
    const int kKB = t.kb, buf_bytes = kKB * row_bytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x;
    const int2 hdr = t.tile_hdr[tile];
    const int cb = __builtin_amdgcn_readfirstlane(hdr.x), ce = __builtin_amdgcn_readfirstlane(hdr.y);
    const int nblk = (ce - cb + kKB - 1) / kKB;
    const int* seg = t.seg_ptr + ((size_t)tile * t.nblk_max) * kTallTW + wave;     // seg[b*8], seg[b*8+1]
    const unsigned dbg = DBG ? a.flags : 0u;
    const bool do_fill = !(dbg & kDbgNoFill), do_comp = !(dbg & kDbgNoCompute), do_barrier = !(dbg & kDbgNoBarrier);

    for (int r = 0; r < kTallRPW; ++r)                   // zero the accumulators v[60:255]
        asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\t"
                     "v_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\ts_set_gpr_idx_off" ::"s"(r * 4) : WGNN_TALL_CLOB);

    const int lane16 = lane * 16, row_mask = 0x3FF00;
    // this wave's share of the global -> LDS DMA of `rows` source rows (see agg_tiled_flat4::fill_rows for the two forms)
    // `part` of `parts`: the wave's run of pieces may be issued in portions at its chunk boundaries instead of one burst at the
    // top of the block (contiguous form only; see kParts below for what was measured).
    auto fill_rows = [&](int r0, int rows, int buf, int part, int parts) {
        constexpr int nw = kTallTW;
        const bool contiguous = g_row == row_bytes;
        const int per = (rows + nw - 1) / nw;
        int first = contiguous ? wave * per : wave;
        int np = contiguous ? max(0, min(per, rows - first)) : (rows > wave ? (rows - wave + nw - 1) / nw : 0);
        if (contiguous) {                                     // this portion: pieces [part * q, (part + 1) * q) of the run, q = ceil(np / parts)
            const int q = (np + parts - 1) / parts, lo = min(np, part * q);
            np = min(np, lo + q) - lo;
            first += lo;
        } else if (part != 0) {
            np = 0;                                           // the general form goes out in one burst with portion 0
        }
        if (np == 0) return;
        const char* g = reinterpret_cast<const char*>(a.src) + ((size_t)r0 + first) * g_row;
        const int l = (int)(size_t)smem + buf * buf_bytes + first * row_bytes;
        int left = np;
        if (contiguous) {
#define WGNN_GROUP4(O1, O2, O3, STEP4, STEP1)                                                                                 \
            asm volatile("s_mov_b64 s[92:93], exec\n\ts_sub_u32 s91, 64, %[n4]\n\ts_lshr_b64 exec, s[92:93], s91\n\t"           \
                         "s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\t"                                                \
                         "s_cmp_lt_u32 %[left], 4\n\ts_cbranch_scc1 .Lw8_g1_%=\n\t"                                          \
                         ".Lw8_g4_%=:\n\t"                                                                                   \
                         "global_load_lds_dwordx4 %[vo], s[94:95]\n\t"                                                        \
                         "global_load_lds_dwordx4 %[vo], s[94:95] offset:" #O1 "\n\t"                                         \
                         "global_load_lds_dwordx4 %[vo], s[94:95] offset:" #O2 "\n\t"                                         \
                         "global_load_lds_dwordx4 %[vo], s[94:95] offset:" #O3 "\n\t"                                         \
                         "s_add_u32 m0, m0, " #STEP4 "\n\ts_add_u32 s94, s94, " #STEP4 "\n\ts_addc_u32 s95, s95, 0\n\t"        \
                         "s_sub_u32 %[left], %[left], 4\n\ts_cmp_ge_u32 %[left], 4\n\ts_cbranch_scc1 .Lw8_g4_%=\n\t"         \
                         "s_cmp_eq_u32 %[left], 0\n\ts_cbranch_scc1 .Lw8_ge_%=\n\t"                                          \
                         ".Lw8_g1_%=:\n\t"                                                                                   \
                         "global_load_lds_dwordx4 %[vo], s[94:95]\n\t"                                                        \
                         "s_add_u32 m0, m0, " #STEP1 "\n\ts_add_u32 s94, s94, " #STEP1 "\n\ts_addc_u32 s95, s95, 0\n\t"        \
                         "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lw8_g1_%=\n\t"         \
                         ".Lw8_ge_%=:\n\t"                                                                                   \
                         "s_mov_b64 exec, s[92:93]"                                                                          \
                         : [left] "+s"(left)                                                                                 \
                         : [g] "s"(g), [l] "s"(l), [vo] "v"(lane16), [n4] "s"(n4)                                            \
                         : "m0", "memory", "scc", "s91", "s92", "s93", "s94", "s95")
            if (row_bytes == 1024) { WGNN_GROUP4(1024, 2048, 3072, 4096, 1024); }
            else if (row_bytes == 512) { WGNN_GROUP4(512, 1024, 1536, 2048, 512); }
            else { WGNN_GROUP4(256, 512, 768, 1024, 256); }
#undef WGNN_GROUP4
            return;
        }
        // global row shorter than its LDS slot: one piece per address setting (v41: a statement-local running lane offset)
        asm volatile("s_mov_b64 s[92:93], exec\n\ts_sub_u32 s91, 64, %[n4]\n\ts_mul_i32 s90, %[n4], %[gs]\n\t"
                     "s_lshr_b64 exec, s[92:93], s91\n\t"
                     "s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\tv_mov_b32 v41, %[vo]\n\t"
                     ".Lw8_f1_%=:\n\t"
                     "global_load_lds_dwordx4 v41, s[94:95]\n\ts_add_u32 m0, m0, %[ms]\n\tv_add_u32 v41, s90, v41\n\t"
                     "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lw8_f1_%=\n\t"
                     "s_mov_b64 exec, s[92:93]"
                     : [left] "+s"(left)
                     : [g] "s"(g), [l] "s"(l), [vo] "v"(lane16), [n4] "s"(n4), [ms] "s"(nw * row_bytes), [gs] "s"(nw * 16)
                     : "m0", "memory", "scc", "s90", "s91", "s92", "s93", "s94", "s95", "v41");
    };
    // kParts = 1: one burst at the top of the block.  Portions at the chunk boundaries (kParts = 3) measured WORSE (cfg3
    // cells<-genes 1.236 vs 1.166 ms): a global_load_lds issued by a wave that also runs the entry pipeline costs ~150 clk, not the
    // ~34 clk of a wave that does nothing else - which is what the 16-wave kernel's dedicated loader wave is for.
    constexpr int kParts = 1;
    auto fill = [&](int b, int part, int parts) { fill_rows(cb + b * kKB, min(kKB, ce - (cb + b * kKB)), b & 1, part, parts); };
    // chunk j of segment [s, e) -> chunk register j of set SET, RIGHT-aligned (lane i <- entry s + 64 j + i - (64 - n));
    // nothing is requested for an empty chunk (the block's single vmcnt(0) needs no load count)
    auto chunk_issue = [&](auto set, auto jj, int s, int e) {
        constexpr int SET = decltype(set)::value, J = decltype(jj)::value;
        const int n = min(64, e - (s + 64 * J));
        if (n <= 0) return;
        tall_chunk_load<SET, J>(max(lane - (64 - n), 0) * 8, t.entries + s + 64 * J);
    };
    auto chunks_issue = [&](auto set, int s, int e) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        chunk_issue(set, I0{}, s, e); chunk_issue(set, I1{}, s, e); chunk_issue(set, I2{}, s, e); chunk_issue(set, I3{}, s, e);
    };
    auto seg_load = [&](const int* p) {                   // {begin, end} of one (block, wave) -> v[42:43]
        asm volatile("global_load_dwordx2 v[42:43], %0, %1" ::"v"(0), "s"(p) : "memory", "v42", "v43");
    };
    const int wstrip_addr = (int)(size_t)smem + 2 * buf_bytes + wave * 256;
    const int wlane_addr = wstrip_addr + lane * 4;
    auto consume = [&](const int2& ent, int n, int buf_addr) {
        // packed entry: LDS address of the source row (bits 8..17) | 4*slot (bits 0..7) | 4*slot of a shared pair's second
        // entry (bits 18..25) - slots are 6 bits wide here (49 rows per wave)
        const int pk = (buf_addr + (ent.x & 0xFF) * row_bytes) | (((ent.x >> 8) & 0x3F) << 2) | (((ent.x >> 16) & 0x3F) << 20);
        const bool mine = lane >= 64 - n;
        const int wv = mine ? ent.y : 0;
        const int m = (n + 1) >> 1;
        const int n_s = __popcll(__ballot(mine && ent.x < 0));
        const int sw = max(32 - (n_s >> 1), 33 - m);
        asm volatile("ds_write_b32 %[wa], %[wv]\n\t" WGNN_TALL_ASM
                     ::[pk] "v"(pk), [wa] "v"(wlane_addr), [wv] "v"(wv), [wb] "v"(wstrip_addr), [lb] "v"(lane16),
                       [mk] "v"(row_mask), [m] "s"(m), [sw] "s"(sw)
                     : WGNN_TALL_CLOB);
    };
    // `nb` >= 0: block nb's DMA portions 1 and 2 go out behind this block's first and second chunk (portion 0 went out at the top)
    auto compute = [&](auto set, int cs, int ce0, int buf_addr, int nb) {
        constexpr int SET = decltype(set)::value;
        if (cs + 0 < ce0) consume(tall_chunk_get<SET, 0>(), min(64, ce0 - cs), buf_addr);
        if (nb >= 0) fill(nb, 1, kParts);
        if (cs + 64 < ce0) consume(tall_chunk_get<SET, 1>(), min(64, ce0 - cs - 64), buf_addr);
        if (nb >= 0) fill(nb, 2, kParts);
        if (cs + 128 < ce0) consume(tall_chunk_get<SET, 2>(), min(64, ce0 - cs - 128), buf_addr);
        if (cs + 192 < ce0) consume(tall_chunk_get<SET, 3>(), min(64, ce0 - cs - 192), buf_addr);
        for (int s = cs + 64 * kTallNCH; s < ce0; s += 64) {   // rare: more than 256 entries for this wave in one block
            const int n = min(64, ce0 - s);
            tall_chunk_load<SET, 0>(max(lane - (64 - n), 0) * 8, t.entries + s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the next block's requests are in flight behind it: wait for all)
            consume(tall_chunk_get<SET, 0>(), n, buf_addr);
        }
    };
    // One source block.  SET = chunk set of block b; (cs, ce0) = this block's segment; (ns, ne) receive block b+1's.
    auto block = [&](auto set, auto other, int b, int cs, int ce0, int& ns, int& ne) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my DMA pieces of block b, my chunks of block b, the bounds of block b+1
        if (do_barrier) __builtin_amdgcn_s_barrier();       // everyone's pieces landed; everyone is done with block b-1
        if (b + 1 < nblk) {
            asm volatile("v_readfirstlane_b32 %0, v42\n\tv_readfirstlane_b32 %1, v43" : "=s"(ns), "=s"(ne)::"memory");
            if (b + 2 < nblk) seg_load(seg + (b + 2) * kTallTW);
            if (do_fill) fill(b + 1, 0, kParts);
            chunks_issue(other, ns, ne);
        }
        const int nb = (b + 1 < nblk && do_fill) ? b + 1 : -1;
        if (do_comp) compute(set, cs, ce0, (int)(size_t)smem + (b & 1) * buf_bytes, nb);
        else if (nb >= 0) { fill(nb, 1, kParts); fill(nb, 2, kParts); }
    };
    if (nblk > 0) {
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        cptr_t sseg = (cptr_t)seg;                        // the first segment through the scalar cache
        int sA = sseg[0], eA = sseg[1], sB = 0, eB = 0;
        if (nblk > 1) seg_load(seg + kTallTW);
        if (do_fill) fill(0, 0, 1);
        chunks_issue(S0{}, sA, eA);
        for (int b = 0; b < nblk; b += 2) {                   // two blocks per trip: the chunk sets swap roles statically
            block(S0{}, S1{}, b, sA, eA, sB, eB);
            if (b + 1 < nblk) block(S1{}, S0{}, b + 1, sB, eB, sA, eA);
        }
    }
    tile_epilogue<TOut, EPI, kTallRPW>(t, (size_t)tile * kTallRows + wave * kTallRPW, lane, [](int i) {
        float4 v;
        asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\tv_mov_b32 %0, v60\n\tv_mov_b32 %1, v61\n\t"
                     "v_mov_b32 %2, v62\n\tv_mov_b32 %3, v63\n\ts_set_gpr_idx_off"
                     : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w) : "s"(i * 4) : "m0");
        return v;
    });
}

// LDS bytes of one launch: the flat kernel's LDS rows are 256 / 512 / 1024 bytes (+ the per-wave weight strips)
inline bool use_flat(int D, unsigned flags) { return D <= 256 && !(flags & (1u << 19)); }   // bit 19: force the generic kernel (A/B)
inline long tiled_lds_bytes(int D, int block_rows, unsigned flags) {
    return use_flat(D, flags) ? 2L * block_rows * flat_lds_row_bytes(D) + kWStripBytes : 2L * block_rows * D * (long)sizeof(float);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: the "already raised to" mark is kept
// per device (the current one - the caller's stream must belong to it) in atomics, so that one process may drive several
// GPUs from several threads.  Two threads racing on the same device at worst both set the (idempotent) attribute.
constexpr int kMaxDevices = 64;
struct LdsMarks { std::atomic<int> v[kMaxDevices]; };

inline int raise_lds_limit(LdsMarks& marks, const void* fn, int lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return WGNN_ERR_LAUNCH;
    if (marks.v[dev].load(std::memory_order_acquire) >= lds) return WGNN_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return WGNN_ERR_LAUNCH;
    int seen = marks.v[dev].load(std::memory_order_relaxed);
    while (seen < lds && !marks.v[dev].compare_exchange_weak(seen, lds, std::memory_order_release)) {}
    return WGNN_OK;
}

template <typename TOut, int EPI>
int launch_tiled(const KArgs& a, const TArgs& t, long n_tiles, hipStream_t st) {
    const int lds = (int)tiled_lds_bytes(a.D, t.kb, a.flags);
    if (lds > 160 * 1024) return WGNN_ERR_PLAN;
    static LdsMarks generic_marks{}, flat_marks{}, flat_dbg_marks{}, tall_marks{}, tall_dbg_marks{};      // per instantiation, per device
    if (t.tall) {
        if (!use_flat(a.D, a.flags)) return WGNN_ERR_PLAN;          // a tall plan has no generic-kernel form
        if (a.flags & 0xFFFF0000u) {                                // timing-experiment switches: the instantiation that reads them
            if (int rc = raise_lds_limit(tall_dbg_marks, reinterpret_cast<const void*>(&agg_tiled_tall<TOut, EPI, true>), lds)) return rc;
            hipLaunchKernelGGL((agg_tiled_tall<TOut, EPI, true>), dim3((unsigned)n_tiles), dim3(kTallTW * 64), lds, st, a, t);
        } else {
            if (int rc = raise_lds_limit(tall_marks, reinterpret_cast<const void*>(&agg_tiled_tall<TOut, EPI, false>), lds)) return rc;
            hipLaunchKernelGGL((agg_tiled_tall<TOut, EPI, false>), dim3((unsigned)n_tiles), dim3(kTallTW * 64), lds, st, a, t);
        }
    } else if (use_flat(a.D, a.flags)) {
        if (int rc = raise_lds_limit(flat_marks, reinterpret_cast<const void*>(&agg_tiled_flat4<TOut, EPI, false>), lds)) return rc;
        if (a.flags & 0xFFFF0000u)
            if (int rc = raise_lds_limit(flat_dbg_marks, reinterpret_cast<const void*>(&agg_tiled_flat4<TOut, EPI, true>), lds)) return rc;
        if (a.flags & 0xFFFF0000u)                       // timing-experiment switches: the instantiation that reads them
            hipLaunchKernelGGL((agg_tiled_flat4<TOut, EPI, true>), dim3((unsigned)n_tiles), dim3(kTW * 64), lds, st, a, t);
        else
            hipLaunchKernelGGL((agg_tiled_flat4<TOut, EPI, false>), dim3((unsigned)n_tiles), dim3(kTW * 64), lds, st, a, t);
    } else {
        if (int rc = raise_lds_limit(generic_marks, reinterpret_cast<const void*>(&agg_tiled<TOut, EPI>), lds)) return rc;
        hipLaunchKernelGGL((agg_tiled<TOut, EPI>), dim3((unsigned)n_tiles), dim3(kTW * 64), lds, st, a, t);
    }
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

// `block_rows` of the tiled entry points carries the plan's geometry in its high half (include/wgnn.h: WGNN_PLAN_TALL)
inline int plan_rows(int32_t block_rows) { return block_rows & 0xFFFF; }
inline int plan_tall(int32_t block_rows) { return (block_rows & WGNN_PLAN_TALL) ? 1 : 0; }

}  // namespace

namespace wgnn {
int launch_finalize_f32(const KArgs& a, int epi, hipStream_t st);     // defined in wgnn_kernels.hip
}

extern "C" int wgnn_agg_fwd_tiled(const void* rowptr, const float* alpha, int alpha_mode, int32_t self_idx,
                                  const float* h_src, int64_t n_src, float* src_scratch,
                                  const float* h_self, int64_t ld_self,
                                  const int32_t* row_ids, const float* inv_deg, const float* bias,
                                  float* out, int64_t ld_out, float* neigh_sum, int64_t n_out, int32_t D, uint32_t flags,
                                  const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max, int32_t block_rows,
                                  const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                                  const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                                  void* stream) {
    if (!h_src || !out || n_out < 0 || n_tiles < 0 || n_src < 0 || nblk_max < 0) return WGNN_ERR_BAD_ARG;
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (!inv_deg && !rowptr && !(flags & WGNN_FLAG_NO_MEAN)) return WGNN_ERR_BAD_ARG;
    if (D <= 0 || D % 4 || ld_out % 4 || (h_self && ld_self % 4)) return WGNN_ERR_ALIGNMENT;
    if (D > 256) return WGNN_ERR_UNSUPPORTED;                     // one float4 per lane
    const int tall = plan_tall(block_rows);
    if (block_rows & ~(0xFFFF | WGNN_PLAN_TALL)) return WGNN_ERR_PLAN;
    block_rows = plan_rows(block_rows);
    if (block_rows < 16 || block_rows > 255 || tiled_lds_bytes(D, block_rows, flags) > 160 * 1024) return WGNN_ERR_PLAN;
    if (!aligned16(h_src) || !aligned16(out) || (h_self && !aligned16(h_self)) || (bias && !aligned16(bias)))
        return WGNN_ERR_ALIGNMENT;
    if (n_tiles > 0 && (!tile_items || !tile_hdr || !entries || !seg_ptr)) return WGNN_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !partials || n_partials <= 0)) return WGNN_ERR_WORKSPACE;
    const bool fold_alpha = alpha_mode == WGNN_SRC_IS_GENE && !(flags & WGNN_FLAG_SRC_PRESCALED);
    if (fold_alpha && (!src_scratch || !aligned16(src_scratch))) return WGNN_ERR_WORKSPACE;
    if (n_out == 0 || n_tiles == 0) return WGNN_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* src = h_src;
    if (fold_alpha) {                                             // (h*alpha) once per source row, gnn.py:54
        const long n4 = (long)n_src * (D / 4);
        hipLaunchKernelGGL(scale_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, h_src, alpha, src_scratch,
                           (long)n_src, D / 4);
        src = src_scratch;
    }
    KArgs a{};
    a.rowptr = rowptr;
    a.src = src; a.ld_src = D; a.alpha = alpha; a.mode = alpha_mode; a.self_idx = self_idx;
    a.self = h_self; a.ld_self = ld_self; a.row_ids = row_ids; a.inv_deg = inv_deg; a.bias = bias;
    a.out = out; a.ld_out = ld_out; a.D = D; a.flags = flags; a.aux1 = neigh_sum;
    if (neigh_sum && !aligned16(neigh_sum)) return WGNN_ERR_ALIGNMENT;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    TArgs t{reinterpret_cast<const int2*>(entries), seg_ptr, reinterpret_cast<const int4*>(tile_items),
            reinterpret_cast<const int2*>(tile_hdr), nblk_max, block_rows, tall};
    int rc = launch_tiled<float, EPI_FWD>(a, t, n_tiles, st);
    if (rc) return rc;
    if (n_long > 0) return launch_finalize_f32(a, EPI_FWD, st);
    return WGNN_OK;
}

// ---- tiled backward: K2 over the transposed structure, K3 on the forward structure -------------------------
static int tiled_common_check(int32_t D, int32_t block_rows, const void* entries, const void* seg_ptr, const void* tile_items,
                              const void* tile_hdr, int64_t n_tiles, const void* long_rows, int64_t n_long,
                              const float* partials, int64_t n_partials) {
    if (D <= 0 || D % 4) return WGNN_ERR_ALIGNMENT;
    if (D > 256) return WGNN_ERR_UNSUPPORTED;
    if (block_rows & ~(0xFFFF | WGNN_PLAN_TALL)) return WGNN_ERR_PLAN;
    block_rows = plan_rows(block_rows);
    if (block_rows < 16 || block_rows > 255 || tiled_lds_bytes(D, block_rows, 0) > 160 * 1024) return WGNN_ERR_PLAN;
    if (n_tiles < 0 || (n_tiles > 0 && (!tile_items || !tile_hdr || !entries || !seg_ptr))) return WGNN_ERR_BAD_ARG;
    if (n_long > 0 && (!long_rows || !partials || n_partials <= 0)) return WGNN_ERR_WORKSPACE;
    return WGNN_OK;
}

extern "C" int wgnn_agg_bwd_src_tiled(const float* alpha, int alpha_mode, const float* col_scale,
                                      const float* g, int64_t n_dst, float* g_scratch,
                                      const float* h_src, int64_t ld_src, float* dh_src, int64_t ld_dh, float* dalpha,
                                      int accumulate, int64_t n_src, int32_t D,
                                      const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max, int32_t block_rows,
                                      const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                                      const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                                      void* stream) {
    int rc = tiled_common_check(D, block_rows, entries, seg_ptr, tile_items, tile_hdr, n_tiles, long_rows, n_long, partials, n_partials);
    if (rc) return rc;
    if (!g || !dh_src || n_src < 0 || n_dst < 0) return WGNN_ERR_BAD_ARG;
    if (col_scale && !g_scratch) return WGNN_ERR_WORKSPACE;      // col_scale == NULL: `g` already carries the per-row factors
    if (alpha_mode < WGNN_SRC_IS_GENE || alpha_mode > WGNN_NO_ALPHA) return WGNN_ERR_BAD_ARG;
    if (alpha_mode != WGNN_NO_ALPHA && !alpha) return WGNN_ERR_BAD_ARG;
    if (ld_dh % 4 || (h_src && ld_src % 4)) return WGNN_ERR_ALIGNMENT;
    if (!aligned16(g) || (g_scratch && !aligned16(g_scratch)) || !aligned16(dh_src) || (h_src && !aligned16(h_src))) return WGNN_ERR_ALIGNMENT;
    if (n_src == 0 || n_tiles == 0) return WGNN_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // per-destination factors (inv_deg[r], x alpha[r] for cell->gene edges) are folded into the gradient rows once
    // (wgnn_agg_bwd_prepare writes the gradient rows already multiplied: then col_scale is NULL and `g` is the source table)
    if (col_scale) {
        const long n4 = (long)n_dst * (D / 4);
        hipLaunchKernelGGL(scale_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, g, col_scale, g_scratch, (long)n_dst, D / 4);
    }
    KArgs a{};
    a.src = col_scale ? g_scratch : g; a.ld_src = D; a.alpha = alpha; a.mode = alpha_mode;
    a.self = h_src; a.ld_self = ld_src; a.out = dh_src; a.ld_out = ld_dh; a.aux1 = dalpha;
    a.D = D; a.flags = WGNN_FLAG_NO_MEAN; a.accumulate = accumulate;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    TArgs t{reinterpret_cast<const int2*>(entries), seg_ptr, reinterpret_cast<const int4*>(tile_items),
            reinterpret_cast<const int2*>(tile_hdr), nblk_max, plan_rows(block_rows), plan_tall(block_rows)};
    rc = launch_tiled<float, EPI_BWD_SRC>(a, t, n_tiles, st);
    if (rc) return rc;
    if (n_long > 0) return launch_finalize_f32(a, EPI_BWD_SRC, st);
    return WGNN_OK;
}

extern "C" int wgnn_agg_bwd_alpha_tiled(const float* inv_deg, const float* g, int64_t ld_g,
                                        const float* h_src, const float* h_self, int64_t ld_self,
                                        float* dalpha_row, float* dself_row, int64_t n_out, int32_t D,
                                        const int32_t* entries, const int32_t* seg_ptr, int32_t nblk_max, int32_t block_rows,
                                        const int32_t* tile_items, const int32_t* tile_hdr, int64_t n_tiles,
                                        const int32_t* long_rows, int64_t n_long, float* partials, int64_t n_partials,
                                        void* stream) {
    int rc = tiled_common_check(D, block_rows, entries, seg_ptr, tile_items, tile_hdr, n_tiles, long_rows, n_long, partials, n_partials);
    if (rc) return rc;
    if (!g || !h_src || !inv_deg || n_out < 0) return WGNN_ERR_BAD_ARG;
    if (ld_g % 4 || (h_self && ld_self % 4)) return WGNN_ERR_ALIGNMENT;
    if (!aligned16(g) || !aligned16(h_src) || (h_self && !aligned16(h_self))) return WGNN_ERR_ALIGNMENT;
    if (n_out == 0 || n_tiles == 0) return WGNN_OK;
    KArgs a{};
    a.src = h_src; a.ld_src = D; a.mode = WGNN_NO_ALPHA;
    a.self = h_self; a.ld_self = ld_self; a.inv_deg = inv_deg;
    a.g = g; a.ld_g = ld_g; a.aux1 = dalpha_row; a.aux2 = dself_row;
    a.D = D; a.flags = 0;
    a.long_rows = reinterpret_cast<const int4*>(long_rows); a.n_long = n_long; a.partials = partials;
    TArgs t{reinterpret_cast<const int2*>(entries), seg_ptr, reinterpret_cast<const int4*>(tile_items),
            reinterpret_cast<const int2*>(tile_hdr), nblk_max, plan_rows(block_rows), plan_tall(block_rows)};
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = launch_tiled<float, EPI_BWD_ALPHA>(a, t, n_tiles, st);
    if (rc) return rc;
    if (n_long > 0) return launch_finalize_f32(a, EPI_BWD_ALPHA, st);
    return WGNN_OK;
}

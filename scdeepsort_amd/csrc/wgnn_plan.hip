// wgnn_plan.hip - device-side construction of the tile plan's `entries` (round 5; rebuilt in round 6).
//
// The tile plan re-orders the CSR into (tile, LDS block, wave) segments with the layout [unshared][pad][shared pairs]
// (include/wgnn.h, wgnn_agg_fwd_tiled).  Rounds 1-4 built it with framework index arithmetic: two stable sorts and ~40
// element-wise passes over 8e7 non-zeros, ~90 ms per direction at BASELINE cfg3 (18-22 ms after round 5's trimming) - more
// than ten forwards, paid by every one-shot inference (the reference builds a graph per prediction, predict.py:44-54).
// The structure makes a sort unnecessary: a (tile, wave) pair owns <= 64 destination rows whose non-zeros are already
// sorted by column, so ONE wavefront per (tile, wave, block range) - lane = destination slot - can place them.  Round 5 walked
// the rows in lock step, one source-row group at a time (a wave minimum of the pending columns, a ballot, running counts), twice:
// 9.8 / 12.4 ms per plan.  Round 6:
//   COUNT - a segment is padded to an EVEN number of entries whenever its entry count is odd (the pair count is even, so "odd
//           unshared run" == "odd total": the pad costs no pair step, an odd chunk has an idle half step anyway), so a segment's
//           size is a function of its entry count alone: an LDS histogram of the wave's rows' columns over the tile's blocks,
//           filled entry-parallel from coalesced reads (tile_plan_count).
//   FILL  - per LDS block: a bitmap of slots per source row (ds_or), one wave scan over the source rows for the groups' offsets,
//           entry-parallel placement from a ballot-appended entry list (tile_plan_fill).
// 3.2 / 4.1 ms per plan.  No atomics on global memory, deterministic.  Reference counterpart: none (DGL built its own CSR,
// preprocess_internal.py:215).
#include <climits>
#include "wgnn_common.h"

namespace {
using namespace wgnn;

constexpr int kPlanWavesPerBlock = 4;
constexpr int kPairFlag = (int)0x80000000u, kPadFlag = 1 << 30;

struct PlanArgs {
    const int* rowptr; const int* col; const float* val;
    const int* slot_vrow;                       // [n_row_tiles * waves * rpw]  virtual row of (row tile, wave, slot) | -1
    const int* vrow_row; const int* vrow_part; const int* vrow_k;   // per virtual row: CSR row, part j of k (non-zeros j, j+k, ..)
    const int* flat_t;                          // [n_flat] row tile of the tile launched at position f
    const int2* tile_hdr;                       // [n_flat] {col_begin, col_end}
    int n_flat, waves, rpw, nblk_max, kb;
    int nsplit;                                 // wavefronts per (tile, wave): each walks 1/nsplit of the tile's blocks
    int* seg_total;                             // [n_flat * nblk_max * waves]  COUNT: written
    const int* seg_ptr; int2* entries;          // FILL: [n_seg + 1] offsets (prefix sum of seg_total rounded up to even), output
};

// Exclusive scan over the 64 lanes on the VALU's data-parallel primitives - a row-wise inclusive scan by row shifts, the rows
// chained through the row broadcasts of gfx9 (~8 instructions, no LDS traffic; the shuffle form - six ds_bpermute per call -
// made the LDS pipe the bottleneck of round 5's first walk); returns the exclusive prefix of `v`, `total` = the wave's sum.
__device__ __forceinline__ int wave_excl_scan(int v, int& total) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);     // row_shr:8   inclusive inside a row of 16
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3 += total of rows 0, 2
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3 += total of rows 0..1
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

struct RowCursor {                     // one lane's (virtual) row: non-zeros base + i * stride, i < n
    long base; int stride, n;
};

typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));       // four consecutive columns, dword-aligned
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// Entries i .. i+3 of a lane's row (INT_MAX behind its end).  A lane that fetches one column per load makes the walk
// bound by line accesses (16 scattered lines per wave instruction, one per 4 bytes used: the round-5 count pass ran at 0.34 TB/s);
// contiguous rows (stride 1: every row but the virtual parts of hub rows) take four columns per load.
__device__ __forceinline__ int4u load_cols(const int* __restrict__ col, const RowCursor& rc, int i) {
    if (rc.stride == 1 && i + 4 <= rc.n) return *reinterpret_cast<const int4u*>(col + rc.base + i);
    int4u q;
    q.x = i < rc.n ? col[rc.base + (long)i * rc.stride] : INT_MAX;
    q.y = i + 1 < rc.n ? col[rc.base + (long)(i + 1) * rc.stride] : INT_MAX;
    q.z = i + 2 < rc.n ? col[rc.base + (long)(i + 2) * rc.stride] : INT_MAX;
    q.w = i + 3 < rc.n ? col[rc.base + (long)(i + 3) * rc.stride] : INT_MAX;
    return q;
}

// The lane's position in its row: `c.x` = the next column (entry j), the rest of its quad behind it, and the NEXT quad already
// requested (one load per four entries, issued four entries ahead of its use).
struct ColCursor {
    int4u c, nx;
    int j, k;                                    // next entry; entries of the current quad already consumed
    __device__ __forceinline__ void init(const int* col, const RowCursor& rc, int j0) {
        j = j0; k = 0;
        c = load_cols(col, rc, j0);
        nx = load_cols(col, rc, j0 + 4);
    }
    __device__ __forceinline__ int front() const { return c.x; }
    __device__ __forceinline__ void pop(const int* col, const RowCursor& rc) {
        ++j;
        if (++k == 4) {
            c = nx; k = 0;
            nx = load_cols(col, rc, j + 4);
        } else {
            c.x = c.y; c.y = c.z; c.z = c.w; c.w = INT_MAX;
        }
    }
};

__device__ __forceinline__ void plan_preamble(const PlanArgs& a, int gw0, int lane, int& f, int& w, int& cb, int& ce,
                                              RowCursor& rc, int& b_begin, int& b_end, int& j) {
    const int gw = gw0 / a.nsplit, piece = gw0 - gw * a.nsplit;
    f = gw / a.waves; w = gw - f * a.waves;
    const int2 hdr = a.tile_hdr[f];
    cb = hdr.x; ce = hdr.y;
    const int v = lane < a.rpw ? a.slot_vrow[((size_t)a.flat_t[f] * a.waves + w) * a.rpw + lane] : -1;
    rc.base = 0; rc.stride = 1; rc.n = 0;
    if (v >= 0) {
        const int row = a.vrow_row[v], part = a.vrow_part[v], k = a.vrow_k[v];
        const int rbeg = a.rowptr[row], rlen = a.rowptr[row + 1] - rbeg;
        rc.n = rlen / k + (part < rlen % k ? 1 : 0);
        rc.base = (long)rbeg + part;
        rc.stride = k;
    }
    const int nblk = (ce - cb + a.kb - 1) / a.kb;
    const int per = (nblk + a.nsplit - 1) / a.nsplit;                        // my range of blocks
    b_begin = min(nblk, piece * per); b_end = min(nblk, b_begin + per);
    const int first_col = cb + b_begin * a.kb;
    j = 0;                                       // my next non-zero: the first one at or behind my first block
    if (first_col > 0) {
        int lo = 0, hi = rc.n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.col[rc.base + (long)mid * rc.stride] < first_col) lo = mid + 1; else hi = mid;
        }
        j = lo;
    }
}

// COUNT: seg_total[s] = entries of segment s = a histogram of the wave's rows' columns over the tile's blocks.  ENTRY-parallel:
// the 64 lanes read one row's columns 64 at a time (coalesced) and add into the wave's LDS histogram; the part of a row inside
// the wave's column range is found by two binary searches per row (lane = slot).  ~8 instructions per 64 entries - the
// lane-per-row walk of the first round-6 form spent ~215 per (wave, block) on its divergent steps (0.96 / 1.43 ms at cfg3).
constexpr int kCountMaxBlocks = 1024;           // blocks per wavefront (4 KiB of LDS per wave); more blocks -> more pieces
__global__ void __launch_bounds__(kPlanWavesPerBlock * 64) tile_plan_count(const PlanArgs a) {
    __shared__ int s_hist[kPlanWavesPerBlock][kCountMaxBlocks];
    const int wib = (int)(threadIdx.x >> 6);
    const int gw0 = blockIdx.x * kPlanWavesPerBlock + wib;                          // one wave per (tile, wave slot, block range)
    if (gw0 >= a.n_flat * a.waves * a.nsplit) return;
    const int lane = threadIdx.x & 63;
    const int gw = gw0 / a.nsplit, piece = gw0 - gw * a.nsplit;
    const int f = gw / a.waves, w = gw - f * a.waves;
    const int2 hdr = a.tile_hdr[f];
    const int cb = hdr.x, ce = hdr.y;
    const int nblk = (ce - cb + a.kb - 1) / a.kb;
    const int per = (nblk + a.nsplit - 1) / a.nsplit;
    const int b_begin = min(nblk, piece * per), b_end = min(nblk, b_begin + per), nb = b_end - b_begin;
    if (nb <= 0) return;
    const int c_lo = cb + b_begin * a.kb, c_hi = min(ce, cb + b_end * a.kb);
    int* hist = s_hist[wib];
    for (int i = lane; i < nb; i += 64) hist[i] = 0;
    // my slot's row and its entries inside [c_lo, c_hi)
    const int v = lane < a.rpw ? a.slot_vrow[((size_t)a.flat_t[f] * a.waves + w) * a.rpw + lane] : -1;
    long base = 0;
    int stride = 1, i_lo = 0, i_hi = 0;
    if (v >= 0) {
        const int row = a.vrow_row[v], part = a.vrow_part[v], k = a.vrow_k[v];
        const int rbeg = a.rowptr[row], rlen = a.rowptr[row + 1] - rbeg;
        const int n = rlen / k + (part < rlen % k ? 1 : 0);
        base = (long)rbeg + part; stride = k;
        auto lower = [&](int key) {                                               // first entry with column >= key
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a.col[base + (long)mid * stride] < key) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        i_lo = lower(c_lo); i_hi = lower(c_hi);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float inv_kb = 1.0f / (float)a.kb;          // (x + 0.5) / kb never sits within rounding of an integer: exact floor for x < 2^23
    for (int r = 0; r < a.rpw; ++r) {
        const int lo = __builtin_amdgcn_readlane(i_lo, r), hi = __builtin_amdgcn_readlane(i_hi, r);
        if (lo >= hi) continue;
        const int st = __builtin_amdgcn_readlane(stride, r);
        const long bs = ((long)__builtin_amdgcn_readlane((int)(base >> 32), r) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)(base & 0xFFFFFFFFl), r);
        for (int i = lo + lane; i < hi; i += 64) {
            const int c = a.col[bs + (long)i * st];
            atomicAdd(&hist[(int)(((float)(c - c_lo) + 0.5f) * inv_kb)], 1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nb; i += 64) a.seg_total[((size_t)f * a.nblk_max + b_begin + i) * a.waves + w] = hist[i];
}

// FILL (round 6): entries[seg_ptr[s] .. seg_ptr[s + 1]) = [unshared, by source row][pad iff the count is odd][shared pairs, by
// source row] without walking the groups one by one.  Per block the wave keeps, in its private LDS strip, a BITMAP per source
// row of the block: bit l of bm[r] = "the row in slot l reads source row r".  (1) every lane ORs its entries in (its own few
// steps), (2) the wave prefix-sums, over the <= 255 source rows, the group sizes split into their paired (even) part and the
// odd one out - four source rows per lane, one wave scan - which gives every group its offset in the unshared run and in
// the pair run, (3) every lane revisits its entries: the bitmap word tells an entry its rank in the group (popcount of the
// lower slots), whether it pairs up, and the slot of its pair's second member; its position follows from the group's offset.
// ~250 instructions per (wave, block) segment of ~45 entries where the round-5 group walk took ~1200 (a wave minimum, a
// ballot and the bookkeeping for each of the ~30 groups).  LDS operations of one wave execute in order; only this wave touches
// its strip, so wave-level fences order the phases.  Same layout as the group walk produced.
// The revisit of step (3) is ENTRY-parallel when the segment fits the wave's entry list (<= 256 entries: every flat-geometry
// segment in practice): step (1) also appends {source row, slot, CSR position} of every entry to a list in the strip - the lanes
// that still have an entry in the block take consecutive list slots, from a ballot - and step (3) gives each lane one list entry
// (one trip for a segment of <= 64 entries) instead of each lane its row's entries (as many trips as the busiest row has).
constexpr int kPlanMaxKb = 256, kPlanListCap = 256;
__global__ void __launch_bounds__(kPlanWavesPerBlock * 64) tile_plan_fill(const PlanArgs a) {
    __shared__ unsigned long long s_bm[kPlanWavesPerBlock][kPlanMaxKb];
    __shared__ int s_base[kPlanWavesPerBlock][kPlanMaxKb];
    __shared__ int2 s_list[kPlanWavesPerBlock][kPlanListCap];
    const int wib = (int)(threadIdx.x >> 6);
    const int gw0 = blockIdx.x * kPlanWavesPerBlock + wib;
    if (gw0 >= a.n_flat * a.waves * a.nsplit) return;
    const int lane = threadIdx.x & 63;
    int f, w, cb, ce, b_begin, b_end, j;
    RowCursor rc;
    plan_preamble(a, gw0, lane, f, w, cb, ce, rc, b_begin, b_end, j);
    unsigned long long* bm = s_bm[wib];
    int* gbase = s_base[wib];
    int2* list = s_list[wib];
    const unsigned long long my_bit = 1ull << lane, below = my_bit - 1ull;
    ColCursor cur;
    cur.init(a.col, rc, j);
    for (int b = b_begin; b < b_end; ++b) {
        const int lo_col = cb + b * a.kb, hi_col = min(ce, lo_col + a.kb);
        const size_t seg = ((size_t)f * a.nblk_max + b) * a.waves + w;
        const int p0 = a.seg_ptr[seg], tot = a.seg_total[seg];
        if (tot == 0) continue;                                              // (uniform; nobody has an entry in this block)
#pragma unroll
        for (int k = 0; k < kPlanMaxKb / 64; ++k) bm[k * 64 + lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (1) my entries of this block -> the bitmap
        const int j0 = cur.j;
        const bool listed = tot <= kPlanListCap;                             // (uniform)
        if (listed) {
            int run = 0;
            for (;;) {                                                       // lock step: one entry per lane that still has one
                const bool have = cur.front() < hi_col;
                const unsigned long long act = __ballot(have);
                if (act == 0ull) break;
                if (have) {
                    const int s = cur.front() - lo_col;
                    atomicOr(&bm[s], my_bit);
                    list[run + __popcll(act & below)] = make_int2((s << 8) | lane, (int)(rc.base + (long)cur.j * rc.stride));
                    cur.pop(a.col, rc);
                }
                run += __popcll(act);
            }
        } else {
            while (cur.front() < hi_col) {
                atomicOr(&bm[cur.front() - lo_col], my_bit);
                cur.pop(a.col, rc);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (2) group offsets: unshared run in the low half, pair run in the high half of one packed sum (a segment holds
        //     < 2^15 entries of either kind: <= 64 slots x 255 source rows)
        int loc[kPlanMaxKb / 64], mine = 0;
#pragma unroll
        for (int k = 0; k < kPlanMaxKb / 64; ++k) {
            const int g = __popcll(bm[lane * (kPlanMaxKb / 64) + k]);
            loc[k] = mine;
            mine += (g & 1) | ((g & ~1) << 16);
        }
        int total;
        const int before = wave_excl_scan(mine, total);
#pragma unroll
        for (int k = 0; k < kPlanMaxKb / 64; ++k) gbase[lane * (kPlanMaxKb / 64) + k] = before + loc[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n_un = total & 0xFFFF;
        const int sh0 = p0 + n_un + (tot & 1);                               // the pair run starts at an even offset
        // (3) place my entries
        auto place = [&](int s, int slot, int wbits) {
            const unsigned long long M = bm[s];
            const int gb = gbase[s];
            const int g = __popcll(M), r = __popcll(M & ((1ull << slot) - 1ull)), p = g & ~1;
            int meta = (slot << 8) | s;
            int pos;
            if (r < p) {
                meta |= kPairFlag;
                if ((r & 1) == 0) meta |= (slot + 1 + __builtin_ctzll(M >> (slot + 1))) << 16;       // the second's slot
                pos = sh0 + (gb >> 16) + r;
            } else {
                pos = p0 + (gb & 0xFFFF);
                if ((tot & 1) && pos == p0 + n_un - 1)                       // the last unshared entry: its zero-weight copy pads
                    a.entries[pos + 1] = make_int2((meta & 0xFFFF) | kPadFlag, 0);
            }
            a.entries[pos] = make_int2(meta, wbits);
        };
        const int j1 = cur.j;
        if (listed) {
            for (int i = lane; i < tot; i += 64) {
                const int2 it = list[i];
                place(it.x >> 8, it.x & 63, __float_as_int(a.val[it.y]));
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        for (int e = j0; e < j1; e += 4) {                                   // (my entries again: cache-hot, four per load)
            if (rc.stride == 1 && e + 4 <= rc.n) {
                const int4u cq = *reinterpret_cast<const int4u*>(a.col + rc.base + e);
                const float4u vq = *reinterpret_cast<const float4u*>(a.val + rc.base + e);
                place(cq.x - lo_col, lane, __float_as_int(vq.x));
                if (e + 1 < j1) place(cq.y - lo_col, lane, __float_as_int(vq.y));
                if (e + 2 < j1) place(cq.z - lo_col, lane, __float_as_int(vq.z));
                if (e + 3 < j1) place(cq.w - lo_col, lane, __float_as_int(vq.w));
            } else {
                for (int t = e; t < min(e + 4, j1); ++t) {
                    const long at = rc.base + (long)t * rc.stride;
                    place(a.col[at] - lo_col, lane, __float_as_int(a.val[at]));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                                     // (the next block clears the strip)
    }
}

// wavefronts per (tile, wave): enough of them to fill the chip's 8192 wave slots twice (a few-row operand has few tiles)
int pick_nsplit(long n_tile_waves, int nblk_max) {
    long want = (2 * 8192 + n_tile_waves - 1) / (n_tile_waves > 0 ? n_tile_waves : 1);
    if (want < 1) want = 1;
    if (want > 16) want = 16;
    if (want > nblk_max) want = nblk_max;
    if (want < 1) want = 1;                      // (nblk_max < 1 is rejected by check(); never divide by zero before it)
    return (int)want;
}

int check(const PlanArgs& a) {
    if (!a.rowptr || !a.col || !a.slot_vrow || !a.vrow_row || !a.vrow_part || !a.vrow_k || !a.flat_t || !a.tile_hdr ||
        !a.seg_total)
        return WGNN_ERR_BAD_ARG;
    const bool geom_ok = (a.waves == 16 && a.rpw == 16) || (a.waves == 8 && a.rpw == 49);   // the two plan geometries (6-bit slots)
    if (a.n_flat < 0 || !geom_ok || a.nblk_max < 1 || a.kb < 16 || a.kb > 255 || a.nsplit < 1)
        return WGNN_ERR_BAD_ARG;
    return WGNN_OK;
}

}  // namespace

extern "C" int wgnn_tile_plan_count(const int32_t* rowptr, const int32_t* col, const int32_t* slot_vrow,
                                    const int32_t* vrow_row, const int32_t* vrow_part, const int32_t* vrow_k,
                                    const int32_t* flat_t, const int32_t* tile_hdr, int64_t n_flat, int32_t waves, int32_t rpw,
                                    int32_t nblk_max, int32_t block_rows, int32_t* seg_total, int32_t* seg_pairs, void* stream) {
    (void)seg_pairs;                                   // (0.2.4 wrote the paired entries per segment here; unused since 0.2.5)
    if (n_flat < 0 || n_flat > INT_MAX / 256) return n_flat < 0 ? WGNN_ERR_BAD_ARG : WGNN_ERR_UNSUPPORTED;
    PlanArgs a{rowptr, col, nullptr, slot_vrow, vrow_row, vrow_part, vrow_k, flat_t, reinterpret_cast<const int2*>(tile_hdr),
               (int)n_flat, waves, rpw, nblk_max, block_rows, 1, seg_total, nullptr, nullptr};
    a.nsplit = pick_nsplit((long)n_flat * waves, nblk_max);
    if ((nblk_max + a.nsplit - 1) / a.nsplit > kCountMaxBlocks) a.nsplit = (nblk_max + kCountMaxBlocks - 1) / kCountMaxBlocks;
    if (int rc = check(a)) return rc;
    if (n_flat == 0) return WGNN_OK;
    const long n_waves = (long)n_flat * waves * a.nsplit;
    hipLaunchKernelGGL(tile_plan_count, dim3((unsigned)((n_waves + kPlanWavesPerBlock - 1) / kPlanWavesPerBlock)),
                       dim3(kPlanWavesPerBlock * 64), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

extern "C" int wgnn_tile_plan_fill(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* slot_vrow,
                                   const int32_t* vrow_row, const int32_t* vrow_part, const int32_t* vrow_k,
                                   const int32_t* flat_t, const int32_t* tile_hdr, int64_t n_flat, int32_t waves, int32_t rpw,
                                   int32_t nblk_max, int32_t block_rows, const int32_t* seg_total, const int32_t* seg_pairs,
                                   const int32_t* seg_ptr, int32_t* entries, void* stream) {
    (void)seg_pairs;
    if (n_flat < 0 || n_flat > INT_MAX / 256) return n_flat < 0 ? WGNN_ERR_BAD_ARG : WGNN_ERR_UNSUPPORTED;
    PlanArgs a{rowptr, col, val, slot_vrow, vrow_row, vrow_part, vrow_k, flat_t, reinterpret_cast<const int2*>(tile_hdr),
               (int)n_flat, waves, rpw, nblk_max, block_rows, 1, const_cast<int32_t*>(seg_total),
               seg_ptr, reinterpret_cast<int2*>(entries)};
    a.nsplit = pick_nsplit((long)n_flat * waves, nblk_max);
    if (int rc = check(a)) return rc;
    if (!val || !seg_ptr || !entries) return WGNN_ERR_BAD_ARG;
    if (n_flat == 0) return WGNN_OK;
    const long n_waves = (long)n_flat * waves * a.nsplit;
    hipLaunchKernelGGL(tile_plan_fill, dim3((unsigned)((n_waves + kPlanWavesPerBlock - 1) / kPlanWavesPerBlock)),
                       dim3(kPlanWavesPerBlock * 64), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

// wgnn_plan.hip - device-side construction of the tile plan's `entries` (round 5).
//
// The tile plan re-orders the CSR into (tile, LDS block, wave) segments with the layout [unshared][pad][shared pairs]
// (include/wgnn.h, wgnn_agg_fwd_tiled).  Rounds 1-4 built it with framework index arithmetic: two stable sorts and ~40
// element-wise passes over 8e7 non-zeros, ~90 ms per direction at BASELINE cfg3 (18-22 ms after round 5's trimming) - more
// than ten forwards, paid by every one-shot inference (the reference builds a graph per prediction, predict.py:44-54).
// The structure makes a sort unnecessary: a (tile, wave) pair owns <= 64 destination rows whose non-zeros are already
// sorted by column, so ONE wavefront per (tile, wave) - lane = destination slot - walks its rows in lock step through
// the tile's source range: per LDS block it repeatedly takes the smallest pending column (a wave minimum), finds the lanes
// that hold it (a ballot = the group of entries on that source row), and knows every entry's place in the segment from
// running counts.  Round 6: the COUNT pass no longer walks groups.  A segment is padded to an EVEN number of entries whenever its
// entry count is odd (the pair count is even, so "odd unshared run" == "odd total": the pad costs no pair step, an odd chunk has
// an idle half step anyway), which makes a segment's size a function of its entry count alone - and that needs only every
// lane stepping through ITS row once per block (no wave minimum, no ballot): ~10x fewer instructions than the group walk.  FILL
// (after the prefix sum over the segments) places the unshared entries from the segment's start upwards and the shared pairs
// from its end DOWNWARDS, so it needs no pair count either; the pad lands between them.
// No atomics, deterministic.  Reference counterpart: none (DGL built its own CSR, preprocess_internal.py:215).
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

// Minimum over the 64 lanes, on the VALU's data-parallel primitives (row shifts inside a row of 16, then the row broadcasts of
// gfx9): ~8 instructions and no LDS traffic.  (The shuffle form - six ds_bpermute per call - made the LDS pipe the bottleneck
// of the whole walk: 4.6 + 7.1 ms for the two passes at cfg3.)
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x118, 0xf, 0xf, false));     // row_shr:8  -> lane 15 of a row: its minimum
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Sum over the 64 lanes, same data-parallel primitives (a row-wise inclusive scan, then the row broadcasts).
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8  -> lane 15 of a row: its sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

template <bool FILL>
__global__ void __launch_bounds__(kPlanWavesPerBlock * 64) tile_plan_walk(const PlanArgs a) {
    const int gw0 = blockIdx.x * kPlanWavesPerBlock + (int)(threadIdx.x >> 6);      // one wave per (tile, wave slot, block range)
    if (gw0 >= a.n_flat * a.waves * a.nsplit) return;
    const int lane = threadIdx.x & 63;
    const int gw = gw0 / a.nsplit, piece = gw0 - gw * a.nsplit;
    const int f = gw / a.waves, w = gw - f * a.waves;
    const int2 hdr = a.tile_hdr[f];
    const int cb = hdr.x, ce = hdr.y;
    const int v = lane < a.rpw ? a.slot_vrow[((size_t)a.flat_t[f] * a.waves + w) * a.rpw + lane] : -1;
    long base = 0;
    int stride = 1, n = 0;
    if (v >= 0) {
        const int row = a.vrow_row[v], part = a.vrow_part[v], k = a.vrow_k[v];
        const int rbeg = a.rowptr[row], rlen = a.rowptr[row + 1] - rbeg;
        n = rlen / k + (part < rlen % k ? 1 : 0);
        base = (long)rbeg + part;
        stride = k;
    }
    const int nblk = (ce - cb + a.kb - 1) / a.kb;
    const int per = (nblk + a.nsplit - 1) / a.nsplit;                        // my range of blocks
    const int b_begin = min(nblk, piece * per), b_end = min(nblk, b_begin + per);
    const int first_col = cb + b_begin * a.kb;
    int j = 0;                                   // my next non-zero: the first one at or behind my first block
    if (first_col > 0) {
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (a.col[base + (long)mid * stride] < first_col) lo = mid + 1; else hi = mid;
        }
        j = lo;
    }
    // a window of the next four columns of my row: the load that refills it is three groups ahead of its use (an L2 / HBM
    // latency per group otherwise sets the pace of the whole walk)
    auto colat = [&](int i) { return i < n ? a.col[base + (long)i * stride] : INT_MAX; };
    int c0 = colat(j), c1 = colat(j + 1), c2 = colat(j + 2), c3 = colat(j + 3);
    for (int b = b_begin; b < b_end; ++b) {
        const int lo_col = cb + b * a.kb, hi_col = min(ce, lo_col + a.kb);
        const size_t seg = ((size_t)f * a.nblk_max + b) * a.waves + w;
        if (!FILL) {
            // COUNT: my row's entries in this block; lanes run their own number of steps (a handful), one sum per block
            int cnt = 0;
            while (c0 < hi_col) {
                ++cnt; ++j;
                c0 = c1; c1 = c2; c2 = c3;
                c3 = colat(j + 3);
            }
            const int tot = wave_sum(cnt);
            if (lane == 0) a.seg_total[seg] = tot;
            continue;
        }
        const int p0 = a.seg_ptr[seg];
        int out_un = p0, out_sh = a.seg_ptr[seg + 1], last_un = 0;          // unshared upwards, pairs downwards
        for (;;) {
            const int s = wave_min(c0 < hi_col ? c0 : INT_MAX);             // the smallest pending source row of this block
            if (s == INT_MAX) break;
            const bool match = c0 == s;
            const unsigned long long M = __ballot(match);                   // the destination slots that read it: one group
            const int g = __popcll(M), p = g & ~1;                          // p of them pair up, an odd one stays unshared
            out_sh -= p;
            if (match) {
                const int r = __popcll(M & ((1ull << lane) - 1ull));        // my rank in the group (slots ascending)
                const int wbits = __float_as_int(a.val[base + (long)j * stride]);
                int meta = (lane << 8) | (s - lo_col);
                if (r < p) {
                    meta |= kPairFlag;
                    if ((r & 1) == 0) meta |= (lane + 1 + __builtin_ctzll(M >> (lane + 1))) << 16;   // the second's slot
                    a.entries[out_sh + r] = make_int2(meta, wbits);
                } else {
                    a.entries[out_un] = make_int2(meta, wbits);
                }
                ++j;
                c0 = c1; c1 = c2; c2 = c3;
                c3 = colat(j + 3);
            }
            if (g & 1) {
                last_un = ((63 - __builtin_clzll(M)) << 8) | (s - lo_col);
                ++out_un;
            }
        }
        // an odd segment: one slot is left between the unshared run and the pairs - the zero-weight filler (a copy of the entry
        // before it; every odd segment has an unshared entry, the pair count being even)
        if (out_un < out_sh && lane == 0) a.entries[out_un] = make_int2((last_un & 0xFFFF) | kPadFlag, 0);
    }
}

// wavefronts per (tile, wave): enough of them to fill the chip's 8192 wave slots twice (a few-row operand has few tiles)
int pick_nsplit(long n_tile_waves, int nblk_max) {
    long want = (2 * 8192 + n_tile_waves - 1) / (n_tile_waves > 0 ? n_tile_waves : 1);
    if (want < 1) want = 1;
    if (want > 16) want = 16;
    if (want > nblk_max) want = nblk_max;
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
    if (int rc = check(a)) return rc;
    if (n_flat == 0) return WGNN_OK;
    const long n_waves = (long)n_flat * waves * a.nsplit;
    hipLaunchKernelGGL(tile_plan_walk<false>, dim3((unsigned)((n_waves + kPlanWavesPerBlock - 1) / kPlanWavesPerBlock)),
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
    hipLaunchKernelGGL(tile_plan_walk<true>, dim3((unsigned)((n_waves + kPlanWavesPerBlock - 1) / kPlanWavesPerBlock)),
                       dim3(kPlanWavesPerBlock * 64), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

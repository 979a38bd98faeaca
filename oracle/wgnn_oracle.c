/* wgnn_oracle.c - plain-C CPU restatement of scDeepSort's aggregation (TEST INFRASTRUCTURE ONLY).
 *
 * Used by tests/ as a second, independent checker and by bench.py as the timed
 * `cpu_baseline` ("port": the reference itself cannot run - DGL 0.4.3 is absent).
 * PARITY: pinned against the reference's own Python code executed over a DGL stand-in, UNPINNED against DGL itself
 * (see oracle/wgnn_oracle.py header).
 *
 * Follows the reference's arithmetic order:
 *   message   m_e = (h[src] * alpha[k(e)]) * w_e                 models/gnn.py:54,56
 *   k(e)      gene->cell: src gene id; cell->gene: dst gene id;
 *             gene self-loop: G; cell self-loop: G+1             models/gnn.py:49-53
 *   reduce    neigh[v] = (sum_e m_e) / in_degree(v)              models/gnn.py:65  [DGL fn.mean]
 *             (in-degree counts the unit self-loop added after normalisation,
 *              utils/preprocess_internal.py:211-214)
 *   normalise w <- deg*w/sum(w) per destination                  utils/preprocess_internal.py:15-23
 * Accumulation is fp32 like DGL's CPU sum-reduce; rows are independent so OpenMP
 * over rows does not change any result.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* mode 0: rows are cells, sources are genes (alpha[col]);  mode 1: rows are genes, sources are cells (alpha[row]) */
void oracle_aggregate(const int32_t* rowptr, const int32_t* col, const float* val, const float* alpha, int mode,
                      int32_t self_idx, const float* h_src, int64_t ld_src, const float* h_self, int64_t ld_self,
                      float* out, int64_t ld_out, int64_t n_rows, int32_t D) {
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * (size_t)D);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n_rows; ++r) {
            memset(acc, 0, sizeof(float) * (size_t)D);
            const int32_t b = rowptr[r], e = rowptr[r + 1];
            for (int32_t j = b; j < e; ++j) {
                const float a = (mode == 0) ? alpha[col[j]] : alpha[r];
                const float w = val[j];
                const float* h = h_src + (int64_t)col[j] * ld_src;
                for (int32_t d = 0; d < D; ++d) acc[d] += (h[d] * a) * w;
            }
            const float as = alpha[self_idx];
            const float* hs = h_self + r * ld_self;
            const float deg = (float)(e - b + 1);
            float* o = out + r * ld_out;
            for (int32_t d = 0; d < D; ++d) o[d] = (acc[d] + (hs[d] * as) * 1.0f) / deg;
        }
        free(acc);
    }
}

void oracle_normalize_rows(const int32_t* rowptr, const float* vin, float* vout, int64_t n_rows) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        double s = 0.0;                                  /* stands in for torch.sum's pairwise fp32 reduction */
        for (int32_t j = b; j < e; ++j) s += vin[j];
        const float sf = (float)s, deg = (float)(e - b);
        for (int32_t j = b; j < e; ++j) vout[j] = deg * vin[j] / sf;
    }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Cache-blocked variant of oracle_aggregate (an honest CPU leg).  Same arithmetic per edge -
 * (h[src] * alpha[k]) * w accumulated in fp32, then (+ self) / in-degree - but the loop nest is the one a tuned CPU SpMM
 * would use, which is also the structure of the GPU tile kernel:
 *   a thread owns a TILE of `tile_rows` destination rows whose accumulators (tile_rows x D floats) stay in its L2, and walks
 *   the source table in BLOCKS of `block_rows` rows (block_rows x D floats, L2-resident too); every row of the tile keeps a
 *   cursor into its (column-sorted) CSR segment and consumes the entries that fall into the current block.
 * So a source row is fetched from L3 / DRAM once per tile instead of once per edge.  alpha of gene->cell edges is a
 * per-source-row factor: like the GPU path it is folded into the table once (pre = h * alpha, the reference's own multiply
 * order); `pre` is caller-provided scratch [n_src x D] (mode 0 only).  Results are identical to oracle_aggregate up to the
 * rounding of a*h being shared by all edges of a source row (it is the same product) - i.e. bitwise for mode 0.
 * target_clones: the AVX-512 body is picked at load time on hosts that have it (the GPU box's EPYC 9575F does).
 * ------------------------------------------------------------------------------------------------------------------- */
__attribute__((target_clones("avx512f", "avx2", "default")))
static void blocked_tile(const int32_t* rowptr, const int32_t* col, const float* val, const float* src, int64_t ld_src,
                         int64_t r0, int64_t r1, int64_t n_src, int32_t D, int32_t block_rows, float* acc, int32_t* cur) {
    for (int64_t r = r0; r < r1; ++r) cur[r - r0] = rowptr[r];
    memset(acc, 0, sizeof(float) * (size_t)(r1 - r0) * (size_t)D);
    for (int64_t s0 = 0; s0 < n_src; s0 += block_rows) {
        const int32_t hi = (int32_t)((s0 + block_rows < n_src) ? s0 + block_rows : n_src);
        for (int64_t r = r0; r < r1; ++r) {
            int32_t j = cur[r - r0];
            const int32_t e = rowptr[r + 1];
            float* a = acc + (size_t)(r - r0) * (size_t)D;
            for (; j < e && col[j] < hi; ++j) {
                const float w = val[j];
                const float* h = src + (int64_t)col[j] * ld_src;
#pragma omp simd
                for (int32_t d = 0; d < D; ++d) a[d] += h[d] * w;
            }
            cur[r - r0] = j;
        }
    }
}

void oracle_aggregate_blocked(const int32_t* rowptr, const int32_t* col, const float* val, const float* alpha, int mode,
                              int32_t self_idx, const float* h_src, int64_t ld_src, const float* h_self, int64_t ld_self,
                              float* out, int64_t ld_out, int64_t n_rows, int64_t n_src, int32_t D,
                              int32_t tile_rows, int32_t block_rows, float* pre) {
    const float* src = h_src;
    int64_t lds = ld_src;
    if (mode == 0) {                                        /* (h * alpha[gene]) once per source row */
#pragma omp parallel for schedule(static)
        for (int64_t s = 0; s < n_src; ++s) {
            const float a = alpha[s];
            for (int32_t d = 0; d < D; ++d) pre[s * (int64_t)D + d] = h_src[s * ld_src + d] * a;
        }
        src = pre; lds = D;
    }
    const int64_t n_tiles = (n_rows + tile_rows - 1) / tile_rows;
#pragma omp parallel
    {
        float* acc = (float*)aligned_alloc(64, ((sizeof(float) * (size_t)tile_rows * (size_t)D + 63) / 64) * 64);
        int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)tile_rows);
#pragma omp for schedule(dynamic, 1)
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int64_t r0 = t * tile_rows, r1 = (r0 + tile_rows < n_rows) ? r0 + tile_rows : n_rows;
            blocked_tile(rowptr, col, val, src, lds, r0, r1, n_src, D, block_rows, acc, cur);
            const float as = alpha[self_idx];
            for (int64_t r = r0; r < r1; ++r) {
                const float* a = acc + (size_t)(r - r0) * (size_t)D;
                const float* hs = h_self + r * ld_self;
                const float deg = (float)(rowptr[r + 1] - rowptr[r] + 1);
                const float ar = (mode == 1) ? alpha[r] : 1.0f;     /* cell->gene edges: the DESTINATION gene's alpha */
                float* o = out + r * ld_out;
                for (int32_t d = 0; d < D; ++d) o[d] = (a[d] * ar + hs[d] * as) / deg;
            }
        }
        free(acc); free(cur);
    }
}

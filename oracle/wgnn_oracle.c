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

// wgnn_transpose.hip - stable CSR transpose on the device (round 6): the gene-major copy of the (cells x genes) operand.
//
// The reference adds a cell->gene and a gene->cell edge for every stored expression value (preprocess_internal.py:170-173) and
// normalises per DESTINATION (:17-23), so the genes<-cells direction needs the raw values re-ordered gene-major, cells ascending
// inside a gene.  Rounds 1-5 did that with framework primitives - bincount, a stable radix sort of 8e7 keys, two 4-byte gathers,
// a repeat_interleave: 5.1 of the 7.0 ms of a cfg3 graph build.  The structure makes a sort unnecessary: cells are already in
// order, so an entry's place inside its gene's row is the number of EARLIER cells that express the gene.
//   count : the cell axis is cut into chunks; one workgroup per chunk histograms its cells' gene ids in LDS (4 bytes per gene:
//           n_cols <= 32768) and writes counts[chunk][gene]; a second kernel sums a gene's counts over the chunks
//   caller: t_rowptr = exclusive prefix sum of the per-gene totals
//   fill  : one workgroup per chunk loads base[gene] = t_rowptr[gene] + the gene's counts in the chunks before it into LDS and
//           walks its cells IN ORDER (a barrier per cell; the next cell's entries are already in registers): thread t takes
//           entry t of the cell, pos = base[gene]++ - a cell lists a gene once (the operand's precondition), so no two threads of
//           a step touch the same counter - and writes the cell id and the raw value to pos.
// Deterministic (no global atomics, LDS atomics only for order-free counting), stable, one read of the operand per pass.
// `row_keep` (optional, one byte per cell) drops cells: test cells of a predict graph feed nothing back (preprocess.py:184-187).
#include <atomic>
#include <climits>
#include "wgnn_common.h"

namespace {
using namespace wgnn;

constexpr int kTrThreads = 1024, kTrAhead = 8;
constexpr int kTrMaxCols = 32768;            // 128 KiB of LDS counters

__global__ void __launch_bounds__(kTrThreads) transpose_count(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                              const unsigned char* __restrict__ keep, long n_rows, int n_cols,
                                                              long rows_per_chunk, int* __restrict__ counts) {
    extern __shared__ int s_cnt[];
    const long c0 = (long)blockIdx.x * rows_per_chunk, c1 = min(n_rows, c0 + rows_per_chunk);
    for (int g = threadIdx.x; g < n_cols; g += kTrThreads) s_cnt[g] = 0;
    __syncthreads();
    if (c0 >= c1) {                           // (a chunk behind the last row: zero counts)
    } else if (!keep) {                       // every cell counts: the chunk's entries are one contiguous range
        const long b = rowptr[c0], e = rowptr[c1];
        for (long j = b + threadIdx.x; j < e; j += kTrThreads) atomicAdd(&s_cnt[col[j]], 1);
    } else {                                  // a wave per cell
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (long c = c0 + wave; c < c1; c += kTrThreads / 64) {
            if (!keep[c]) continue;
            const int b = rowptr[c], e = rowptr[c + 1];
            for (int j = b + lane; j < e; j += 64) atomicAdd(&s_cnt[col[j]], 1);
        }
    }
    __syncthreads();
    int* out = counts + (size_t)blockIdx.x * n_cols;
    for (int g = threadIdx.x; g < n_cols; g += kTrThreads) out[g] = s_cnt[g];
}

// per gene: total over the chunks (-> t_count) - and, second use (t_rowptr given), counts[chunk][gene] becomes the gene's first
// free position for that chunk: t_rowptr[gene] + the counts of the chunks in front
__global__ void __launch_bounds__(256) transpose_scan(int* __restrict__ counts, int n_cols, int n_chunks, int* __restrict__ t_count,
                                                      const int* __restrict__ t_rowptr) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_cols) return;
    int run = t_rowptr ? t_rowptr[g] : 0;
    for (int k = 0; k < n_chunks; ++k) {
        const int c = counts[(size_t)k * n_cols + g];
        if (t_rowptr) counts[(size_t)k * n_cols + g] = run;
        run += c;
    }
    if (t_count) t_count[g] = run;
}

__global__ void __launch_bounds__(kTrThreads) transpose_fill(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                             const float* __restrict__ val, const unsigned char* __restrict__ keep,
                                                             long n_rows, int n_cols, long rows_per_chunk,
                                                             const int* __restrict__ base, int* __restrict__ t_col,
                                                             float* __restrict__ t_val) {
    extern __shared__ int s_pos[];
    const long c0 = (long)blockIdx.x * rows_per_chunk, c1 = min(n_rows, c0 + rows_per_chunk);
    const int* in = base + (size_t)blockIdx.x * n_cols;
    for (int g = threadIdx.x; g < n_cols; g += kTrThreads) s_pos[g] = in[g];
    __syncthreads();
    // kTrAhead cells per memory round trip: their first kTrThreads entries are fetched into registers together, then the cells
    // are placed one after the other with a barrier in between (LDS work only).  One cell per trip left the walk waiting a
    // full global-memory latency per cell (4.1 ms for the cfg3 operand).
    for (long cg = c0; cg < c1; cg += kTrAhead) {
        int g[kTrAhead], b[kTrAhead], e[kTrAhead];
        float v[kTrAhead];
#pragma unroll
        for (int k = 0; k < kTrAhead; ++k) {
            const long c = cg + k;
            g[k] = -1; v[k] = 0.f; b[k] = e[k] = 0;
            if (c < c1 && (!keep || keep[c])) {
                b[k] = rowptr[c]; e[k] = rowptr[c + 1];
                const int j = b[k] + (int)threadIdx.x;
                if (j < e[k]) { g[k] = col[j]; v[k] = val[j]; }
            }
        }
#pragma unroll
        for (int k = 0; k < kTrAhead; ++k) {
            const long c = cg + k;
            if (c >= c1) break;                           // (uniform)
            if (g[k] >= 0) {
                const int pos = s_pos[g[k]];
                s_pos[g[k]] = pos + 1;                    // a cell lists a gene once: no other thread of this step owns g[k]
                t_col[pos] = (int)c; t_val[pos] = v[k];
            }
            for (int j = b[k] + kTrThreads + (int)threadIdx.x; j < e[k]; j += kTrThreads) {      // (a cell with > 1024 genes)
                const int gg = col[j];
                const int pos = s_pos[gg];
                s_pos[gg] = pos + 1;
                t_col[pos] = (int)c; t_val[pos] = val[j];
            }
            __syncthreads();                              // the next cell sees this cell's increments
        }
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per device (cf. wgnn_tiled.hip): remember per device what was raised
constexpr int kMaxDevices = 64;
std::atomic<int> g_lds_count[kMaxDevices], g_lds_fill[kMaxDevices];

int raise_lds(std::atomic<int>* marks, const void* fn, int lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return WGNN_ERR_LAUNCH;
    if (marks[dev].load(std::memory_order_acquire) >= lds) return WGNN_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return WGNN_ERR_LAUNCH;
    int seen = marks[dev].load(std::memory_order_relaxed);
    while (seen < lds && !marks[dev].compare_exchange_weak(seen, lds, std::memory_order_release)) {}
    return WGNN_OK;
}

long chunks_for(long n_rows) {                            // whole rounds of workgroups over the chip, >= 64 rows per chunk
    long n = 512;
    while (n > 1 && n_rows / n < 64) n /= 2;
    return n;
}

}  // namespace

extern "C" int wgnn_csr_transpose_workspace(int64_t n_rows, int32_t n_cols, int64_t* n_chunks, int64_t* bytes) {
    if (n_rows < 0 || n_cols <= 0 || !n_chunks || !bytes) return WGNN_ERR_BAD_ARG;
    if (n_cols > kTrMaxCols) return WGNN_ERR_UNSUPPORTED;
    *n_chunks = chunks_for(n_rows);
    *bytes = *n_chunks * (int64_t)n_cols * 4;
    return WGNN_OK;
}

extern "C" int wgnn_csr_transpose_count(const int32_t* rowptr, const int32_t* col, const uint8_t* row_keep, int64_t n_rows,
                                        int32_t n_cols, int64_t n_chunks, int32_t* counts, int32_t* t_count, void* stream) {
    if (!rowptr || !col || !counts || !t_count || n_rows < 0 || n_cols <= 0 || n_chunks <= 0) return WGNN_ERR_BAD_ARG;
    if (n_cols > kTrMaxCols || n_chunks > 65535) return WGNN_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rpc = (n_rows + n_chunks - 1) / n_chunks;
    if (int rc = raise_lds(g_lds_count, reinterpret_cast<const void*>(transpose_count), n_cols * 4)) return rc;
    hipLaunchKernelGGL(transpose_count, dim3((unsigned)n_chunks), dim3(kTrThreads), (size_t)n_cols * 4, st, rowptr, col, row_keep,
                       (long)n_rows, n_cols, rpc, counts);
    hipLaunchKernelGGL(transpose_scan, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, counts, n_cols, (int)n_chunks, t_count,
                       (const int*)nullptr);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

extern "C" int wgnn_csr_transpose_fill(const int32_t* rowptr, const int32_t* col, const float* val, const uint8_t* row_keep,
                                       int64_t n_rows, int32_t n_cols, int64_t n_chunks, int32_t* counts, const int32_t* t_rowptr,
                                       int32_t* t_col, float* t_val, void* stream) {
    if (!rowptr || !col || !val || !counts || !t_rowptr || !t_col || !t_val || n_rows < 0 || n_cols <= 0 || n_chunks <= 0)
        return WGNN_ERR_BAD_ARG;
    if (n_cols > kTrMaxCols || n_chunks > 65535) return WGNN_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rpc = (n_rows + n_chunks - 1) / n_chunks;
    if (int rc = raise_lds(g_lds_fill, reinterpret_cast<const void*>(transpose_fill), n_cols * 4)) return rc;
    hipLaunchKernelGGL(transpose_scan, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, counts, n_cols, (int)n_chunks,
                       (int*)nullptr, t_rowptr);
    hipLaunchKernelGGL(transpose_fill, dim3((unsigned)n_chunks), dim3(kTrThreads), (size_t)n_cols * 4, st, rowptr, col, val, row_keep,
                       (long)n_rows, n_cols, rpc, (const int*)counts, t_col, t_val);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

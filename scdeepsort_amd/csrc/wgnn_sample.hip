// wgnn_sample.hip - K5: seeded neighbour subsampling on the device (gfx950), static output shapes.
//
// Replaces, for the `num_neighbors > 0` training mode of the reference (train.py:37-40,71-78:
// NeighborSampler(expand_factor = num_neighbors, neighbor_type = 'in')), DGL 0.4.3's sampler: for every requested
// destination row at most k of its in-edges are drawn uniformly WITHOUT replacement.  The unit self-loop is one of the
// candidates (it is an explicit edge of the reference's graph, preprocess_internal.py:213-214), so a row with `deg`
// real in-edges has m = deg + 1 candidates and min(k, m) are drawn; fn.mean then divides by the number drawn.
//
// One wavefront per row.  m <= k: every candidate is taken.  Otherwise Robert Floyd's algorithm draws a uniform
// k-subset of {0..m-1} in k steps (j = m-k .. m-1: t ~ U{0..j}; take t unless already taken, else take j): the chosen
// set lives one element per lane (k <= 256 -> up to 4 per lane) and membership is one ballot.  Random numbers are a
// counter-based hash of (seed, step, stream, row, draw) - no generator state is read back, so the call is free of host
// synchronisation and a captured hipGraph draws a fresh sample on every replay (`step` is a device counter the caller
// increments on the stream).  Output is ELL: row i owns out_col/out_val[i*k .. i*k + out_cnt[i]), real edges first.
#include "wgnn_common.h"

namespace {
using namespace wgnn;

struct SArgs {
    const int* rowptr; const int* col; const float* val; const int* row_ids; long n_rows; int k;
    unsigned long long seed; const long long* step; int stream_id;
    int* out_col; float* out_val; int* out_cnt; float* out_self; float* out_inv;
};

__device__ __forceinline__ unsigned mix32(unsigned long long x) {          // splitmix64 finaliser, upper half
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (unsigned)(x >> 32);
}

constexpr int kMaxQ = 4;                                                   // k <= 64 * kMaxQ

__global__ void __launch_bounds__(kBlock) sample_rows(const SArgs a) {
    const int lane = threadIdx.x & 63;
    const long slot = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (slot >= a.n_rows) return;
    const long r = a.row_ids ? a.row_ids[slot] : slot;
    const int b = a.rowptr[r], deg = a.rowptr[r + 1] - b;
    const int m = deg + 1, k = a.k;
    const int Q = (min(k, m) + 63) >> 6;
    int sel[kMaxQ];
#pragma unroll
    for (int q = 0; q < kMaxQ; ++q) sel[q] = -1;
    if (m <= k) {
#pragma unroll
        for (int q = 0; q < kMaxQ; ++q) {
            const int c = q * 64 + lane;
            if (c < m) sel[q] = c;
        }
    } else {
        const unsigned long long base = a.seed ^ ((unsigned long long)a.step[0] * 0xD6E8FEB86659FD93ull) ^
                                        ((unsigned long long)(unsigned)a.stream_id * 0xA24BAED4963EE407ull) ^
                                        ((unsigned long long)r * 0x9FB21C651E98DF25ull);
        for (int i = 0; i < k; ++i) {
            const int j = m - k + i;
            const unsigned h = mix32(base + (unsigned long long)i * 0xC2B2AE3D27D4EB4Full);
            const int t = (int)(((unsigned long long)h * (unsigned long long)(j + 1)) >> 32);       // U{0..j}
            bool found = false;
#pragma unroll
            for (int q = 0; q < kMaxQ; ++q) found |= (sel[q] == t);
            const int x = __ballot(found) ? j : t;                          // wave-uniform
            if (lane == (i & 63)) {
#pragma unroll
                for (int q = 0; q < kMaxQ; ++q)
                    if (q == (i >> 6)) sel[q] = x;
            }
        }
    }
    // real edges to the front of the row's ELL slot; the self-loop candidate (index deg) becomes a flag
    int n_real = 0;
    bool self = false;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int q = 0; q < Q; ++q) {
        int s = -1;
#pragma unroll
        for (int qq = 0; qq < kMaxQ; ++qq)
            if (qq == q) s = sel[qq];
        const bool is_self = s == deg, real = s >= 0 && !is_self;
        const unsigned long long mask = __ballot(real);
        if (real) {
            const long o = slot * k + n_real + __popcll(mask & lt);
            a.out_col[o] = a.col[b + s];
            a.out_val[o] = a.val[b + s];
        }
        n_real += __popcll(mask);
        self |= __ballot(is_self) != 0;
    }
    if (lane == 0) {
        a.out_cnt[slot] = n_real;
        a.out_self[slot] = self ? 1.0f : 0.0f;
        a.out_inv[slot] = 1.0f / fmaxf(1.0f, (float)n_real + (self ? 1.0f : 0.0f));
    }
}

}  // namespace

extern "C" int wgnn_sample_rows(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* row_ids,
                                int64_t n_rows, int32_t k, uint64_t seed, const int64_t* step, int32_t stream_id,
                                int32_t* out_col, float* out_val, int32_t* out_cnt, float* out_self, float* out_inv,
                                void* stream) {
    if (!rowptr || !col || !val || !step || !out_col || !out_val || !out_cnt || !out_self || !out_inv || n_rows < 0)
        return WGNN_ERR_BAD_ARG;
    if (k <= 0) return WGNN_ERR_BAD_ARG;
    if (k > 64 * kMaxQ) return WGNN_ERR_UNSUPPORTED;
    if (n_rows == 0) return WGNN_OK;
    SArgs a{rowptr, col, val, row_ids, (long)n_rows, k, (unsigned long long)seed, reinterpret_cast<const long long*>(step),
            stream_id, out_col, out_val, out_cnt, out_self, out_inv};
    const long nb = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(sample_rows, dim3((unsigned)nb), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? WGNN_OK : WGNN_ERR_LAUNCH;
}

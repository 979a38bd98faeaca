// chunk_bench.hip - what does one CHUNK of the production entry pipeline cost, transitions included?  (round 5)
// pipe_bench.hip prices the steady state (a full chunk looped without its prologue); here the real statement of the kernels
// (ds_write of the weights + WGNN_FLAT4_ASM / WGNN_TALL_ASM: computed entry, warm-up, steps, drain) runs per chunk of n
// entries, with the packed words formed by compiler code as in consume().  16 waves x 128 VGPRs use the flat map, 8 waves x 256
// the tall map.  Timing only.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -Iscdeepsort_amd/csrc -o scratch/variants/chunk_bench scratch/chunk_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "wgnn_flat_asm.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int kLds = 158 * 1024;
struct Args { int iters; int n; int shared; int rpw; char* sink; };

template <bool TALL>
__global__ void __launch_bounds__(TALL ? 512 : 1024) chunk_k(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < kLds / 4; i += (TALL ? 512 : 1024)) {
        const unsigned h = (unsigned)i * 2654435761u;
        reinterpret_cast<float*>(smem)[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;       // random-looking data
    }
    __syncthreads();
    const int n = a.n, m = (n + 1) >> 1;
    const int n_s = a.shared ? (n & ~1) - 2 : 0;                      // all but the first pair shared (as the kernels run them)
    const int sw = max(32 - (n_s >> 1), 33 - m);
    const int wstrip = 154 * 1024 + wave * 256, wlane = wstrip + lane * 4;
    const int lane16 = lane * 16, row_mask = 0x3FF00;
    unsigned seed = lane * 2654435761u + wave * 40503u + blockIdx.x * 97u;
    for (int it = 0; it < a.iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        // an entry word as the plan stores it -> the packed word, as consume() forms it
        const int entx = (int)((seed >> 8) % 150) | (int)(((seed >> 3) % a.rpw) << 8) | (int)(((seed >> 17) % a.rpw) << 16);
        const int pk = (0 + (entx & 0xFF) * 1024) | (((entx >> 8) & 0x3F) << 2) | (((entx >> 16) & 0x3F) << 20);
        const bool mine = lane >= 64 - n;
        const int wv = mine ? (int)(seed | 0x3F000000u) & 0x3FFFFFFF : 0;
        if constexpr (TALL)
            asm volatile("ds_write_b32 %[wa], %[wv]\n\t" WGNN_TALL_ASM
                         ::[pk] "v"(pk), [wa] "v"(wlane), [wv] "v"(wv), [wb] "v"(wstrip), [lb] "v"(lane16), [mk] "v"(row_mask), [m] "s"(m), [sw] "s"(sw)
                         : "m0", "memory", "scc", WGNN_TALL_VGPRS, WGNN_HAND_SGPRS);
        else
            asm volatile("ds_write_b32 %[wa], %[wv]\n\t" WGNN_FLAT4_ASM
                         ::[pk] "v"(pk), [wa] "v"(wlane), [wv] "v"(wv), [wb] "v"(wstrip), [lb] "v"(lane16), [mk] "v"(row_mask), [m] "s"(m), [sw] "s"(sw)
                         : "m0", "memory", "scc", WGNN_HAND_VGPRS, WGNN_HAND_SGPRS);
    }
    if (a.iters == 12345) a.sink[threadIdx.x] = smem[threadIdx.x];
}

template <bool TALL> void run(int n, int shared) {
    Args a{1500, n, shared, TALL ? 50 : 16, nullptr};
    CK(hipMalloc(&a.sink, 4096));
    auto k = chunk_k<TALL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    const int threads = TALL ? 512 : 1024, waves = threads / 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), kLds, 0, a); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(256), dim3(threads), kLds, 0, a); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double entries = (double)n * a.iters * waves, chunks = (double)a.iters * waves;
    printf("%s waves %2d  n %2d  %-8s : %7.3f ms  %6.2f ns per entry per CU  %7.1f ns per chunk per CU  (cfg3 pass of 311.7 k entries per CU: %.3f ms)\n",
           TALL ? "tall" : "flat", waves, n, shared ? "shared" : "unshared", ms[2], ms[2] * 1e6 / entries, ms[2] * 1e6 / chunks, ms[2] / entries * 311.7e3);
    fflush(stdout);
}
int main() {
    for (int sh : {0, 1}) for (int n : {64, 48, 32, 16}) { run<false>(n, sh); run<true>(n, sh); }
    return 0;
}

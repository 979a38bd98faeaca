// stream_bench.hip - stand-alone L2 -> LDS stream micro-benchmark (round 5).
//
// The tile kernel re-streams the whole source table (20.5 MB at cfg3) through LDS once per tile: 10.5 GB per launch, which
// alone takes 0.84 ms = 12.5 TB/s = ~20 B/clk/CU, about a third of the L2 figure of MI355X_MICROARCH.md.  This program
// reproduces that stream without any compute (one 1024-thread workgroup per CU, 2 x 78 KiB LDS buffers, one barrier per
// block of 78 rows of 1 KiB) and varies: the cache-policy bits of global_load_lds_dwordx4, the number of issuing waves, the
// block order of the workgroups of one XCD (in phase vs rotated), the destination (LDS vs VGPRs), the table size, the
// piece width, the workgroup count, and the barrier.
//
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/stream_bench scratch/stream_bench.hip      run: ./scratch/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kKB = 78;            // rows per LDS block
constexpr int kRow = 1024;         // bytes per row

struct Args {
    const char* table;
    int n_rows;        // rows of the table
    int issuers;       // waves that issue the stream (1..16)
    int rot;           // block rotation per workgroup-in-XCD: start block = (blockIdx.x / 8) * rot  (mod nblk)
    int passes;        // whole-table passes per workgroup
    int barrier;       // 1: s_barrier per block
    int xcd_split;     // >1: XCD x streams only rows [x * n_rows / xcd_split ...) (disjoint slices: pure-L2 residency test)
    float* sink;
};

template <int AUX, int MODE>   // MODE 0: global_load_lds_dwordx4, 1: global_load_dwordx4 -> VGPR, 2: global_load_lds_dword (256 B pieces)
__global__ void __launch_bounds__(1024) stream_k(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int row0 = 0, n_rows = a.n_rows;
    if (a.xcd_split > 1) { const int x = blockIdx.x % 8 % a.xcd_split; n_rows = a.n_rows / a.xcd_split; row0 = x * n_rows; }
    const int nblk = (n_rows + kKB - 1) / kKB;
    const int start = (int)(((long)(blockIdx.x / 8) * a.rot) % nblk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < a.passes; ++p) {
        for (int bi = 0; bi < nblk; ++bi) {
            int b = bi + start; if (b >= nblk) b -= nblk;
            const int r0 = row0 + b * kKB;
            const int rows = min(kKB, row0 + n_rows - r0);
            const char* g = a.table + (size_t)r0 * kRow;
            char* l = smem + (bi & 1) * kKB * kRow;
            if (wave < a.issuers) {
                if (MODE == 0) {
                    for (int q = wave; q < rows; q += a.issuers)
                        __builtin_amdgcn_global_load_lds((gptr_t)(g + (size_t)q * kRow + lane * 16), (lptr_t)(l + q * kRow), 16, 0, AUX);
                } else if (MODE == 2) {
                    for (int q = wave; q < rows * 4; q += a.issuers)
                        __builtin_amdgcn_global_load_lds((gptr_t)(g + (size_t)q * 256 + lane * 4), (lptr_t)(l + q * 256), 4, 0, AUX);
                } else {
                    for (int q = wave; q < rows; q += a.issuers) {
                        const float4 v = *reinterpret_cast<const float4*>(g + (size_t)q * kRow + lane * 16);
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (a.barrier) __builtin_amdgcn_s_barrier();
        }
    }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) a.sink[0] = acc.x;
    if (MODE != 1 && a.sink && threadIdx.x == 0 && blockIdx.x == 0xFFFFFF) a.sink[0] = smem[lane];
}

template <int AUX, int MODE>
double run(const Args& a, int n_wg, int reps = 5) {
    const int lds = 2 * kKB * kRow;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_k<AUX, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream_k<AUX, MODE>), dim3(n_wg), dim3(1024), lds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_k<AUX, MODE>), dim3(n_wg), dim3(1024), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

static void report(const char* name, const Args& a, int n_wg, double ms) {
    const double rows = a.xcd_split > 1 ? a.n_rows / a.xcd_split : a.n_rows;
    const double bytes = (double)n_wg * a.passes * rows * kRow;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-58s wg %4d issuers %2d rot %4d bar %d : %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU@2.4GHz\n", name, n_wg, a.issuers, a.rot,
           a.barrier, ms, tbs, tbs * 1e12 / 256 / 2.4e9);
    fflush(stdout);
}

int main() {
    const int S = 20000;
    const size_t big_rows = 100000;
    char* table; CK(hipMalloc(&table, big_rows * kRow));
    CK(hipMemset(table, 1, big_rows * kRow));
    float* sink; CK(hipMalloc(&sink, 64));
    Args base{table, S, 16, 0, 2, 1, 0, sink};

    printf("== policy bits (20.5 MB table, 256 WG x 2 passes = the production stream of 10.5 GB; production: aux 0, 16 or 1 issuers) ==\n");
    { Args a = base; report("lds_dwordx4 aux=0", a, 256, run<0, 0>(a, 256)); }
    { Args a = base; report("lds_dwordx4 aux=1 (sc0)", a, 256, run<1, 0>(a, 256)); }
    { Args a = base; report("lds_dwordx4 aux=2 (nt)", a, 256, run<2, 0>(a, 256)); }
    { Args a = base; report("lds_dwordx4 aux=3 (sc0 nt)", a, 256, run<3, 0>(a, 256)); }
    { Args a = base; report("lds_dwordx4 aux=16 (sc1)", a, 256, run<16, 0>(a, 256)); }
    { Args a = base; report("lds_dwordx4 aux=17 (sc0 sc1)", a, 256, run<17, 0>(a, 256)); }
    { Args a = base; report("lds_dwordx4 aux=18 (sc1 nt)", a, 256, run<18, 0>(a, 256)); }
    { Args a = base; report("vgpr dwordx4 (plain loads, no LDS)", a, 256, run<0, 1>(a, 256)); }
    { Args a = base; report("lds_dword (256-B pieces) aux=0", a, 256, run<0, 2>(a, 256)); }

    printf("== issuing waves ==\n");
    for (int w : {1, 2, 4, 8, 16}) { Args a = base; a.issuers = w; report("lds_dwordx4 aux=0", a, 256, run<0, 0>(a, 256)); }
    for (int w : {1, 2, 4}) { Args a = base; a.issuers = w; report("lds_dwordx4 aux=2 (nt)", a, 256, run<2, 0>(a, 256)); }
    for (int w : {1, 4, 16}) { Args a = base; a.issuers = w; report("vgpr dwordx4", a, 256, run<0, 1>(a, 256)); }

    printf("== block order: workgroup i of an XCD starts at block i*rot (257 blocks; rot 8 = spread over the whole table) ==\n");
    for (int rot : {0, 1, 2, 4, 8, 64}) { Args a = base; a.rot = rot; report("lds_dwordx4 aux=0", a, 256, run<0, 0>(a, 256)); }
    for (int rot : {0, 1, 8}) { Args a = base; a.rot = rot; a.issuers = 1; report("lds_dwordx4 aux=0", a, 256, run<0, 0>(a, 256)); }
    for (int rot : {1, 8}) { Args a = base; a.rot = rot; report("lds_dwordx4 aux=2 (nt)", a, 256, run<2, 0>(a, 256)); }

    printf("== barrier ==\n");
    { Args a = base; a.barrier = 0; report("lds_dwordx4 aux=0 no barrier", a, 256, run<0, 0>(a, 256)); }
    { Args a = base; a.barrier = 0; a.issuers = 1; report("lds_dwordx4 aux=0 no barrier", a, 256, run<0, 0>(a, 256)); }

    printf("== table size (L2 = 4 MiB per XCD, MALL 256 MiB) ==\n");
    for (int rows : {1024, 2048, 3072, 4096, 8192, 20000, 50000, 100000}) {
        Args a = base; a.n_rows = rows; a.passes = std::max(1, 40000 / rows);
        char nm[96]; snprintf(nm, sizeof nm, "lds_dwordx4 aux=0, table %.1f MB", rows * 1024 / 1e6);
        report(nm, a, 256, run<0, 0>(a, 256));
    }
    for (int rows : {2048, 20000, 100000}) {
        Args a = base; a.n_rows = rows; a.passes = std::max(1, 40000 / rows);
        char nm[96]; snprintf(nm, sizeof nm, "vgpr dwordx4, table %.1f MB", rows * 1024 / 1e6);
        report(nm, a, 256, run<0, 1>(a, 256));
    }
    printf("== every XCD its own 1/8 slice of a 20.5 MB table (2.56 MB per XCD: L2-resident after the first pass) ==\n");
    { Args a = base; a.xcd_split = 8; a.passes = 16; report("lds_dwordx4 aux=0 xcd-sliced", a, 256, run<0, 0>(a, 256)); }
    { Args a = base; a.xcd_split = 8; a.passes = 16; report("vgpr dwordx4 xcd-sliced", a, 256, run<0, 1>(a, 256)); }
    { Args a = base; a.xcd_split = 8; a.passes = 16; a.issuers = 1; report("lds_dwordx4 aux=0 xcd-sliced", a, 256, run<0, 0>(a, 256)); }

    printf("== workgroup count (512 = two rounds, the production launch) ==\n");
    { Args a = base; a.passes = 1; report("lds_dwordx4 aux=0", a, 512, run<0, 0>(a, 512)); }
    { Args a = base; a.passes = 1; a.issuers = 1; report("lds_dwordx4 aux=0", a, 512, run<0, 0>(a, 512)); }
    { Args a = base; a.passes = 2; report("lds_dwordx4 aux=0", a, 128, run<0, 0>(a, 128)); }
    { Args a = base; a.passes = 2; report("lds_dwordx4 aux=0", a, 64, run<0, 0>(a, 64)); }
    { Args a = base; a.passes = 2; report("lds_dwordx4 aux=0", a, 8, run<0, 0>(a, 8)); }
    return 0;
}

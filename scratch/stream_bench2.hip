// stream_bench2.hip - how fast can ONE wave (or a few) issue the L2 -> LDS stream?  (round 5, follow-up to stream_bench.hip)
//
// stream_bench.hip showed that the 12.5 TB/s of the production stream is not an L2 or policy limit (16 issuing waves move the
// same 10.5 GB at 20 TB/s, 30 TB/s without the per-block drain) but the issue rate of the lone loader wave: ~52 clk per
// 1-KiB global_load_lds_dwordx4.  This program times hand-written issue loops of the loader: the production form (one
// piece, M0 += 1 KiB, lane offset += 1 KiB), four / eight pieces per address update through the instruction's immediate
// offset (which moves BOTH the global and the LDS address), a scalar-base form without any VALU, two alternating address
// registers, and 1 / 2 / 3 / 4 loader waves; every variant is verified (the last block's LDS image against the table).
//
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/variants/stream_bench2 scratch/stream_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kKB = 72;            // rows per LDS block (a multiple of 8 and of 1..4 loader waves x 8... see groups below)
constexpr int kRow = 1024;

struct Args {
    const char* table; int n_rows; int loaders; int passes; int hog;   // hog: the other waves run a VALU/LDS loop meanwhile
    unsigned* bad;
};

// VAR 0: production lean loop (1 piece per address update)   1: 4 pieces per update (offset 0..3072)
// VAR 2: 8 pieces per update (offset -4096..3072)            3: 4 pieces, scalar base advance (no VALU)
// VAR 4: 1 piece per update, two alternating lane-offset registers
template <int VAR>
__device__ __forceinline__ void issue_block(const char* g, int lds, int lane16, int wave, int nw) {
    // the wave takes groups wave, wave + nw, ... of GROUP rows
    constexpr int GROUP = VAR == 2 ? 8 : (VAR == 1 || VAR == 3 ? 4 : 1);
    const int n_groups = kKB / GROUP;
    const int mine = n_groups > wave ? (n_groups - wave + nw - 1) / nw : 0;
    if (mine == 0) return;
    const char* g0 = g + (size_t)wave * GROUP * kRow + (VAR == 2 ? 4096 : 0);
    const int l0 = lds + wave * GROUP * kRow + (VAR == 2 ? 4096 : 0);
    const int step = nw * GROUP * kRow;
    int left = mine;
    if constexpr (VAR == 0) {
        asm volatile("s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\tv_mov_b32 v46, %[vo]\n\t"
                     ".Lp_%=:\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95]\n\ts_add_u32 m0, m0, %[st]\n\tv_add_u32 v46, %[st], v46\n\t"
                     "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
                     : [left] "+s"(left) : [g] "s"(g0), [l] "s"(l0), [vo] "v"(lane16), [st] "s"(step)
                     : "memory", "scc", "s94", "s95", "v46");
    } else if constexpr (VAR == 1) {
        asm volatile("s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\tv_mov_b32 v46, %[vo]\n\t"
                     ".Lp_%=:\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95]\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:1024\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:2048\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:3072\n\t"
                     "s_add_u32 m0, m0, %[st]\n\tv_add_u32 v46, %[st], v46\n\t"
                     "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
                     : [left] "+s"(left) : [g] "s"(g0), [l] "s"(l0), [vo] "v"(lane16), [st] "s"(step)
                     : "memory", "scc", "s94", "s95", "v46");
    } else if constexpr (VAR == 2) {
        asm volatile("s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\tv_mov_b32 v46, %[vo]\n\t"
                     ".Lp_%=:\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:-4096\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:-3072\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:-2048\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:-1024\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95]\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:1024\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:2048\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95] offset:3072\n\t"
                     "s_add_u32 m0, m0, %[st]\n\tv_add_u32 v46, %[st], v46\n\t"
                     "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
                     : [left] "+s"(left) : [g] "s"(g0), [l] "s"(l0), [vo] "v"(lane16), [st] "s"(step)
                     : "memory", "scc", "s94", "s95", "v46");
    } else if constexpr (VAR == 3) {
        asm volatile("s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\t"
                     ".Lp_%=:\n\t"
                     "global_load_lds_dwordx4 %[vo], s[94:95]\n\t"
                     "global_load_lds_dwordx4 %[vo], s[94:95] offset:1024\n\t"
                     "global_load_lds_dwordx4 %[vo], s[94:95] offset:2048\n\t"
                     "global_load_lds_dwordx4 %[vo], s[94:95] offset:3072\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_add_u32 s94, s94, %[st]\n\ts_addc_u32 s95, s95, 0\n\t"
                     "s_sub_u32 %[left], %[left], 1\n\ts_cmp_lg_u32 %[left], 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
                     : [left] "+s"(left) : [g] "s"(g0), [l] "s"(l0), [vo] "v"(lane16), [st] "s"(step)
                     : "memory", "scc", "s94", "s95");
    } else {
        // two pieces per trip on two lane-offset registers (the VALU update of one never follows its own DMA directly)
        const int pairs = mine / 2, odd = mine & 1;
        int lp = pairs;
        asm volatile("s_mov_b64 s[94:95], %[g]\n\ts_mov_b32 m0, %[l]\n\tv_mov_b32 v46, %[vo]\n\tv_add_u32 v47, %[st], %[vo]\n\t"
                     "s_lshl_b32 s93, %[st], 1\n\t"
                     "s_cmp_eq_u32 %[lp], 0\n\ts_cbranch_scc1 .Lt_%=\n\t"
                     ".Lp_%=:\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95]\n\ts_add_u32 m0, m0, %[st]\n\t"
                     "global_load_lds_dwordx4 v47, s[94:95]\n\ts_add_u32 m0, m0, %[st]\n\t"
                     "v_add_u32 v46, s93, v46\n\tv_add_u32 v47, s93, v47\n\t"
                     "s_sub_u32 %[lp], %[lp], 1\n\ts_cmp_lg_u32 %[lp], 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
                     ".Lt_%=:\n\t"
                     "s_cmp_eq_u32 %[odd], 0\n\ts_cbranch_scc1 .Le_%=\n\t"
                     "global_load_lds_dwordx4 v46, s[94:95]\n\t"
                     ".Le_%=:\n\t"
                     : [lp] "+s"(lp) : [g] "s"(g0), [l] "s"(l0), [vo] "v"(lane16), [st] "s"(step), [odd] "s"(odd)
                     : "memory", "scc", "s93", "s94", "s95", "v46", "v47");
    }
}

template <int VAR>
__global__ void __launch_bounds__(1024) stream_k(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nblk = a.n_rows / kKB;
    float4 h = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int p = 0; p < a.passes; ++p) {
        for (int b = 0; b < nblk; ++b) {
            const char* g = a.table + (size_t)b * kKB * kRow;
            const int l = (int)(size_t)smem + (b & 1) * kKB * kRow;
            if (wave < a.loaders) issue_block<VAR>(g, l, lane * 16, wave, a.loaders);
            else if (a.hog) {                 // stand-in for the computing waves: LDS reads + FMAs on the OTHER buffer
                const float4* lb = reinterpret_cast<const float4*>(smem + ((b & 1) ^ 1) * kKB * kRow) + lane;
                for (int j = 0; j < a.hog; ++j) {
                    const float4 x = lb[((j * 7 + wave) % kKB) * 64];
                    h.x = fmaf(x.x, 0.5f, h.x); h.y = fmaf(x.y, 0.5f, h.y); h.z = fmaf(x.z, 0.5f, h.z); h.w = fmaf(x.w, 0.5f, h.w);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (h.x == 12345.678f) a.bad[1] = 1;
    // verify the LAST block's LDS image against the table
    __syncthreads();
    const unsigned* want = reinterpret_cast<const unsigned*>(a.table + (size_t)(nblk - 1) * kKB * kRow);
    const unsigned* got = reinterpret_cast<const unsigned*>(smem + ((nblk - 1) & 1) * kKB * kRow);
    unsigned bad = 0;
    for (int i = threadIdx.x; i < kKB * kRow / 4; i += 1024) bad += want[i] != got[i];
    if (bad) atomicAdd(a.bad, bad);
}

__global__ void fill_k(unsigned* t, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) t[i] = (unsigned)i * 2654435761u + 12345u;
}

template <int VAR>
void run(const char* name, Args a, int n_wg = 256, int reps = 5) {
    const int lds = 2 * kKB * kRow;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(a.bad, 0, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream_k<VAR>), dim3(n_wg), dim3(1024), lds, 0, a);
    CK(hipDeviceSynchronize());
    unsigned bad[2]; CK(hipMemcpy(bad, a.bad, 8, hipMemcpyDeviceToHost));
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_k<VAR>), dim3(n_wg), dim3(1024), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double t = ms[ms.size() / 2];
    const int nblk = a.n_rows / kKB;
    const double bytes = (double)n_wg * a.passes * nblk * kKB * kRow;
    const double pieces_per_wave = (double)a.passes * nblk * kKB / std::max(1, a.loaders);
    printf("%-44s loaders %d hog %3d : %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU  %5.1f clk/piece/wave  %s\n", name, a.loaders, a.hog, t,
           bytes / (t * 1e-3) / 1e12, bytes / (t * 1e-3) / 256 / 2.4e9, t * 1e-3 * 2.4e9 / pieces_per_wave,
           bad[0] ? "LDS IMAGE WRONG" : "verified");
    fflush(stdout);
}

int main() {
    const int S = 19944;                                   // 277 blocks of 72 rows (~20.4 MB)
    char* table; CK(hipMalloc(&table, (size_t)S * kRow));
    hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, reinterpret_cast<unsigned*>(table), (size_t)S * kRow / 4);
    unsigned* bad; CK(hipMalloc(&bad, 8));
    Args base{table, S, 1, 2, 0, bad};
    printf("== one loader wave, issue-loop variants (10.5 GB: 256 WG x 2 passes x 20.4 MB) ==\n");
    run<0>("1 piece / update (production form)", base);
    run<4>("1 piece / update, two offset registers", base);
    run<1>("4 pieces / update (imm offset)", base);
    run<3>("4 pieces / update, scalar base (no VALU)", base);
    run<2>("8 pieces / update (imm offset -4096..3072)", base);
    printf("== loader waves ==\n");
    for (int L : {2, 3, 4}) { Args a = base; a.loaders = L; run<0>("1 piece / update (production form)", a); }
    for (int L : {2, 3, 4}) { Args a = base; a.loaders = L; run<1>("4 pieces / update (imm offset)", a); }
    { Args a = base; a.loaders = 3; run<2>("8 pieces / update", a); }
    printf("== next to busy waves (the other waves read LDS rows + FMA: hog = row reads per wave per block) ==\n");
    for (int hog : {32, 64}) {
        { Args a = base; a.hog = hog; run<0>("1 piece / update (production form)", a); }
        { Args a = base; a.hog = hog; run<1>("4 pieces / update (imm offset)", a); }
        { Args a = base; a.hog = hog; a.loaders = 2; run<0>("1 piece / update (production form)", a); }
        { Args a = base; a.hog = hog; a.loaders = 2; run<1>("4 pieces / update (imm offset)", a); }
        { Args a = base; a.hog = hog; a.loaders = 0; run<0>("no stream at all (hog only)", a); }
    }
    return 0;
}

#!/usr/bin/env python3
"""Generates scratch/pipe_bench.hip: what does ONE entry cost in the entry pipeline of the tile kernel, by how many entries
share an LDS row read (1 = unshared, 2 = the shared pairs of round 3, 4 / 8 = runs of a tall wave) and by occupancy
(16 waves x 128 VGPRs = 4 per SIMD, the production shape; 8 waves x 256 VGPRs = 2 per SIMD, the shape that holds 49
destination rows per wave)?  Steady state only: every wave runs the generated straight-line steps over one 64-entry chunk in a
loop, software-pipelined exactly like csrc/gen_flat_asm.py (row read one step ahead, packed words / addresses two steps ahead),
no DMA, no barriers, no chunk prologue.  Timing only - the accumulators hold garbage.

Register map (literal): v32 packed words (lane j = entry j) ; v33 lane*16 ; v34 row mask ; v35 weight-strip address
  v[36:39] X0, v[40:43] X1 (row staging, singles use both per step) ; v[44:47] X0b, v[48:51] X1b (second buffer set)
  v[52:55] WA, v[56:59] WB (weights) ; v60, v61 LDS addresses ; accumulators from v64 ; s[80:91] packed-word sets
"""
import sys

LO, HI = "op_sel_hi:[0,1,1]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]"


def fma(x, w, sel):
    return [f"v_pk_fma_f32 v[64:65], v[{w}:{w + 1}], v[{x}:{x + 1}], v[64:65] {sel}",
            f"v_pk_fma_f32 v[66:67], v[{w}:{w + 1}], v[{x + 2}:{x + 3}], v[66:67] {sel}"]


def sset(p, n=3, width=2):
    return 80 + width * (p % n)


def singles(nsteps=32):
    out = []
    for p in range(nsteps):
        xa, xb = (36, 40) if p % 2 == 0 else (44, 48)          # staging of pair p
        nxa, nxb = (44, 48) if p % 2 == 0 else (36, 40)        # staging of pair p+1
        w, nw = (52, 56)[p % 2], (52, 56)[(p + 1) % 2]
        s, s2 = sset(p), sset(p + 2)
        out += [f"ds_read_b128 v[{nxa}:{nxa + 3}], v60", f"ds_read_b128 v[{nxb}:{nxb + 3}], v61",
                f"ds_read_b64 v[{nw}:{nw + 1}], v35 offset:{8 * ((p + 1) % 32)}",
                f"v_readlane_b32 s{s2}, v32, {2 * ((p + 2) % 32)}", f"v_readlane_b32 s{s2 + 1}, v32, {2 * ((p + 2) % 32) + 1}",
                f"v_and_or_b32 v60, s{s2}, v34, v33", f"v_and_or_b32 v61, s{s2 + 1}, v34, v33",
                "s_waitcnt lgkmcnt(3)",
                f"s_set_gpr_idx_on s{s}, gpr_idx(SRC2,DST)"] + fma(xa, w, LO) + [f"s_set_gpr_idx_idx s{s + 1}"] + fma(xb, w, HI) + \
               ["s_set_gpr_idx_off", f"s_cmp_eq_u32 s92, {p + 100}", "s_cbranch_scc1 .Lend_%="]
    return out, 2 * nsteps


def pairs(nsteps=32):
    out = []
    for p in range(nsteps):
        x, nx = (36, 44)[p % 2], (36, 44)[(p + 1) % 2]
        w, nw = (52, 56)[p % 2], (52, 56)[(p + 1) % 2]
        s, s2 = sset(p), sset(p + 2)
        out += [f"ds_read_b128 v[{nx}:{nx + 3}], v60", f"ds_read_b64 v[{nw}:{nw + 1}], v35 offset:{8 * ((p + 1) % 32)}",
                f"v_readlane_b32 s{s2}, v32, {2 * ((p + 2) % 32)}",
                f"v_and_or_b32 v60, s{s2}, v34, v33",
                "s_waitcnt lgkmcnt(2)",
                f"s_set_gpr_idx_on s{s}, gpr_idx(SRC2,DST)"] + fma(x, w, LO) + \
               [f"s_lshr_b32 s93, s{s}, 18", "s_set_gpr_idx_idx s93"] + fma(x, w, HI) + ["s_set_gpr_idx_off"]
    return out, 2 * nsteps


def runs(k, nsteps):
    """k entries (4 or 8) on one LDS row per step: one row read, k/4 broadcast weight reads, k/2 packed words."""
    out = []
    nword = k // 2
    for q in range(nsteps):
        x, nx = (36, 44)[q % 2], (36, 44)[(q + 1) % 2]
        wbase, nwbase = ((52, 56) if k == 4 else (52, 40))[q % 2], ((52, 56) if k == 4 else (52, 40))[(q + 1) % 2]
        # k = 8: weights of step q in 8 registers: v[52:59] / v[40:43]+v[48:51] would collide with staging; use v[52:59] and v[84..] is acc - so
        # octets alternate between v[52:59] and v[40:43] + v[48:51] (the X1 / X1b staging, unused by this stream)
        s, s2 = sset(q, 3, nword), sset(q + 2, 3, nword)
        reads = [f"ds_read_b128 v[{nx}:{nx + 3}], v60"]
        if k == 4:
            reads += [f"ds_read_b128 v[{nwbase}:{nwbase + 3}], v35 offset:{16 * ((q + 1) % nsteps)}"]
            wregs = [wbase, wbase, wbase + 2, wbase + 2]
        else:
            if (q + 1) % 2 == 0:
                reads += [f"ds_read_b128 v[52:55], v35 offset:{32 * ((q + 1) % nsteps)}", f"ds_read_b128 v[56:59], v35 offset:{32 * ((q + 1) % nsteps) + 16}"]
            else:
                reads += [f"ds_read_b128 v[40:43], v35 offset:{32 * ((q + 1) % nsteps)}", f"ds_read_b128 v[48:51], v35 offset:{32 * ((q + 1) % nsteps) + 16}"]
            wr = [52, 52, 54, 54, 56, 56, 58, 58] if q % 2 == 0 else [40, 40, 42, 42, 48, 48, 50, 50]
            wregs = wr
        lanes = [f"v_readlane_b32 s{s2 + j}, v32, {k * ((q + 2) % nsteps) + 2 * j}" for j in range(nword)]
        addr = [f"v_and_or_b32 v60, s{s2}, v34, v33"]
        wait = [f"s_waitcnt lgkmcnt({len(reads)})"]
        f = [f"s_set_gpr_idx_on s{s}, gpr_idx(SRC2,DST)"]
        for e in range(k):
            word = s + e // 2
            if e > 0:
                if e % 2 == 0:
                    f += [f"s_set_gpr_idx_idx s{word}"]
                else:
                    f += [f"s_lshr_b32 s93, s{word}, 18", "s_set_gpr_idx_idx s93"]
            f += fma(x, wregs[e], LO if e % 2 == 0 else HI)
        f += ["s_set_gpr_idx_off"]
        out += reads + lanes + addr + wait + f
    return out, k * nsteps


def kernel(name, waves, body, n_entries, max_vgpr):
    asm = "".join(f'        "{ln}\\n\\t"\n' for ln in body)
    clob = ", ".join(f'"v{i}"' for i in range(32, max_vgpr + 1)) + ", " + ", ".join(f'"s{i}"' for i in range(80, 96))
    return f'''
__global__ void __launch_bounds__({waves * 64}) {name}(const Args a) {{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kLds / 4; i += {waves * 64}) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 1023);
    __syncthreads();
    // packed words: LDS row address (a multiple of 1024 inside the first 150 KiB) | 4 * slot | 4 * slot2 << 18
    const unsigned h = (unsigned)(lane * 2654435761u + wave * 40503u + blockIdx.x * 97u);
    const int row = (h >> 8) % 150, s0 = (h >> 3) % a.rpw, s1 = (h >> 17) % a.rpw;
    const int pk = (row << 10) | (s0 << 2) | (s1 << 20);
    const int wstrip = 150 * 1024 + wave * 256;
    int iters = a.iters;
    asm volatile(
        "v_mov_b32 v32, %[pk]\\n\\tv_mov_b32 v33, %[lb]\\n\\tv_mov_b32 v34, 0x3FF00\\n\\tv_mov_b32 v35, %[wb]\\n\\t"
        "v_mov_b32 v60, %[lb]\\n\\tv_mov_b32 v61, %[lb]\\n\\ts_mov_b32 s92, 0\\n\\t"
        "v_readlane_b32 s80, v32, 0\\n\\tv_readlane_b32 s81, v32, 1\\n\\tv_readlane_b32 s82, v32, 2\\n\\tv_readlane_b32 s83, v32, 3\\n\\t"
        "v_readlane_b32 s84, v32, 4\\n\\tv_readlane_b32 s85, v32, 5\\n\\tv_readlane_b32 s86, v32, 6\\n\\tv_readlane_b32 s87, v32, 7\\n\\t"
        "v_readlane_b32 s88, v32, 8\\n\\tv_readlane_b32 s89, v32, 9\\n\\tv_readlane_b32 s90, v32, 10\\n\\tv_readlane_b32 s91, v32, 11\\n\\t"
        ".Lloop_%=:\\n\\t"
{asm}        "s_sub_u32 %[it], %[it], 1\\n\\ts_cmp_lg_u32 %[it], 0\\n\\ts_cbranch_scc1 .Lloop_%=\\n\\t"
        ".Lend_%=:\\n\\ts_waitcnt lgkmcnt(0)"
        : [it] "+s"(iters)
        : [pk] "v"(pk), [lb] "v"(lane * 16), [wb] "v"(wstrip)
        : "memory", "scc", "m0", {clob});
    if (iters == 12345) a.sink[threadIdx.x] = smem[threadIdx.x];
}}
static const int {name}_entries = {n_entries};
'''


HEAD = r'''// GENERATED by scratch/gen_pipe_bench.py - do not edit.  See that file.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
constexpr int kLds = 156 * 1024;
struct Args { int iters; int rpw; char* sink; };
'''

MAIN = r'''
template <typename K>
double timeit(K kern, int threads, Args a) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), kLds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), kLds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2];
}
static void line(const char* what, int waves, int share, double ms, double entries_per_cu) {
    // cfg3 cells<-genes: 79.8 M entries per pass = 311.7 k per CU
    printf("%-10s waves %2d share %d : %8.3f ms for %.0f k entries per CU -> %6.2f ns per entry per CU, a cfg3 pass of 311.7 k entries per CU = %.3f ms\n",
           what, waves, share, ms, entries_per_cu / 1e3, ms * 1e6 / entries_per_cu, ms / entries_per_cu * 311.7e3);
    fflush(stdout);
}
'''


def main(path):
    src = HEAD
    calls = []
    for waves, maxv, rpw in ((16, 127, 16), (8, 251, 47)):
        for share, (body, n) in ((1, singles()), (2, pairs()), (4, runs(4, 16)), (8, runs(8, 8))):
            name = f"pipe_w{waves}_s{share}"
            src += kernel(name, waves, body, n, maxv)
            calls.append((name, waves, share, rpw))
    src += MAIN + "int main() {\n    char* sink; CK(hipMalloc(&sink, 4096));\n"
    for name, waves, share, rpw in calls:
        src += (f"    {{ Args a{{2000, {rpw}, sink}}; double ms = timeit({name}, {waves * 64}, a); "
                f"line(\"{name}\", {waves}, {share}, ms, (double){name}_entries * 2000 * {waves}); }}\n")
    src += "    return 0;\n}\n"
    open(path, "w").write(src)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "scratch/pipe_bench.hip")

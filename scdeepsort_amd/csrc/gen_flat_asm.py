#!/usr/bin/env python3
"""Generates wgnn_flat_asm.inc: the straight-line, software-pipelined entry loop of agg_tiled_flat (gfx950).

One chunk = up to 64 entries of a (wave, LDS block), RIGHT-aligned in the wave: entry pair p (p = 0..31) sits in lanes
2p, 2p+1 and a chunk of m pairs occupies pairs 32-m .. 31, so the code always ENDS at the same place and only the entry
point depends on m.  Per pair the pipeline is

    step p-2 : v_readlane pk of pair p -> SGPR set p%3 ; v_and_or -> LDS addresses of pair p
    step p-1 : ds_read_b128 x2 of pair p -> staging buffer X[p%2] ; ds_read_b64 (broadcast) of its two weights -> W[p%2]
    step p   : s_waitcnt ; 4 x v_pk_fma_f32 into the GPR-indexed accumulators

(the kernel is VALU-issue bound - 4 VALU per entry: 1 readlane, 1 address, 2 FMA - so the weights come through an LDS
broadcast read of the wave's 256-byte weight strip instead of a second v_readlane)

so every producer is >= 7 instructions ahead of its consumer and no instruction waits on the one before it.
A chunk of m pairs enters through PRE(32-m): the two warm-up steps without their (meaningless) FMAs.

Register contract (literal registers, hidden from the compiler by amdgpu_num_vgpr / amdgpu_num_sgpr):
    v[64:127]  accumulators (16 destination slots x float4 per lane)
    v[48:55]   staging XA, v[56:63] staging XB
    v44, v45   LDS byte addresses of the pair being fetched next
    v[40:41]   W0, v[42:43] W1: the two weights of a pair in every lane (op_sel picks the half)
    s[80:85]   three scalar sets {pk0, pk1}; pk = LDS byte address of the source row (a multiple of 1024) | 4*slot:
               its low byte is the GPR index, (pk & %[mk]) | %[lb] the lane's LDS address
    s92        scratch ; s[94:95] computed-branch target
Operands: %[pk] (VGPR: packed entry), %[wb] (VGPR, uniform: LDS address of the wave's weight strip, entry j at +4j),
%[lb] (VGPR: lane*16), %[mk] (VGPR: 0xFFFFFC00), %[m] (SGPR: number of pairs, 1..32).
"""
import os
import sys

N_PAIRS = 32
ABLATE = set(filter(None, os.environ.get("WGNN_GEN_ABLATE", "").split(",")))   # timing experiments only (wrong results)
XREG = {0: (48, 52), 1: (56, 60)}          # staging buffers by pair parity: first regs of entry 0 / entry 1


def sset(p):
    base = 80 + 2 * (p % 3)
    return dict(pk0=base, pk1=base + 1)


def wreg(p):
    return 40 + 2 * (p % 2)


def readlanes(p):
    s = sset(p)
    out = [f"v_readlane_b32 s{s['pk0']}, %[pk], {2 * p}",
           f"v_readlane_b32 s{s['pk1']}, %[pk], {2 * p + 1}"]
    return out


def addresses(p):
    s = sset(p)
    if "noaddr" in ABLATE:
        return []
    return [f"v_and_or_b32 v44, s{s['pk0']}, %[mk], %[lb]",
            f"v_and_or_b32 v45, s{s['pk1']}, %[mk], %[lb]"]


def reads(p):
    x0, x1 = XREG[p % 2]
    if "nords" in ABLATE:
        return []
    w = wreg(p)
    return [f"ds_read_b128 v[{x0}:{x0 + 3}], v44",
            f"ds_read_b128 v[{x1}:{x1 + 3}], v45",
            f"ds_read_b64 v[{w}:{w + 1}], %[wb] offset:{8 * p}"]


def fmas(p):
    s = sset(p)
    x0, x1 = XREG[p % 2]
    if "nofma" in ABLATE:
        return []
    w = f"v[{wreg(p)}:{wreg(p) + 1}]"
    lo, hi = "op_sel_hi:[0,1,1]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
    return [f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)",
            f"v_pk_fma_f32 v[64:65], {w}, v[{x0}:{x0 + 1}], v[64:65] {lo}",
            f"v_pk_fma_f32 v[66:67], {w}, v[{x0 + 2}:{x0 + 3}], v[66:67] {lo}",
            f"s_set_gpr_idx_idx s{s['pk1']}",
            f"v_pk_fma_f32 v[64:65], {w}, v[{x1}:{x1 + 1}], v[64:65] {hi}",
            f"v_pk_fma_f32 v[66:67], {w}, v[{x1 + 2}:{x1 + 3}], v[66:67] {hi}",
            "s_set_gpr_idx_off"]


def step(p):
    """Steady-state step p: fetch pair p+1, prepare pair p+2, accumulate pair p."""
    out = [f".Lw4_step{p}_%=:"]
    if p + 1 < N_PAIRS:
        out += reads(p + 1)
    if p + 2 < N_PAIRS:
        out += readlanes(p + 2)
    if "nords" in ABLATE:
        pass
    else:
        out.append("s_waitcnt lgkmcnt(3)" if p + 1 < N_PAIRS else "s_waitcnt lgkmcnt(0)")
    out += fmas(p)
    if p + 2 < N_PAIRS:
        out += addresses(p + 2)
    if "xrl" in ABLATE:
        out += ["v_readlane_b32 s92, %[pk], 3", "v_readlane_b32 s92, %[pk], 5"]
    if "xvmov" in ABLATE:
        out += ["v_mov_b32 v39, v39", "v_mov_b32 v39, v39"]
    if "xsalu" in ABLATE:
        out += ["s_mov_b32 s92, s92", "s_mov_b32 s92, s92"]
    if "xlds" in ABLATE:
        out += ["ds_read_b32 v39, %[wb]"]
    return out


def pre(p0):
    """Warm-up for a chunk whose first pair is p0: what steps p0-2 and p0-1 would do, minus their FMAs."""
    out = [f".Lw4_pre{p0}_%=:"]
    out += readlanes(p0)
    if p0 + 1 < N_PAIRS:
        out += readlanes(p0 + 1)
    out += addresses(p0)
    out += reads(p0)
    if p0 + 1 < N_PAIRS:
        out += addresses(p0 + 1)
    out.append(f"s_branch .Lw4_step{p0}_%=")
    return out


def main(path):
    lines = []
    # computed branch: table of s_branch (4 bytes each), indexed by 32 - m
    lines += ["s_getpc_b64 s[94:95]",
              ".Lw4_pc_%=:",
              "s_sub_u32 s92, 32, %[m]",
              "s_lshl_b32 s92, s92, 2",
              "s_add_u32 s92, s92, .Lw4_table_%= - .Lw4_pc_%=",
              "s_add_u32 s94, s94, s92",
              "s_addc_u32 s95, s95, 0",
              "s_setpc_b64 s[94:95]",
              ".Lw4_table_%=:"]
    for p0 in range(N_PAIRS):
        lines.append(f"s_branch .Lw4_pre{p0}_%=")
    for p0 in range(N_PAIRS):
        lines += pre(p0)
    for p in range(N_PAIRS):
        lines += step(p)
    body = "".join(f'    "{ln}\\n\\t"\n' for ln in lines)
    with open(path, "w") as f:
        f.write("// GENERATED by gen_flat_asm.py - do not edit.  See that file for the register contract.\n")
        f.write("#define WGNN_FLAT4_ASM \\\n")
        f.write("".join(f'    "{ln}\\n\\t" \\\n' for ln in lines))
        f.write('    ""\n')
    return len(lines)


if __name__ == "__main__":
    n = main(sys.argv[1] if len(sys.argv) > 1 else "wgnn_flat_asm.inc")
    print(f"{n} lines")

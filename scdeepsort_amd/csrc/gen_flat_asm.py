#!/usr/bin/env python3
"""Generates wgnn_flat_asm.inc: the straight-line, software-pipelined entry loop of agg_tiled_flat (gfx950).

One chunk = up to 64 entries of a (wave, LDS block), RIGHT-aligned in the wave: entry pair p (p = 0..31) sits in lanes
2p, 2p+1 and a chunk of m pairs occupies pairs 32-m .. 31, so the code always ENDS at the same place and only the entry
point depends on m.  Per pair the pipeline is

    step p-2 : v_readlane pk of pair p -> SGPR set p%3 ; v_and_or -> LDS addresses of pair p
    step p-1 : ds_read_b128 x2 of pair p -> staging buffer X[p%2] ; ds_read_b64 (broadcast) of its two weights -> W[p%2]
    step p   : s_waitcnt ; 4 x v_pk_fma_f32 into the GPR-indexed accumulators
Within a step the order is R(eads of p+1) L(readlanes of p+2) A(ddresses of p+2) W(ait) F(mas of p): the address VALU
work sits in the shadow of the LDS wait (measured 2-3 % faster than addresses after the FMAs).

(the kernel is VALU-issue bound - 4 VALU per entry: 1 readlane, 1 address, 2 FMA - so the weights come through an LDS
broadcast read of the wave's 256-byte weight strip instead of a second v_readlane)

so every producer is >= 7 instructions ahead of its consumer and no instruction waits on the one before it.
A chunk of m pairs enters through PRE(32-m): the two warm-up steps without their (meaningless) FMAs.

Shared pairs (round 3).  The plan puts the entries of a segment that read the SAME source row two by two at the END of
the segment (graph._pair_segment_entries), i.e. in the last pair steps of a chunk.  From step %[sw] on the pipeline
continues in a second stream (.Lw4_sstep<p>) whose step handles such a pair with ONE readlane, ONE address and ONE row
read: 6 VALU and 2 LDS operations per pair instead of 8 and 3.  The hand-over is a compare + branch after every unshared
step (SALU, free): what that step fetched - both entries of pair p+1, both addresses of pair p+2 - is a superset of what
the shared step expects, and the LDS counter is in order, so there is no drain and no second warm-up.  The first word of
a shared pair carries 4*slot of the second entry in bits 18..25 (s_lshr -> s_set_gpr_idx_idx).

Register contract (literal registers, hidden from the compiler by amdgpu_num_vgpr / amdgpu_num_sgpr):
    v[64:127]  accumulators (16 destination slots x float4 per lane)
    v[48:55]   staging XA, v[56:63] staging XB
    v44, v45   LDS byte addresses of the pair being fetched next
    v[40:41]   W0, v[42:43] W1: the two weights of a pair in every lane (op_sel picks the half)
    s[80:85]   three scalar sets {pk0, pk1}; pk = LDS byte address of the source row (a multiple of 1024) | 4*slot:
               its low byte is the GPR index, (pk & %[mk]) | %[lb] the lane's LDS address
    s86        GPR index of a shared pair's second slot ; s92 scratch ; s[94:95] computed-branch target
Operands: %[pk] (VGPR: packed entry), %[wb] (VGPR, uniform: LDS address of the wave's weight strip, entry j at +4j),
%[lb] (VGPR: lane*16), %[mk] (VGPR: 0x3FC00, the row-address bits of a packed entry), %[m] (SGPR: number of pairs,
1..32), %[sw] (SGPR: first pair step of the shared stream, > 32 - m; 32 = none).
"""
import os
import sys

N_PAIRS = 32
# Register maps (round 5: two tile geometries share this pipeline).
#   flat: 16 waves x 128 VGPRs, 16 destination rows per wave.  The compiler gets v[0:31]; v[32:127] are hand-owned.
#   tall: 8 waves x 256 VGPRs, 49 destination rows per wave.  The compiler gets v[0:41]; the staging / weight / address
#         registers v[20:41] are STATEMENT-LOCAL (dead between asm statements, named as clobbers: the compiler may use them
#         between statements and in the epilogue, never across one); v[42:255] carry state across statements (segment /
#         entry-chunk registers v[42:59], accumulators v[60:255]) and lie beyond the compiler's cap.
MAPS = {
    "flat": dict(macro="WGNN_FLAT4_ASM", acc=64, x=[(48, 52), (56, 60), (96, 100)], w=[40, 42, 104], a0=44, a1=45,
                 hand=(32, 127), clob="WGNN_HAND_VGPRS", label="w4"),
    "tall": dict(macro="WGNN_TALL_ASM", acc=60, x=[(20, 24), (28, 32)], w=[36, 38], a0=40, a1=41,
                 hand=(20, 255), clob="WGNN_TALL_VGPRS", label="w8"),
}
M = MAPS["flat"]
HAND_VGPR_FIRST, HAND_VGPR_LAST = MAPS["flat"]["hand"]
HAND_SGPR_FIRST, HAND_SGPR_LAST = 80, 95
ABLATE = set(filter(None, os.environ.get("WGNN_GEN_ABLATE", "").split(",")))   # timing experiments only (wrong results)
DEPTH = int(os.environ.get("WGNN_GEN_DEPTH", "1"))      # LDS reads are issued DEPTH steps ahead of their FMAs
def xreg(p):
    return M["x"][p % (DEPTH + 1)]    # staging buffers, first regs of entry 0 / entry 1 (flat's third: DEPTH=2 experiment)


def sset(p):
    base = 80 + 2 * (p % (DEPTH + 2))
    return dict(pk0=base, pk1=base + 1)


def wreg(p):
    return M["w"][p % (DEPTH + 1)]


def acc(k):
    """Accumulator operand of slot 0 (the GPR index adds 4 * slot): registers acc + k, acc + k + 1."""
    return f"v[{M['acc'] + k}:{M['acc'] + k + 1}]"


FMA2 = "fma2" in ABLATE          # v_fma_f32 x4 instead of v_pk_fma_f32 x2 per entry (a REAL variant: results stay correct)
WRL = "wrl" in ABLATE            # experiment: weights through v_readlane into SGPR pairs instead of the LDS strip
SCR = 92                         # scratch SGPR of the computed branch
SSLOT = 86                       # scratch SGPR: GPR index of a shared pair's second slot
SHARED = DEPTH == 1 and not WRL and "noshared" not in ABLATE and "halfmodel" not in ABLATE    # emit the shared-pair stream (see s_step)
ORDER = os.environ.get("WGNN_GEN_ORDER", "RLAWF")     # order of a step's groups: R(eads) L(readlanes) W(ait) F(mas) A(ddresses)
NORL = "norl" in ABLATE          # timing only: no per-pair v_readlane (every entry reuses the packed word of the chunk's last pair) -
                                 # the upper bound of feeding the packed words through the scalar cache instead of the VALU
NOWT = "nowt" in ABLATE          # timing only: no ds_read_b64 of the pair's weights (stale weight registers)
HALF = "halfmodel" in ABLATE     # timing only (D <= 128): the cost structure of a half-wave pipeline - the two entries of a pair share ONE
                                 # ds_read_b128 (lanes 0-31 one source row, lanes 32-63 the other) and one pair of packed FMAs
IDXMODE = "idxmode" in ABLATE    # a REAL variant (results stay correct): GPR-index mode is switched on ONCE per chunk; a step only moves
                                 # the index (s_set_gpr_idx_idx) and parks it at 0 behind its FMAs - no MODE-register write per step
WRLS = "wrls" in ABLATE          # a REAL variant: the SHARED-pair stream takes its weights through v_readlane (SGPR operands of the FMAs)
                                 # instead of the broadcast ds_read_b64 - trades 2 LDS clk per pair for 2 VALU (the LDS is the busier pipe)
SMEM = "smem" in ABLATE          # timing only, with norl,nowt: the cost side of a scalar-cache entry feed - every 4th pair step drains
                                 # lgkmcnt (SMEM returns out of order: only 0 proves a scalar load landed) and issues one
                                 # s_load_dwordx16 (8 entries = 4 pairs) from the chunk's own address (operand %[ep], clobbers s[64:79])


def smem_refill(p):
    if not SMEM or p % 4 != 3:
        return []
    return ["s_waitcnt lgkmcnt(0)", f"s_load_dwordx16 s[64:79], %[ep], {hex(64 * (p // 4))}"]


def wsgpr(p):
    base = 84 + 4 * (p % 2)          # (experiment only: overlaps nothing at DEPTH = 1 with 3 pk sets in s80..s85? no: s86+)
    base = 86 + 4 * (p % 2)
    return base, base + 2


def wreadlanes(p):
    w0, w1 = wsgpr(p)
    return [f"v_readlane_b32 s{w0}, %[wv], {2 * p}", f"v_readlane_b32 s{w1}, %[wv], {2 * p + 1}"]


def readlanes(p):
    s = sset(p)
    if NORL:
        return []
    out = [f"v_readlane_b32 s{s['pk0']}, %[pk], {2 * p}",
           f"v_readlane_b32 s{s['pk1']}, %[pk], {2 * p + 1}"]
    return out


def addresses(p):
    s = sset(p)
    if "noaddr" in ABLATE:
        return []
    return [f"v_and_or_b32 v{M['a0']}, s{s['pk0']}, %[mk], %[lb]",
            f"v_and_or_b32 v{M['a1']}, s{s['pk1']}, %[mk], %[lb]"]


def reads(p):
    x0, x1 = xreg(p)
    if "nords" in ABLATE:
        return []
    w = wreg(p)
    if WRL:
        return [f"ds_read_b128 v[{x0}:{x0 + 3}], v{M['a0']}", f"ds_read_b128 v[{x1}:{x1 + 3}], v{M['a1']}"] + wreadlanes(p)
    if HALF:
        return [f"ds_read_b128 v[{x0}:{x0 + 3}], v{M['a0']}", f"ds_read_b64 v[{w}:{w + 1}], %[wb] offset:{8 * p}"]
    return [f"ds_read_b128 v[{x0}:{x0 + 3}], v{M['a0']}",
            f"ds_read_b128 v[{x1}:{x1 + 3}], v{M['a1']}"] + ([] if NOWT else [f"ds_read_b64 v[{w}:{w + 1}], %[wb] offset:{8 * p}"])


def fmas(p):
    s = sset(p)
    x0, x1 = xreg(p)
    if "nofma" in ABLATE:
        return []
    w = f"v[{wreg(p)}:{wreg(p) + 1}]"
    lo, hi = "op_sel_hi:[0,1,1]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
    if FMA2:       # experiment: four v_fma_f32 per entry instead of two v_pk_fma_f32 (same arithmetic, same registers)
        w0, w1 = wreg(p), wreg(p) + 1
        return [f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)"] + \
               [f"v_fma_f32 v{M['acc'] + k}, v{w0}, v{x0 + k}, v{M['acc'] + k}" for k in range(4)] + \
               [f"s_set_gpr_idx_idx s{s['pk1']}"] + \
               [f"v_fma_f32 v{M['acc'] + k}, v{w1}, v{x1 + k}, v{M['acc'] + k}" for k in range(4)] + \
               ["s_set_gpr_idx_off"]
    if WRL:
        w0, w1 = wsgpr(p)
        return [f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)",
                f"v_pk_fma_f32 {acc(0)}, s[{w0}:{w0 + 1}], v[{x0}:{x0 + 1}], {acc(0)} {lo}",
                f"v_pk_fma_f32 {acc(2)}, s[{w0}:{w0 + 1}], v[{x0 + 2}:{x0 + 3}], {acc(2)} {lo}",
                f"s_set_gpr_idx_idx s{s['pk1']}",
                f"v_pk_fma_f32 {acc(0)}, s[{w1}:{w1 + 1}], v[{x1}:{x1 + 1}], {acc(0)} {lo}",
                f"v_pk_fma_f32 {acc(2)}, s[{w1}:{w1 + 1}], v[{x1 + 2}:{x1 + 3}], {acc(2)} {lo}",
                "s_set_gpr_idx_off"]
    if HALF:
        return [f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)",
                f"v_pk_fma_f32 {acc(0)}, {w}, v[{x0}:{x0 + 1}], {acc(0)} {lo}",
                f"v_pk_fma_f32 {acc(2)}, {w}, v[{x0 + 2}:{x0 + 3}], {acc(2)} {lo}",
                "s_set_gpr_idx_off"]
    return [f"s_set_gpr_idx_idx s{s['pk0']}" if IDXMODE else f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)",
            f"v_pk_fma_f32 {acc(0)}, {w}, v[{x0}:{x0 + 1}], {acc(0)} {lo}",
            f"v_pk_fma_f32 {acc(2)}, {w}, v[{x0 + 2}:{x0 + 3}], {acc(2)} {lo}",
            f"s_set_gpr_idx_idx s{s['pk1']}",
            f"v_pk_fma_f32 {acc(0)}, {w}, v[{x1}:{x1 + 1}], {acc(0)} {hi}",
            f"v_pk_fma_f32 {acc(2)}, {w}, v[{x1 + 2}:{x1 + 3}], {acc(2)} {hi}",
            "s_set_gpr_idx_idx 0" if IDXMODE else "s_set_gpr_idx_off"]


def s_wsgpr(p):
    """WRLS: SGPR pairs of a shared pair's two weights (low words; the odd halves are free - s87 serves as SSLOT)."""
    return (86, 88) if p % 2 == 0 else (90, 92)


def s_wreadlanes(p):
    wa, wb = s_wsgpr(p)
    return [f"v_readlane_b32 s{wa}, %[wv], {2 * p}", f"v_readlane_b32 s{wb}, %[wv], {2 * p + 1}"]


def s_readlanes(p):
    if NORL:
        return []
    return [f"v_readlane_b32 s{sset(p)['pk0']}, %[pk], {2 * p}"]


def s_addresses(p):
    return [f"v_and_or_b32 v{M['a0']}, s{sset(p)['pk0']}, %[mk], %[lb]"]


def s_reads(p):
    x0, _ = xreg(p)
    w = wreg(p)
    return [f"ds_read_b128 v[{x0}:{x0 + 3}], v{M['a0']}"] + ([] if (NOWT or WRLS) else [f"ds_read_b64 v[{w}:{w + 1}], %[wb] offset:{8 * p}"])


def s_fmas(p):
    """A shared pair: both entries read the SAME source row (one LDS read), the second slot rides in pk0[25:18]."""
    s = sset(p)
    x0, _ = xreg(p)
    w = f"v[{wreg(p)}:{wreg(p) + 1}]"
    lo, hi = "op_sel_hi:[0,1,1]", "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
    if WRLS:
        wa, wb = s_wsgpr(p)
        return [f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)",
                f"v_pk_fma_f32 {acc(0)}, s[{wa}:{wa + 1}], v[{x0}:{x0 + 1}], {acc(0)} {lo}",
                f"v_pk_fma_f32 {acc(2)}, s[{wa}:{wa + 1}], v[{x0 + 2}:{x0 + 3}], {acc(2)} {lo}",
                f"s_lshr_b32 s87, s{s['pk0']}, 18",
                "s_set_gpr_idx_idx s87",
                f"v_pk_fma_f32 {acc(0)}, s[{wb}:{wb + 1}], v[{x0}:{x0 + 1}], {acc(0)} {lo}",
                f"v_pk_fma_f32 {acc(2)}, s[{wb}:{wb + 1}], v[{x0 + 2}:{x0 + 3}], {acc(2)} {lo}",
                "s_set_gpr_idx_off"]
    return [f"s_set_gpr_idx_idx s{s['pk0']}" if IDXMODE else f"s_set_gpr_idx_on s{s['pk0']}, gpr_idx(SRC2,DST)",
            f"v_pk_fma_f32 {acc(0)}, {w}, v[{x0}:{x0 + 1}], {acc(0)} {lo}",
            f"v_pk_fma_f32 {acc(2)}, {w}, v[{x0 + 2}:{x0 + 3}], {acc(2)} {lo}",
            f"s_lshr_b32 s{SSLOT}, s{s['pk0']}, 18",
            f"s_set_gpr_idx_idx s{SSLOT}",
            f"v_pk_fma_f32 {acc(0)}, {w}, v[{x0}:{x0 + 1}], {acc(0)} {hi}",
            f"v_pk_fma_f32 {acc(2)}, {w}, v[{x0 + 2}:{x0 + 3}], {acc(2)} {hi}",
            "s_set_gpr_idx_idx 0" if IDXMODE else "s_set_gpr_idx_off"]


def s_step(p):
    """Shared-pair step p (DEPTH = 1 only): as step(p) with one readlane, one address, one row read per pair.  Entered
    from the unshared stream after its step p-1, whose fetches (both entries of pair p, both addresses of pair p+1)
    are a superset of what this step expects; the LDS counter is in order, so lgkmcnt(2) also covers the older three."""
    out = [f".Lw4_sstep{p}_%=:"]
    R = s_reads(p + 1) if p + 1 < N_PAIRS else []
    L = s_readlanes(p + 2) if p + 2 < N_PAIRS else []
    if WRLS and p + 1 < N_PAIRS:
        L = L + s_wreadlanes(p + 1)
    A = s_addresses(p + 2) if p + 2 < N_PAIRS else []
    W = [f"s_waitcnt lgkmcnt({(1 if (NOWT or WRLS) else 2) * min(1, N_PAIRS - 1 - p)})"]
    return out + R + L + A + W + s_fmas(p) + smem_refill(p)


def step(p):
    """Steady-state step p: fetch pair p+DEPTH, prepare pair p+DEPTH+1, accumulate pair p."""
    out = [f".Lw4_step{p}_%=:"]
    R = reads(p + DEPTH) if p + DEPTH < N_PAIRS else []
    L = readlanes(p + DEPTH + 1) if p + DEPTH + 1 < N_PAIRS else []
    A = addresses(p + DEPTH + 1) if p + DEPTH + 1 < N_PAIRS else []
    ahead = min(DEPTH, N_PAIRS - 1 - p)               # pairs fetched after pair p
    W = [] if "nords" in ABLATE else [f"s_waitcnt lgkmcnt({(2 if (WRL or NOWT or HALF) else 3) * ahead})"]
    F = fmas(p)
    parts = dict(R=R, L=L, A=A, W=W, F=F)
    if ORDER == "interleave":                         # FMAs of the two entries around the scalar work
        f0, f1 = F[:3], F[3:]
        out += R + L[:1] + W + f0 + L[1:] + f1 + A
    else:
        for k in ORDER:
            out += parts[k]
    out += smem_refill(p)
    if "xrl" in ABLATE:
        out += ["v_readlane_b32 s86, %[pk], 3", "v_readlane_b32 s86, %[pk], 5"]
    if "xvmov" in ABLATE:
        out += ["v_mov_b32 v39, v39", "v_mov_b32 v39, v39"]
    if "xsalu" in ABLATE:
        out += ["s_mov_b32 s86, s86", "s_mov_b32 s86, s86"]
    if "xlds" in ABLATE:
        out += ["ds_read_b32 v39, %[wb]"]
    if SHARED and p + 1 < N_PAIRS:                    # pairs p+1 .. 31 are shared pairs: continue in that stream
        out += [f"s_cmp_eq_u32 %[sw], {p + 1}", f"s_cbranch_scc1 .Lw4_{'sentry' if WRLS else 'sstep'}{p + 1}_%="]
    return out


def pre(p0):
    """Warm-up for a chunk whose first pair is p0: what the DEPTH+1 steps before step p0 would do, minus their FMAs."""
    out = [f".Lw4_pre{p0}_%=:"]
    if NORL:                                          # all three scalar sets: the chunk's last pair (always present)
        for q in range(DEPTH + 2):
            s = sset(q)
            out += [f"v_readlane_b32 s{s['pk0']}, %[pk], 62", f"v_readlane_b32 s{s['pk1']}, %[pk], 63"]
    for k in range(DEPTH + 1):
        if p0 + k < N_PAIRS:
            out += readlanes(p0 + k)
    for k in range(DEPTH):
        if p0 + k < N_PAIRS:
            out += addresses(p0 + k)
            out += reads(p0 + k)
    if p0 + DEPTH < N_PAIRS:
        out += addresses(p0 + DEPTH)
    out.append(f"s_branch .Lw4_step{p0}_%=")
    return out


def pipeline_lines():
    """The whole straight-line pipeline of one chunk for the current register map M."""
    lines = []
    if IDXMODE:
        lines += ["s_set_gpr_idx_on 0, gpr_idx(SRC2,DST)"]
    # computed branch: table of s_branch (4 bytes each), indexed by 32 - m
    lines += ["s_getpc_b64 s[94:95]",
              ".Lw4_pc_%=:",
              f"s_sub_u32 s{SCR}, 32, %[m]",
              f"s_lshl_b32 s{SCR}, s{SCR}, 2",
              f"s_add_u32 s{SCR}, s{SCR}, .Lw4_table_%= - .Lw4_pc_%=",
              f"s_add_u32 s94, s94, s{SCR}",
              "s_addc_u32 s95, s95, 0",
              "s_setpc_b64 s[94:95]",
              ".Lw4_table_%=:"]
    for p0 in range(N_PAIRS):
        lines.append(f"s_branch .Lw4_pre{p0}_%=")
    for p0 in range(N_PAIRS):
        lines += pre(p0)
    for p in range(N_PAIRS):
        lines += step(p)
    if SHARED:
        lines.append("s_branch .Lw4_end_%=")
        for p in range(1, N_PAIRS):
            lines += s_step(p)
        if WRLS:                                          # hand-over stubs: the first shared pair's weights, then its step
            lines.append("s_branch .Lw4_end_%=")
            for p in range(1, N_PAIRS):
                lines += [f".Lw4_sentry{p}_%=:"] + s_wreadlanes(p) + [f"s_branch .Lw4_sstep{p}_%="]
        lines.append(".Lw4_end_%=:")
    if IDXMODE:
        if not SHARED:
            lines.append(".Lw4_end_%=:")
        lines.append("s_set_gpr_idx_off")
    if SMEM:
        lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def main(path):
    global M
    n = 0
    with open(path, "w") as f:
        f.write("// GENERATED by gen_flat_asm.py - do not edit.  See that file for the register contracts.\n")
        # every register the hand-written statements own, spelled out for their clobber lists (a clobber list takes single
        # registers, not ranges: naming only the end points would leave the ones in between "free")
        for name in ("flat", "tall"):
            lo, hi = MAPS[name]["hand"]
            f.write(f"#define {MAPS[name]['clob']} " + ", ".join(f'"v{i}"' for i in range(lo, hi + 1)) + "\n")
        f.write("#define WGNN_HAND_SGPRS " + ", ".join(f'"s{i}"' for i in range(HAND_SGPR_FIRST, HAND_SGPR_LAST + 1)) + "\n")
        for name in ("flat", "tall"):
            M = MAPS[name]
            if name == "tall" and (DEPTH != 1 or WRL):
                continue                                      # the timing experiments exist for the flat map only
            lines = pipeline_lines()
            n += len(lines)
            f.write(f"#define {M['macro']} \\\n")
            f.write("".join(f'    "{ln}\\n\\t" \\\n' for ln in lines))
            f.write('    ""\n')
        M = MAPS["flat"]
    return n


if __name__ == "__main__":
    n = main(sys.argv[1] if len(sys.argv) > 1 else "wgnn_flat_asm.inc")
    print(f"{n} lines")

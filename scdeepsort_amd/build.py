"""Build the gfx950 shared library in-tree (``scdeepsort_amd/libwgnn_hip.so``).

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRC = [PKG / "csrc" / "wgnn_kernels.hip", PKG / "csrc" / "wgnn_tiled.hip", PKG / "csrc" / "wgnn_linear.hip", PKG / "csrc" / "wgnn_sample.hip", PKG / "csrc" / "wgnn_train.hip", PKG / "csrc" / "wgnn_plan.hip", PKG / "csrc" / "wgnn_transpose.hip"]
LIB = PKG / "libwgnn_hip.so"
GEN = PKG / "csrc" / "gen_flat_asm.py"          # writes csrc/wgnn_flat_asm.inc (the hand-scheduled entry pipeline)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    deps = SRC + [ROOT / "include" / "wgnn.h", GEN] + sorted((PKG / "csrc").glob("*.h*")) + sorted((PKG / "csrc").glob("*.inc"))
    return any(d.stat().st_mtime > LIB.stat().st_mtime for d in deps if d.exists())


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-Wno-inline-asm", f"-I{ROOT / 'include'}"]
HAND_VGPRS = range(32, 128)         # agg_tiled_flat4: hand-owned registers (csrc/gen_flat_asm.py MAPS["flat"])
TALL_STATE_VGPRS = range(42, 256)   # agg_tiled_tall: registers that carry state ACROSS asm statements (segment / chunk registers,
                                    # accumulators); v20..v41 are statement-local there and shared with the compiler
KERNELS = {"agg_tiled_flat4": dict(hand=HAND_VGPRS, vgprs=128), "agg_tiled_tall": dict(hand=TALL_STATE_VGPRS, vgprs=256)}


class RegisterContractError(RuntimeError):
    pass


def flat4_resource_usage() -> dict:
    """Compile csrc/wgnn_tiled.hip to gfx950 assembly and return, per ``agg_tiled_flat4`` instantiation, what the compiler
    did with the register file the kernel splits by hand: the ``-Rpass-analysis=kernel-resource-usage`` remarks, the code
    object metadata, and every compiler-emitted instruction (outside ``;;#ASMSTART`` .. ``;;#ASMEND``) that names a
    hand-owned VECTOR register (v32..v127: they carry state across statements; the literal scalar registers s80..s95 are
    statement-local and declared as clobbers, the compiler may use them in between)."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "wgnn_tiled.s"
        r = subprocess.run([hipcc(), *FLAGS, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                            str(PKG / "csrc" / "wgnn_tiled.hip"), "-o", str(out)], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-4000:])
        asm = out.read_text()
    kernels: dict = {}
    cur = None
    for line in r.stderr.splitlines():                       # remark blocks: "Function Name: <sym>" then one remark per figure
        m = re.search(r"remark: +(?:Function )?Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {"remarks": {}}) if any(k in m.group(1) for k in KERNELS) else None
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur is not None:
            cur["remarks"][m.group(1).strip()] = m.group(2)
    for name, rec in kernels.items():
        m = re.search(r"^\s*\.name:\s+%s\s*$" % re.escape(name), asm, re.M)
        if m is None or f"\n{name}:" not in asm:              # a hipcc whose assembly text this parser does not understand
            raise RegisterContractError(f"cannot audit {name}: its metadata / body was not found in the assembly")
        # metadata entries of one kernel sit in one YAML map: take the fields around .name
        blk_start = asm.rfind("  - .", 0, m.start())
        blk_end = asm.find("\n  - .", m.end())
        blk = asm[blk_start: blk_end if blk_end > 0 else len(asm)]
        rec["metadata"] = {k: int(v) for k, v in re.findall(r"\.(sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|"
                                                            r"vgpr_count|sgpr_count|agpr_count):\s+(\d+)", blk)}
        body_start = asm.index(f"\n{name}:")
        body = asm[body_start: asm.index(".Lfunc_end", body_start)]
        bad, in_asm = [], False
        hand = next(v["hand"] for k, v in KERNELS.items() if k in name)
        rec["want_vgprs"] = next(v["vgprs"] for k, v in KERNELS.items() if k in name)
        for ln in body.splitlines():
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif not in_asm and t and not t.startswith((";", ".", "//")):
                code = t.split(";")[0]
                regs = [("v", int(a), int(b or a)) for a, b in re.findall(r"\bv\[?(\d+)(?::(\d+))?\]?", code)]
                for kind, lo, hi in regs:
                    if hi >= hand.start and lo < hand.stop:
                        # reads of chunk / segment / accumulator registers that the SOURCE asks for are inside asm
                        # statements; anything here was emitted by the compiler on its own
                        bad.append(code)
                        break
        rec["compiler_touches_hand_registers"] = bad
    return kernels


def audit_flat4(usage: dict | None = None) -> dict:
    """Build-time guard of agg_tiled_flat4's hand-allocated register file: raises RegisterContractError when
    any instantiation spills, uses scratch, does not get exactly 128 VGPRs, or when the compiler itself touches a
    hand-owned register."""
    usage = flat4_resource_usage() if usage is None else usage
    for k in KERNELS:
        if sum(k in name for name in usage) < 6:
            raise RegisterContractError(f"expected 6 {k} instantiations, found {sorted(usage)}")
    for name, rec in usage.items():
        md, rm = rec["metadata"], rec["remarks"]
        problems = []
        if md.get("sgpr_spill_count", -1) != 0 or md.get("vgpr_spill_count", -1) != 0:
            problems.append(f"spills: {md}")
        if md.get("private_segment_fixed_size", -1) != 0:
            problems.append(f"scratch: {md}")
        want = rec.get("want_vgprs", 128)
        if md.get("vgpr_count") != want:
            problems.append(f"vgpr_count {md.get('vgpr_count')} != {want} ({512 // want} waves per SIMD)")
        if rm.get("SGPRs Spill") != "0" or rm.get("VGPRs Spill") != "0" or rm.get("ScratchSize") not in ("0", None) or \
                rm.get("VGPRs") != str(want):
            problems.append(f"resource remarks: {rm}")
        if rec["compiler_touches_hand_registers"]:
            problems.append("compiler-emitted instructions name hand-owned registers: "
                            + " | ".join(rec["compiler_touches_hand_registers"][:5]))
        if problems:
            raise RegisterContractError(f"{name}: " + "; ".join(problems))
    return usage


def build(force: bool = False, verbose: bool = False, audit: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    import runpy
    runpy.run_path(str(GEN))["main"](str(PKG / "csrc" / "wgnn_flat_asm.inc"))
    cmd = [hipcc(), *FLAGS, "-shared", "-fPIC", *map(str, SRC), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    if audit:                                                # the hand-split register file of agg_tiled_flat4 is a build-time contract
        try:
            audit_flat4()
        except Exception:                                    # a violated contract OR an audit that could not run (parse error on
            LIB.unlink(missing_ok=True)                      # another hipcc): never leave an unaudited library behind
            raise
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

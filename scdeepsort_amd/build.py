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
SRC = [PKG / "csrc" / "wgnn_kernels.hip", PKG / "csrc" / "wgnn_tiled.hip", PKG / "csrc" / "wgnn_linear.hip", PKG / "csrc" / "wgnn_sample.hip"]
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


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    import runpy
    runpy.run_path(str(GEN))["main"](str(PKG / "csrc" / "wgnn_flat_asm.inc"))
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-pass-failed", "-Wno-inline-asm", f"-I{ROOT / 'include'}", *map(str, SRC), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

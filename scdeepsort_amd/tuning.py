"""Per-shape selection of the library GEMM kernels behind the dense half of a layer (``NodeUpdate``'s Linear, the head:
reference models/gnn.py:18-25,66-67).

The projections are plain library GEMMs (rocBLAS / hipBLASLt through torch).  The kernel the libraries' own heuristics pick
for the shapes of this path - [1e5, 400] x [400, 256] and friends in fp32 - is not their fastest: PyTorch's TunableOp,
timing every candidate solution once per shape, finds 147 us against 232 us for the largest projection of BASELINE cfg3
(``scratch/tune_gemms.py``).  The picks for the shapes of the BASELINE configs (forward, sharded forward, training step)
are tracked in ``tuned_gemms_gfx950.csv`` next to this file; ``use_tuned_gemms()`` makes torch use them (selection only, no
tuning at run time).  The file carries the library versions it was tuned against; torch ignores it on any other stack and
falls back to the libraries' heuristics - there is nothing to break, only microseconds to lose.
"""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Optional

import torch

TUNED_FILE = Path(__file__).resolve().parent / "tuned_gemms_gfx950.csv"
_ACTIVE = False


def active() -> bool:
    """True once ``use_tuned_gemms`` has loaded a result file: the library GEMMs then run their tuned picks."""
    return _ACTIVE


def use_tuned_gemms(path: Optional[Path] = None) -> bool:
    """Select the tracked per-shape GEMM picks (TunableOp on, tuning off).  Returns True when a result file was loaded."""
    global _ACTIVE
    path = Path(path) if path is not None else TUNED_FILE
    if not torch.cuda.is_available() or not path.exists():
        return False
    try:                                             # any missing piece of the TunableOp API = library heuristics, nothing else
        import torch.cuda.tunable as T
        T.enable(True)
        T.tuning_enable(False)
        if hasattr(T, "record_untuned_enable"):
            T.record_untuned_enable(False)
        _ACTIVE = bool(T.read_file(str(path)))
        if not _ACTIVE:
            T.enable(False)
    except Exception:
        _ACTIVE = False
    return _ACTIVE


def tune_gemms(workload: Callable[[], None], path: Optional[Path] = None, max_ms_per_shape: int = 3000, iters: int = 50) -> Path:
    """Run ``workload`` once with TunableOp tuning every GEMM shape it meets and write the picks to ``path``."""
    import torch.cuda.tunable as T
    path = Path(path) if path is not None else TUNED_FILE
    T.enable(True)
    T.tuning_enable(True)
    T.set_max_tuning_duration(int(max_ms_per_shape))
    T.set_max_tuning_iterations(int(iters))
    T.set_filename(str(path))
    if path.exists():
        T.read_file(str(path))                       # keep earlier picks, add new shapes
    workload()
    torch.cuda.synchronize()
    T.tuning_enable(False)
    # the result file in TunableOp's own format (this torch has no write_file(); the C++ side only writes at exit)
    with open(path, "w") as f:
        for name, value in T.get_validators():
            f.write(f"Validator,{name},{value}\n")
        for op, params, solution, t in T.get_results():
            f.write(f"{op},{params},{solution},{t}\n")
    return path

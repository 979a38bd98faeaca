"""ctypes binding of the C ABI declared in ``include/wgnn.h``.

The product path has NO CPU fallback: if ``libwgnn_hip.so`` is missing the
import of :func:`lib` raises, and every wrapper raises on a non-zero return.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libwgnn_hip.so"

# error codes / enums (mirror include/wgnn.h)
SRC_IS_GENE, DST_IS_GENE, NO_ALPHA = 0, 1, 2
F32, F16 = 0, 1
PLAN_TALL = 1 << 16                # tile-plan geometry bit OR-ed into `block_rows` (include/wgnn.h WGNN_PLAN_TALL, 0.2.4)
FLAG_RELU, FLAG_NO_MEAN, FLAG_NO_SELF, FLAG_SELF_COMPACT, FLAG_ROWPTR_I64, FLAG_SRC_PRESCALED, FLAG_OUT_SCALE_ALPHA = 1, 2, 4, 8, 16, 32, 64
ABI_MAJOR = 2                      # include/wgnn.h WGNN_VERSION / 100
ABI_MIN = 206                      # 0.2.1: shared-pair marks in tile-plan entries; 0.2.2: WGNN_FLAG_OUT_SCALE_ALPHA (gnn.GNN sets it); 0.2.3: fused training glue; 0.2.5: even-padded plan segments

_vp, _i32, _i64, _u32, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_int

SIGNATURES = {
    "wgnn_version": (C.c_int, []),
    "wgnn_last_error_string": (C.c_char_p, [C.c_int]),
    "wgnn_agg_workspace_bytes": (C.c_int, [_i64, _i64, _i32, _int, _int, _vp, _vp]),
    "wgnn_plan_build_host": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "wgnn_plan_build_host_i64": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "wgnn_agg_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _int, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                               _vp, _i64, _vp, _i64, _i32, _int, _int, _u32,
                               _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "wgnn_agg_fwd_tiled": (C.c_int, [_vp, _vp, _int, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp,
                                     _vp, _i64, _vp, _i64, _i32, _u32, _vp, _vp, _i32, _i32, _vp, _vp, _i64,
                                     _vp, _i64, _vp, _i64, _vp]),
    "wgnn_agg_bwd_src_tiled": (C.c_int, [_vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _int, _i64, _i32,
                                         _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "wgnn_agg_bwd_alpha_tiled": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32,
                                           _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "wgnn_agg_bwd_src": (C.c_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, _i64, _vp, _i64,
                                   _vp, _i64, _vp, _int, _i64, _i32,
                                   _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "wgnn_agg_bwd_alpha": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                     _i64, _i32, _u32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "wgnn_normalize_rows": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "wgnn_normalize_rows_i64": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "wgnn_sample_rows": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, C.c_uint64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wgnn_linear_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _u32, _vp]),
    "wgnn_linear_fwd_ex": (C.c_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _u32, _vp]),
    "wgnn_linear_wgrad_workspace": (C.c_int, [_i64, _i32, _i32, _vp, _vp]),
    "wgnn_linear_wgrad": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _int, _vp, _i64, _vp]),
    "wgnn_agg_bwd_prepare_workspace": (C.c_int, [_i64, _i32, _vp]),
    "wgnn_agg_bwd_prepare": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _int, _i32, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp,
                                       _i64, _i32, _vp, _i64, _vp]),
    "wgnn_ce_sum_workspace": (C.c_int, [_i64, _vp]),
    "wgnn_ce_sum_fwd_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _i64, _vp]),
    "wgnn_tile_plan_count": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "wgnn_csr_transpose_workspace": (C.c_int, [_i64, _i32, _vp, _vp]),
    "wgnn_csr_transpose_count": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "wgnn_csr_transpose_fill": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "wgnn_tile_plan_fill": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "wgnn_agg_linear_relu_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _int, _i32, _vp, _i64, _vp, _i64, _vp, _vp,
                                           _i64, _i32, _u32, _vp, _i64, _vp, _i64, _vp, _i64, _vp,
                                           _vp, _i64, _vp, _i32, _u32, _vp, _i64, _vp]),
}


class WgnnError(RuntimeError):
    pass


@lru_cache(maxsize=1)
def lib() -> C.CDLL:
    if not LIB_PATH.exists():
        raise WgnnError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m scdeepsort_amd.build` or `__graft_entry__.build()`.")
    dll = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(dll, name)          # AttributeError here = header/ABI drift
        fn.restype, fn.argtypes = res, args
    if dll.wgnn_version() // 100 != ABI_MAJOR or dll.wgnn_version() < ABI_MIN:
        raise WgnnError(f"{LIB_PATH} speaks ABI {dll.wgnn_version()}, this binding needs major version {ABI_MAJOR}, "
                        f"at least {ABI_MIN}: rebuild it")
    return dll


def call(device, name: str, *args) -> int:
    """Invoke a kernel-launching entry point with ``device`` current: the library launches on the stream it is given and
    keeps per-device kernel attributes keyed by hipGetDevice(), so the caller's current device must be the one that owns
    the tensors / stream (matters as soon as a process drives more than one GPU or gpu_id != 0)."""
    import torch
    with torch.cuda.device(device):
        return getattr(lib(), name)(*args)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().wgnn_last_error_string(rc).decode()
        raise WgnnError(f"{what} failed: {msg} (code {rc})")

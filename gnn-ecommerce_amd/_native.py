"""ctypes binding of ``csrc/liblgconv_hip.so`` (C ABI: ``include/lgconv_hip.h``).

There is no fallback: if the library is missing or fails to load, every operator in this
package raises ``NativeLibraryError``.  The library is opened only AFTER ``import torch`` so
that its ``NEEDED libamdhip64.so.7`` binds to the HIP runtime torch has already mapped (one
runtime per process); ``runtime_libraries()`` lets tests assert that.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import CFUNCTYPE, POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p
from typing import Optional

import torch

LIB_NAME = "liblgconv_hip.so"
# LGCN_LIB_PATH selects another build of the SAME library (A/B kernel experiments); never a fallback.
LIB_PATH = os.environ.get("LGCN_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)
ABI_VERSION = 12

# status bits (include/lgconv_hip.h)
ST_INDEX_OOB = 1
ST_SAMPLER_EXHAUSTED = 2

class SweepCfg(Structure):
    """lgc_sweep_cfg"""
    _fields_ = [("n_bands", c_int32), ("waves_per_band_round", c_int32), ("row_cap", c_int32), ("piece_cap", c_int32),
                ("lookahead", c_int32), ("groups", c_int32), ("round_order", c_int32)]


class SweepDims(Structure):
    """lgc_sweep_dims"""
    _fields_ = [("n_bands", c_int32), ("rounds", c_int32), ("row_cap", c_int32), ("piece_cap", c_int32),
                ("n_rows", c_int32), ("groups", c_int32),
                ("n_waves", c_int64), ("n_slabs", c_int64), ("n_slots", c_int64), ("n_entries", c_int64),
                ("n_steps", c_int64), ("n_padding", c_int64)]


class TileClassC(Structure):
    """lgc_tile_class"""
    _fields_ = [("order", c_void_p), ("meta", c_void_p), ("slab", c_void_p), ("n_tiles", c_int32), ("width", c_int32)]


class SweepArraysC(Structure):
    """lgc_sweep_arrays"""
    _fields_ = [("slabs", c_void_p), ("wave_slab_ptr", c_void_p), ("wave_npieces", c_void_p), ("piece_slot", c_void_p),
                ("multi", c_void_p), ("multi_wide", c_void_p), ("partials", c_void_p), ("n_waves", c_int64),
                ("row_cap", c_int32), ("n_rows", c_int32), ("n_wide", c_int32), ("groups", c_int32)]


class OperatorC(Structure):
    """lgc_operator"""
    _fields_ = [("rowptr", c_void_p), ("entries", c_void_p), ("chunks", c_void_p), ("multi", c_void_p),
                ("partials", c_void_p), ("sweep", POINTER(SweepArraysC)), ("tiles", TileClassC * 3),
                ("row_begin", c_int32), ("row_end", c_int32), ("short_max", c_int32), ("n_chunks", c_int32),
                ("n_multi", c_int32), ("n_tile_classes", c_int32), ("tiles_per_wave", c_int32), ("reserved", c_int32)]


EXCHANGE_FN = CFUNCTYPE(c_int, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p)   # lgc_exchange_fn


# symbol -> (restype, argtypes); tests check every name against the header and the .so
SIGNATURES = {
    "lgc_abi_version": (c_int, []),
    "lgc_error_string": (c_char_p, [c_int]),
    "lgc_dim_ok": (c_int, [c_int32]),
    "lgc_build_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "lgc_build_csr": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lgc_spmm": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                         c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_float, c_float,
                         c_int32, c_void_p]),
    "lgc_spmm_rows": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                              c_void_p, c_int64, c_float, c_float, c_int32, c_void_p]),
    "lgc_spmm_rows_split": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                    c_int64, c_int64, c_void_p, c_int64, c_float, c_float, c_int32, c_int32, c_void_p, c_void_p,
                                    c_int64, c_void_p]),
    "lgc_bipartite_split": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "lgc_row_plan_workspace_bytes": (c_size_t, [c_int64]),
    "lgc_row_plan_count": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lgc_row_plan_fill": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lgc_tile_classes_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "lgc_tile_classes": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_void_p, c_size_t, c_void_p,
                                 c_void_p, c_void_p]),
    "lgc_tile_pack": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "lgc_build_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "lgc_spmm_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_void_p, c_int64,
                               c_void_p, c_int64, c_void_p, c_int64, c_float, c_float, c_int32, c_void_p]),
    "lgc_sweep_plan_create": (c_void_p, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(SweepCfg),
                                         POINTER(c_int)]),
    "lgc_sweep_plan_dims": (c_int, [c_void_p, POINTER(SweepDims)]),
    "lgc_sweep_plan_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lgc_sweep_plan_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lgc_sweep_plan_export_multi": (c_int, [c_void_p, c_void_p]),
    "lgc_sweep_plan_free": (None, [c_void_p]),
    "lgc_sweep_dplan_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, POINTER(SweepCfg)]),
    "lgc_sweep_dplan_create": (c_void_p, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int64, c_int32, c_int32,
                                          POINTER(SweepCfg), c_void_p, c_size_t, c_void_p, POINTER(c_int)]),
    "lgc_sweep_dplan_dims": (c_int, [c_void_p, POINTER(SweepDims)]),
    "lgc_sweep_dplan_fill": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lgc_sweep_dplan_export_multi": (c_int, [c_void_p, c_void_p]),
    "lgc_sweep_dplan_free": (None, [c_void_p]),
    "lgc_sweep_ok": (c_int, [c_int32, c_int64, c_int64]),
    "lgc_spmm_sweep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p,
                               c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_float, c_float,
                               c_int32, c_void_p]),
    "lgc_apply": (c_int, [POINTER(OperatorC), c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_float, c_float,
                          c_int32, c_void_p]),
    "lgc_hop_exchange": (c_int, [POINTER(OperatorC), POINTER(OperatorC), c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                 c_int64, c_float, c_float, c_int32, c_int32, c_int32, c_int32, EXCHANGE_FN, c_void_p, c_void_p]),
    "lgc_segment_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_int64, c_int64, c_int32,
                                c_int32, c_void_p]),
    "lgc_pair_dot_rows": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "lgc_reg_rows": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_float,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "lgc_bpr_loss": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "lgc_pair_seed_vals": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "lgc_seed_prepare": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "lgc_seed_flags": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p]),
    "lgc_seed_pull": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    "lgc_seed_mark": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p]),
    "lgc_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                              c_float, c_void_p]),
    "lgc_adam_step_hp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "lgc_lincomb": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "lgc_mask_topk": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                              c_void_p, c_void_p, c_void_p]),
    "lgc_sample_triples": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_uint64,
                                   c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lgc_pair_dot": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                             c_void_p, c_void_p]),
}


class NativeLibraryError(RuntimeError):
    """The HIP library is missing, stale, or a call into it failed."""


_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the MI355X propagation library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gnn-ecommerce_amd/csrc`). "
            "There is no CPU or PyTorch fallback for this path.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype, fn.argtypes = restype, argtypes
    if lib.lgc_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH} has ABI {lib.lgc_abi_version()}, expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        text = load().lgc_error_string(code).decode()
        raise NativeLibraryError(f"{what} failed: {text} (code {code})")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw device pointer, or None (-> NULL)."""
    return None if t is None else t.data_ptr()


def stream_of(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise NativeLibraryError(
            f"{name} is on {t.device}: the LightGCN propagation path runs only on a ROCm device "
            "(MI355X); move the model and graph tensors to 'cuda'. No CPU fallback is provided.")


MAX_TERMS = 8
SEED_MAX = 8192        # LGC_SEED_MAX: ids one lgc_seed_prepare launch sorts


def lincomb(y: torch.Tensor, terms) -> torch.Tensor:
    """y = sum_t coef_t * src_t over 2-D fp32 views of equal shape (``terms``: list of (coef, tensor))."""
    lib = load()
    n = len(terms)
    if not 1 <= n <= MAX_TERMS:
        raise ValueError(f"1..{MAX_TERMS} terms")
    ptrs = (c_void_p * n)(*[t.data_ptr() for _, t in terms])
    strides = (c_int64 * n)(*[t.stride(0) for _, t in terms])
    coefs = (c_float * n)(*[float(c) for c, _ in terms])
    with torch.cuda.device(y.device):
        code = lib.lgc_lincomb(y.data_ptr(), y.stride(0), ptrs, strides, coefs, n, y.size(0), y.size(1),
                               stream_of(y.device))
    check(code, "lgc_lincomb")
    return y


def runtime_libraries() -> list:
    """Paths of every libamdhip64 / libhsa-runtime64 mapped into this process."""
    seen = set()
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.split()[-1]
            if "libamdhip64" in path or "libhsa-runtime64" in path:
                seen.add(path)
    return sorted(seen)

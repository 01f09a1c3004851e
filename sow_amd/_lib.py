"""ctypes binding of libsow_amd.so (the C ABI declared in include/sow_amd.h).

The library is the product: there is NO CPU or PyTorch fallback behind these
calls.  If the shared object is missing, or a call returns a non-zero code, a
RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOW_AMD_LIB: an alternative build of the same library (the `make STAMPS=1` timeline build used by tools/chain_stamps.py)
LIB_PATH = os.environ.get("SOW_AMD_LIB") or os.path.join(_HERE, "lib", "libsow_amd.so")

F32, BF16 = 0, 1
ACC_NONE, ACC_LOWRANK, ACC_DENSE = 0, 1, 2
H_COLS = 64
BWD_DATA, BWD_WEIGHTS, BWD_WEIGHTS_PARTIAL, BWD_WEIGHTS_REDUCE = 1, 2, 4, 8
BWD_GROUP_SLABS = 16     # sow_backward_group: slab counts planned over the group (deferred reduction from group descriptors)



class LayerArgs(ctypes.Structure):
    """sow_layer_args of include/sow_amd.h (field order and types one to one)."""
    _fields_ = [("x", c_void_p), ("A", c_void_p), ("B", c_void_p), ("acc_down", c_void_p), ("acc_up", c_void_p),
                ("bias", c_void_p), ("y", c_void_p), ("h_save", c_void_p), ("dy", c_void_p), ("dx", c_void_p),
                ("dA", c_void_p), ("dB", c_void_p), ("dbias", c_void_p), ("T", c_int64), ("d_in", ctypes.c_int32),
                ("d_out", ctypes.c_int32), ("r_live", ctypes.c_int32), ("r_acc", ctypes.c_int32), ("acc_kind", ctypes.c_int32),
                ("scale", c_float), ("grad_beta", c_float), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class AccumulateArgs(ctypes.Structure):
    """sow_accumulate_args of include/sow_amd.h."""
    _fields_ = [("acc", c_void_p), ("A", c_void_p), ("B", c_void_p), ("draw", c_void_p), ("ld_draw", c_int64),
                ("A_new", c_void_p), ("zero", c_void_p), ("zero_bytes", c_int64), ("d_in", ctypes.c_int32),
                ("d_out", ctypes.c_int32), ("r", ctypes.c_int32), ("r_new", ctypes.c_int32), ("draw_cols", ctypes.c_int32),
                ("scale", c_float), ("acc_beta", c_float), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


TT_MAX_ORDER = 6


class TtDesc(ctypes.Structure):
    """sow_tt_desc of include/sow_amd.h."""
    _fields_ = [("cores", c_void_p * TT_MAX_ORDER), ("order", ctypes.c_int32), ("ranks", ctypes.c_int32 * (TT_MAX_ORDER + 1)),
                ("in_dims", ctypes.c_int32 * TT_MAX_ORDER), ("out_dims", ctypes.c_int32 * TT_MAX_ORDER),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32)]


class TtAdamItem(ctypes.Structure):
    """sow_ttadam_item of include/sow_amd.h."""
    _fields_ = [("m", TtDesc), ("v", TtDesc), ("param", c_void_p), ("grad", c_void_p), ("ld_param", c_int64),
                ("ld_grad", c_int64), ("step_size", c_float), ("lr_times_wd", c_float), ("has_state", ctypes.c_int32),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


# name -> (restype, argtypes); mirrors include/sow_amd.h one to one
SIGNATURES = {
    "sow_version": (c_int, []),
    "sow_error_string": (c_char_p, [c_int]),
    "sow_set_switch": (c_int, [c_char_p, c_int]),
    "sow_get_switch": (c_int, [c_char_p]),
    "sow_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sow_forward_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sow_h_save_elems": (c_size_t, [c_int64, c_int]),
    "sow_forward": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p,
                                               c_size_t, c_void_p]),
    "sow_backward": (c_int, [c_void_p] * 11 + [c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                                                 c_void_p, c_size_t, c_void_p]),
    "sow_backward_ex": (c_int, [c_void_p] * 11 + [c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                                    c_void_p, c_size_t, c_int, c_void_p]),
    "sow_forward_group": (c_int, [POINTER(LayerArgs), c_int, c_int, c_void_p]),
    "sow_backward_group": (c_int, [POINTER(LayerArgs), c_int, c_int, c_int, c_void_p]),
    "sow_backward_group_reduce_desc": (c_int, [POINTER(LayerArgs), c_int, c_int, c_int, c_void_p, POINTER(c_int)]),
    "sow_backward_group_plan": (c_int, [POINTER(LayerArgs), c_int, c_int, c_int, POINTER(c_int)]),
    "sow_accumulate_batch": (c_int, [POINTER(AccumulateArgs), c_int, c_int, c_void_p]),
    "sow_reduce_desc_bytes": (c_size_t, []),
    "sow_backward_reduce_desc": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                         c_void_p, c_size_t, c_void_p, c_void_p]),
    "sow_reduce_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sow_gemm": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64,
                         c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "sow_gemm_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "sow_gemm_ex": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64,
                            c_int, c_int, c_float, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "sow_qr_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "sow_qr_thin": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int,
                            c_void_p, c_size_t, c_void_p]),
    "sow_zero_state": (c_int, [POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p]),
    "sow_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                               c_float, c_int, c_float, c_int, c_int, c_void_p]),
    "sow_ttadam_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                 c_float, c_int, c_void_p]),
    "sow_tt_kron_core": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sow_tt_decompose_workspace_bytes": (c_size_t, [POINTER(TtDesc)]),
    "sow_ttadam_workspace_bytes": (c_size_t, [POINTER(TtDesc)]),
    "sow_tt_reconstruct_batch": (c_int, [POINTER(TtDesc), POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p]),
    "sow_tt_decompose_batch": (c_int, [POINTER(TtDesc), POINTER(c_void_p), POINTER(c_int64), c_int, POINTER(c_void_p),
                                       POINTER(c_size_t), c_void_p]),
    "sow_ttadam_batch": (c_int, [POINTER(TtAdamItem), c_int, c_float, c_float, c_float, c_void_p]),
    "sow_absmax": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sow_small_inverse": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sow_axpby": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_int, c_void_p]),
    "sow_cast_copy": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p]),
}

_lib = None


class SowLibraryError(RuntimeError):
    pass


def load():
    """Load libsow_amd.so (once).  Raises SowLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SowLibraryError(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). "
            "sow_amd has no CPU/PyTorch fallback for the SoW hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().sow_error_string(code).decode()
        raise SowLibraryError(f"libsow_amd {what} failed with code {code}: {msg}")


class switch:
    """Context manager over sow_set_switch for tests and A/B tools: `with _lib.switch(NO_FUSED_H=1): ...` sets the
    kernel-selection switch explicitly (no environment, no getenv on the launch path) and restores it afterwards."""

    def __init__(self, **values):
        self.values = values
        self.old = {}

    @staticmethod
    def _forget_sizes():
        # workspace sizes depend on the kernel choice (slab counts): drop the memoised queries of sow_amd.ops
        from . import ops
        ops._WS_BYTES.clear()
        ops._FWD_WS_BYTES.clear()

    def __enter__(self):
        lib = load()
        for k, v in self.values.items():
            self.old[k] = lib.sow_get_switch(k.encode())
            check(lib.sow_set_switch(k.encode(), int(v)), f"sow_set_switch({k})")
        self._forget_sizes()
        return self

    def __exit__(self, *exc):
        lib = load()
        for k, v in self.old.items():
            lib.sow_set_switch(k.encode(), v)
        self._forget_sizes()
        return False

"""ctypes binding of libskp_hip.so (the C-ABI declared in include/skp.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing the
native ops raises.  The product path never routes through `oracle/` or through eager PyTorch
re-implementations of these kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: libskp_hip.so has to bind to the HIP runtime
#                              that PyTorch-ROCm already loaded (one runtime per process, shared streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SKP_LIB_PATH: another BUILD of the same library (same-box A/B of two kernel versions, tools/ab_build.py); never a fallback
LIB_PATH = os.environ.get("SKP_LIB_PATH") or os.path.join(_HERE, "csrc", "libskp_hip.so")
ABI_VERSION = 39

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes   (restype is always int)
SIGNATURES = {
    "skp_abi_version": [],
    "skp_tune_set": [C.c_char_p, _i],
    "skp_tune_get": [C.c_char_p],
    "skp_group_norm_onepass_ok": [_i, _i, _i, _i],
    "skp_gemm_nt_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i,
                        _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp],
    "skp_qk_logits_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "skp_attn_map_fwd_f32": [C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "skp_attn_map_fwd_ex_f32": [C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i64, _i, _vp],
    "skp_attn_map_bwd_ex_f32": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                                _i, _i64, _i, _vp],
    "skp_attn_map_bwd_workspace": [C.POINTER(_i), _i, _i, _i, _i, _i],
    "skp_attn_map_bwd_f32": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "skp_group_norm_coef_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _f, _vp],
    "skp_conv3x3_f4_gn_ok": [_i, _i, _i, _i, _i],
    "skp_conv3x3_f4_gn_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "skp_unwarp_accumulate_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "skp_attn_map_fwd_wide_f32": [C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp],
    "skp_attn_map_fwd_wide_ok": [C.POINTER(_i), _i, _i, _i],
    "skp_attn_map_bwd_col_ok": [C.POINTER(_i), _i, _i, _i, _i, _i],
    "skp_attn_map_bwd_col_workspace": [C.POINTER(_i), _i, _i, _i, _i, _i, _i],
    "skp_attn_map_bwd_col_f32": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp,
                                 _i, _vp],
    "skp_attn_map_bwd_sparse_workspace": [C.POINTER(_i), _i, _i, _i, _i, _i, _i],
    "skp_attn_map_bwd_sparse_f32": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp,
                                    _i, _vp],
    "skp_token_stats_f32": [_vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    "skp_select_tokens": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "skp_select_tokens_batched": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "skp_losses_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _f, C.POINTER(_f), _vp, _vp, _vp, _vp, _vp],
    "skp_losses_fwd_dev_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "skp_rows_axpy_f32": [_vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp],
    "skp_cross_attn_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "skp_cross_attn_bwd_workspace": [_i, _i, _i, _i, _i],
    "skp_cross_attn_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "skp_group_norm_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "skp_group_norm_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "skp_group_norm_bwd_add_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "skp_add_bias_residual_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "skp_add_layer_norm_ok": [_i],
    "skp_add_layer_norm_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp],
    "skp_add_layer_norm_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "skp_conv3x3_filter_f32": [_vp, _vp, _i, _i, _i, _vp],
    "skp_conv3x3_workspace": [_i, _i, _i, _i, _i, _i],
    "skp_conv3x3_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_f4_filter_f32": [_vp, _vp, _i, _i, _i, _vp],
    "skp_conv3x3_f4_workspace": [_i, _i, _i, _i, _i],
    "skp_conv3x3_f4_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "skp_flash_attn_fwd_split_ok": [_i, _i, _i, _i, _i, _i],
    "skp_flash_attn_fwd_split_workspace": [_i, _i, _i, _i, _i, _i],
    "skp_flash_attn_fwd_split_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_float, _vp],
    "skp_flash_attn_bwd_split_ok": [_i, _i, _i, _i, _i, _i],
    "skp_flash_attn_bwd_split_workspace": [_i, _i, _i, _i, _i, _i],
    "skp_flash_attn_bwd_split_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "skp_flash_attn_bwd_split_ld_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "skp_conv3x3_f4r_ok": [_i, _i, _i, _i, _i],
    "skp_conv3x3_f4r_filter_f32": [_vp, _vp, _i, _i, _i, _vp],
    "skp_conv3x3_f4r_workspace": [_i, _i, _i, _i, _i],
    "skp_conv3x3_f4r_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_s2_filter_f32": [_vp, _vp, _i, _i, _vp],
    "skp_conv3x3_s2_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_s2_workspace": [_i, _i, _i, _i, _i],
    "skp_conv3x3_s2_ws_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_small_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_small_stats_blocks": [_i, _i, _i, _i, _i],
    "skp_conv3x3_small_stats_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_f4_stats_blocks": [_i, _i, _i, _i, _i],
    "skp_conv3x3_f4_stats_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "skp_conv3x3_s2_stats_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "skp_group_norm_fwd_blocks_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "skp_geglu_fwd_f32": [_vp, _vp, _i64, _i, _vp],
    "skp_geglu_bwd_f32": [_vp, _vp, _vp, _i64, _i, _vp],
    "skp_nchw_to_tokens_f32": [_vp, _vp, _i, _i, _i, _vp],
    "skp_tokens_to_nchw_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "skp_flash_attn_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "skp_flash_attn_bwd_workspace": [_i, _i, _i, _i, _i, _i],
    "skp_flash_attn_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "skp_flash_attn_bwd_ld_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "skp_self_attn_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "skp_self_attn_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
}

_ERR = {-1: "SKP_E_BADARG (null pointer / non-positive size)",
        -2: "SKP_E_RANGE (size outside the built kernel range)",
        -3: "SKP_E_LDS (tile exceeds 160 KiB LDS)"}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the CDLL; raises NativeLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C stablekeypoints_amd/csrc`). There is no CPU/eager fallback for the hot path.")
    l = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        if not hasattr(l, name):
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}")
        fn = getattr(l, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name.endswith("_workspace") else C.c_int
    v = l.skp_abi_version()
    if v != ABI_VERSION:
        raise NativeLibraryError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
    _lib = l
    return l


# measurement aids / experiments live in their own library (tools/csrc -> libskp_lab.so, tools/csrc/skp_lab.h); the product
# library does not contain them
# (SKP_LAB_PATH overrides the in-checkout location, e.g. when the package is installed outside the repository tree)
LAB_PATH = os.environ.get("SKP_LAB_PATH") or os.path.join(os.path.dirname(_HERE), "tools", "csrc", "libskp_lab.so")
LAB_SIGNATURES = {
    "skp_probe_mfma_f32": [_i, _i, _vp, _vp, _vp],
}
_lab = None


def lab():
    """The measurement / experiment library (bench.py's issue-rate probe, the split-bf16 GEMM experiment)."""
    global _lab
    if _lab is None:
        if not os.path.exists(LAB_PATH):
            raise NativeLibraryError(f"{LAB_PATH} is missing: `make -C tools/csrc` (measurement aids, not the product library)")
        l = C.CDLL(LAB_PATH)
        for name, argtypes in LAB_SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes, fn.restype = argtypes, C.c_int
        _lab = l
    return _lab


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERR.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP launch failed with hipError_t={rc}")


def ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) for p in ptrs])
    return C.cast(arr, C.POINTER(C.c_void_p)), arr


def int_array(vals):
    arr = (C.c_int * len(vals))(*[int(v) for v in vals])
    return C.cast(arr, C.POINTER(C.c_int)), arr


def float_array(vals):
    arr = (C.c_float * len(vals))(*[float(v) for v in vals])
    return C.cast(arr, C.POINTER(C.c_float)), arr


def tune(key: str, value: int) -> None:
    """Developer override of a launch plan (tests / tools; include/skp.h: skp_tune_set).  0 restores the library's choice."""
    check(lib().skp_tune_set(key.encode(), int(value)), f"skp_tune_set({key})")

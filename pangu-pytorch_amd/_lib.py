"""ctypes binding of libpangu_hip.so (the C ABI declared in include/pangu_hip.h).

The library handle lives in this module (never on an nn.Module), so models stay picklable /
deep-copyable (reference models/pangu_sample.py:162-164 deep-copies and pickles the whole model).
There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# PANGU_HIP_LIB: development override (tools/ A/B builds of one kernel file linked into a second library); the product loads
# the in-tree library next to this file
LIB_PATH = os.environ.get("PANGU_HIP_LIB") or os.path.join(_HERE, "libpangu_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "pangu_hip.h")

_c = ctypes
_P, _I, _F = _c.c_void_p, _c.c_int, _c.c_float

# name -> argtypes; restype is int unless listed in _RESTYPES. Mirrors include/pangu_hip.h one to one
# (tests/test_abi.py checks the header's declarations against this table and the built library).
SIGNATURES = {
    "pangu_abi_version": [],
    "pangu_error_string": [_I],
    "pangu_window_index_export": [_P, _P, _I, _I, _I, _I],
    "pangu_window_mask_export": [_P, _P, _I, _I, _I],
    "pangu_linear_fwd": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pangu_linear_wgrad": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I],
    "pangu_linear_wgrad_ws": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _c.c_longlong],
    "pangu_window_attn_bwd": [_P] * 10 + [_I] * 6,
    "pangu_ln_residual_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _F],
    "pangu_downsample_ln_bwd": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pangu_upsample_ln_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "pangu_patch_recover_gather_bwd": [_P, _P, _P, _P, _P, _I, _I],
    "pangu_lat_weighted_sums": [_P, _P, _P, _P, _P, _I, _I, _I],
    "pangu_linear_fwd_bf16": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I],
    "pangu_window_attn_fwd_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "pangu_window_attn_qkv_fwd_bf16": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "pangu_ln_residual_fwd_bf16": [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _F],
    "pangu_linear_ln_residual_fwd_bf16": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "pangu_mlp_ln_residual_fwd_bf16": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F],
    "pangu_mlp_ln_residual_train_fwd_bf16": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _F],
    "pangu_linear_gelu_bwd_bf16": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P],
    "pangu_downsample_ln_fwd_bf16": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I],
    "pangu_upsample_ln_fwd_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "pangu_patch_embed_gather_bf16": [_P] * 11 + [_I, _I, _I],
    "pangu_linear_wgrad_bf16": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I],
    "pangu_linear_wgrad_bf16_ws": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _c.c_longlong],
    "pangu_shadow_refresh_bf16": [_P, _P, _I, _c.c_longlong],
    "pangu_adam_step_multi": [_P, _P, _I, _c.c_longlong] + [_c.c_double] * 5 + [_F, _F],
    "pangu_weighted_l1_loss_blocks": [_I, _I, _c.c_longlong, _I, _c.c_longlong, _I],
    "pangu_weighted_l1_loss_fwd": [_P] * 9 + [_I, _I, _c.c_longlong, _I, _c.c_longlong, _I, _I] + [_P] * 4,
    "pangu_weighted_l1_loss_bwd": [_P] * 10 + [_I, _I, _c.c_longlong, _I, _c.c_longlong, _I, _I] + [_P] * 4,
    "pangu_host_copy": [_P, _P, _c.c_longlong, _I],
    "pangu_window_attn_bwd_bf16": [_P] * 10 + [_I] * 6,
    "pangu_ln_residual_bwd_bf16": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _F],
    "pangu_downsample_ln_bwd_bf16": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pangu_upsample_ln_bwd_bf16": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "pangu_patch_recover_gather_bwd_bf16": [_P, _P, _P, _P, _P, _I, _I],
    "pangu_window_attn_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "pangu_window_attn_fwd_compact": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "pangu_attn_windows_fwd": [_P, _P, _P, _P, _c.c_longlong, _P, _I, _I, _I, _I],
    "pangu_attn_windows_bwd": [_P, _P, _P, _P, _c.c_longlong, _P, _P, _P, _I, _I, _I, _I],
    "pangu_ln_residual_fwd": [_P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _F],
    "pangu_linear_ln_residual_fwd": [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _F],
    "pangu_downsample_ln_fwd": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I],
    "pangu_upsample_ln_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "pangu_patch_embed_gather": [_P] * 11 + [_I, _I, _I],
    "pangu_patch_embed_gather_bwd": [_P] * 7 + [_I, _I, _I],
    "pangu_patch_recover_scatter": [_P, _P, _P, _P, _P, _I, _I],
    "pangu_patch_recover_scatter_denorm": [_P] * 11 + [_I, _I],
    "pangu_traffic_copy": [_P, _P, _P, _c.c_longlong, _I],
}
_RESTYPES = {"pangu_error_string": _c.c_char_p, "pangu_weighted_l1_loss_blocks": _c.c_longlong}

_lib = None


def header_functions():
    """Function names declared in include/pangu_hip.h."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pangu_[a-z0-9_]+)\s*\(", src)))


def load():
    """Load the HIP library once; raises RuntimeError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C pangu-pytorch_amd/csrc`). There is no CPU fallback for this path.")
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so both share one HIP runtime)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _I)
    if lib.pangu_abi_version() != 1:
        raise RuntimeError("libpangu_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pangu_error_string(rc)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc} (code {rc})")

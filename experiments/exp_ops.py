"""Python bindings of experiments/libpangu_experiments.so: the two kernels that were built, are parity-green and LOST their A/B
(round 5: profiles/r05_walk_attn_ab.md, profiles/r05_mlp_f32_ab.md).  Not imported by the product; experiments/test_experiments.py
and experiments/tools use them."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import pangu_pytorch_amd as P   # noqa: E402
from pangu_pytorch_amd import _lib, ops   # noqa: E402
from pangu_pytorch_amd import ops_bf16 as ob   # noqa: E402

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIG = {
    "pangu_window_attn_qkv_walk_fwd_bf16": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "pangu_mlp_ln_residual_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F],
}
_handle = None


def load():
    global _handle
    if _handle is None:
        # PANGU_EXP_LIB: an ablation build of this library (experiments/tools/ablate_mlp_f32.sh)
        path = os.environ.get("PANGU_EXP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpangu_experiments.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: run `make -C experiments`")
        _lib.load()
        lib = ctypes.CDLL(path)
        for name, argtypes in _SIG.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, _I
        _handle = lib
    return _handle


def window_attention_qkv_walk(x, w_qkv, b_qkv, esb, Z, H, W, heads, shifted, variant, want_lse=False):
    """The longitude-walking form of ops_bf16.window_attention_qkv (csrc/attn_walk_bf16.hip); variant = 10 * pipelines + bias mode:
    40, 30, 20 (bias rows re-read from L2 per window), 21, 11 (resident in registers)."""
    lib = load()
    xp, ldx = ob._rows(x, "attn_qkv.x")
    N, C = x.shape
    out = torch.empty((N, C), dtype=torch.bfloat16, device=x.device)
    lse = torch.empty((N, heads), dtype=torch.float32, device=x.device) if want_lse else None
    _lib.check(lib.pangu_window_attn_qkv_walk_fwd_bf16(
        ops._stream(x), xp, ldx, ob._p(w_qkv, "w_qkv"), ob._p(b_qkv, "b_qkv", torch.float32), ob._p(esb, "esb"), out.data_ptr(),
        lse.data_ptr() if want_lse else None, Z, H, W, C, heads, int(shifted), int(variant)), "window_attn_qkv_walk_fwd_bf16")
    return (out, lse) if want_lse else out


def mlp_ln_residual_f32(x, w1, b1, w2, b2, gamma, beta, out=None, branch_scale=1.0):
    """out = x + branch_scale * (LayerNorm(GELU(x @ w1^T + b1) @ w2^T + b2) * gamma + beta) in ONE fp32 launch, C = 192
    (csrc/mlp_fused_f32.hip): the (M, 4C) hidden activation never reaches memory.  x / out may be row-strided."""
    lib = load()
    xp, ldx = ops._rows(x, "mlp_ln.x")
    M, C = x.shape
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=x.device)
    op, ldo = ops._rows(out, "mlp_ln.out")
    chunks = ops._row_chunks(M, 4 * ldx, 4 * ldo)
    if chunks is not None:
        for m0, m1 in chunks:
            mlp_ln_residual_f32(x[m0:m1], w1, b1, w2, b2, gamma, beta, out[m0:m1], branch_scale)
        return out
    c = ops._chk
    _lib.check(lib.pangu_mlp_ln_residual_fwd(ops._stream(x), xp, ldx, c(w1, "w1"), c(b1, "b1"), c(w2, "w2"), c(b2, "b2"),
                                             c(gamma, "gamma"), c(beta, "beta"), op, ldo, M, C, float(branch_scale)),
               "mlp_ln_residual_fwd")
    return out

"""Thin Python wrappers over the C ABI: device pointers in, torch-allocated outputs out.

Every function requires CUDA(HIP) fp32 contiguous tensors and launches on torch's current stream.
"""
import torch

from . import _lib

ACT_NONE, ACT_GELU = 0, 1

# Optional live kernel timing (bench.py): HIP events recorded on the launch stream around each launch.
_timing = None


def timing_start():
    global _timing
    _timing = []


def timing_stop(kind):
    """-> (total ms, total algorithmic FLOP (or bytes), launches) of the launches tagged `kind`; stops timing."""
    global _timing
    rec, _timing = _timing or [], None
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for k, a, b, w in rec if k == kind)
    work = sum(w for k, a, b, w in rec if k == kind)
    return ms, work, sum(1 for r in rec if r[0] == kind)


class _timed:
    def __init__(self, kind, work):
        self.kind, self.work = kind, work

    def __enter__(self):
        if _timing is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if _timing is not None:
            self.b.record(torch.cuda.current_stream())
            _timing.append((self.kind, self.a, self.b, self.work))



def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the Pangu HIP path needs tensors on an MI355X device (got {t.device}); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


def _rows(t, name):
    """2-D tensor with unit inner stride (row stride may exceed the width)."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{name}: expected a CUDA float32 2-D tensor with unit inner stride")
    return t.data_ptr(), t.stride(0)


def window_index(Z, H, W, shifted, device):
    lib = _lib.load()
    Hp = H + 5
    out = torch.empty((W // 12, (Z // 2) * (Hp // 6), 144), dtype=torch.int32, device=device)
    _lib.check(lib.pangu_window_index_export(_stream(), out.data_ptr(), Z, H, W, int(shifted)), "window_index_export")
    return out


def window_mask(Z, H, W, device):
    lib = _lib.load()
    Hp = H + 5
    out = torch.empty(((Z // 2) * (Hp // 6), 144, 144), dtype=torch.float32, device=device)
    _lib.check(lib.pangu_window_mask_export(_stream(), out.data_ptr(), Z, H, W), "window_mask_export")
    return out


def linear(a, weight, bias=None, act=ACT_NONE, out=None):
    """out[M,N] = act(a[M,K] @ weight[N,K]^T + bias). `a`/`out` may be row-strided views."""
    lib = _lib.load()
    ap, lda = _rows(a, "linear.a")
    M, K = a.shape
    w2 = weight.reshape(weight.shape[0], -1)          # Conv1d(k=1) weights are (out,in,1)
    N = w2.shape[0]
    if w2.shape[1] != K:
        raise RuntimeError(f"linear: weight {tuple(weight.shape)} does not match input width {K}")
    wp = _chk(w2, "linear.weight")
    bp = _chk(bias, "linear.bias") if bias is not None else None
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    op, ldc = _rows(out, "linear.out")
    with _timed("linear", 2.0 * M * N * K):
        _lib.check(lib.pangu_linear_fwd(_stream(), ap, lda, wp, bp, op, ldc, M, N, K, act), "linear_fwd")
    return out


def window_attention(qkv, qkv_bias, esb, Z, H, W, heads, shifted, want_lse=False):
    lib = _lib.load()
    N, C3 = qkv.shape
    C = C3 // 3
    if N != Z * H * W:
        raise RuntimeError(f"window_attention: {N} tokens != {Z}x{H}x{W}")
    out = torch.empty((N, C), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((N, heads), dtype=torch.float32, device=qkv.device) if want_lse else None
    Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144          # padded token count: the core's FLOPs (4*Np*144*C)
    with _timed("attn", 4.0 * Np * 144 * C):
        _lib.check(lib.pangu_window_attn_fwd(_stream(), _chk(qkv, "qkv"), _chk(qkv_bias, "qkv_bias"), _chk(esb, "esb"),
                                             out.data_ptr(), lse.data_ptr() if want_lse else None, Z, H, W, C, heads,
                                             int(shifted)), "window_attn_fwd")
    return (out, lse) if want_lse else out


def ln_residual(y, shortcut, gamma, beta, out=None, branch_scale=1.0, want_stats=False):
    lib = _lib.load()
    N, C = y.shape
    sp, lds = _rows(shortcut, "ln_residual.shortcut")
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=y.device)
    op, ldo = _rows(out, "ln_residual.out")
    stats = torch.empty((N, 2), dtype=torch.float32, device=y.device) if want_stats else None
    _lib.check(lib.pangu_ln_residual_fwd(_stream(), _chk(y, "ln_residual.y"), sp, lds, _chk(gamma, "gamma"),
                                         _chk(beta, "beta"), op, ldo, stats.data_ptr() if want_stats else None, N, C,
                                         float(branch_scale)), "ln_residual_fwd")
    return (out, stats) if want_stats else out


def downsample_ln(x, gamma, beta, Z, H, W, want_stats=False):
    lib = _lib.load()
    xp, ldx = _rows(x, "downsample.x")
    C = x.shape[1]
    rows = Z * ((H + 1) // 2) * (W // 2)
    out = torch.empty((rows, 4 * C), dtype=torch.float32, device=x.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if want_stats else None
    _lib.check(lib.pangu_downsample_ln_fwd(_stream(), xp, ldx, _chk(gamma, "gamma"), _chk(beta, "beta"), out.data_ptr(),
                                           stats.data_ptr() if want_stats else None, Z, H, W, C), "downsample_ln_fwd")
    return (out, stats) if want_stats else out


def upsample_ln(y, gamma, beta, Z, H2, W2, H, want_stats=False):
    lib = _lib.load()
    Co = y.shape[1] // 4
    rows = Z * H * 2 * W2
    out = torch.empty((rows, Co), dtype=torch.float32, device=y.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=y.device) if want_stats else None
    _lib.check(lib.pangu_upsample_ln_fwd(_stream(), _chk(y, "upsample.y"), _chk(gamma, "gamma"), _chk(beta, "beta"),
                                         out.data_ptr(), stats.data_ptr() if want_stats else None, Z, H2, W2, H, Co),
               "upsample_ln_fwd")
    return (out, stats) if want_stats else out


def patch_embed_gather(inp, inp_surface, s_mean, s_std, u_mean, u_std, maps, const_h):
    """One sample: inp (5,13,LAT,LON), inp_surface (4,LAT,LON) -> (a_surface [H4*W4,112], a_upper [7*H4*W4,192])."""
    lib = _lib.load()
    LAT, LON = inp.shape[-2], inp.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    a_s = torch.empty((H4 * W4, 112), dtype=torch.float32, device=inp.device)
    a_u = torch.empty((7 * H4 * W4, 192), dtype=torch.float32, device=inp.device)
    _lib.check(lib.pangu_patch_embed_gather(_stream(), _chk(inp, "input"), _chk(inp_surface, "input_surface"),
                                            _chk(s_mean, "surface_mean"), _chk(s_std, "surface_std"),
                                            _chk(u_mean, "upper_mean"), _chk(u_std, "upper_std"), _chk(maps, "maps"),
                                            _chk(const_h, "const_h"), a_s.data_ptr(), a_u.data_ptr(), LAT, LON),
               "patch_embed_gather")
    return a_s, a_u


def patch_recover_scatter(y_upper, y_surface, LAT, LON):
    lib = _lib.load()
    out = torch.empty((5, 13, LAT, LON), dtype=torch.float32, device=y_upper.device)
    out_s = torch.empty((4, LAT, LON), dtype=torch.float32, device=y_upper.device)
    _lib.check(lib.pangu_patch_recover_scatter(_stream(), _chk(y_upper, "y_upper"), _chk(y_surface, "y_surface"),
                                               out.data_ptr(), out_s.data_ptr(), LAT, LON), "patch_recover_scatter")
    return out, out_s

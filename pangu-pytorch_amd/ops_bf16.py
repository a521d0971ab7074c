"""bf16 op wrappers (activations and weight shadows bf16; bias / LayerNorm parameters / softmax fp32)."""
import torch

from . import _lib
from . import ops
from .ops import _row_chunks, _stream, _timed, _zeros

ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_ADD = 0, 1, 2, 3
BF16, F32 = 1, 0


def _p(t, name, dtype=torch.bfloat16):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous CUDA {dtype} tensor (got {t.dtype}, {t.device})")
    return t.data_ptr()


def _rows(t, name):
    if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{name}: expected a CUDA 2-D tensor with unit inner stride")
    return t.data_ptr(), t.stride(0)


def linear(a, weight, bias=None, act=ACT_NONE, out=None, aux=None, out_dtype=torch.bfloat16):
    lib = _lib.load()
    ap, lda = _rows(a, "linear.a")
    M, K = a.shape
    N = weight.shape[0]
    if weight.shape[1] != K or a.dtype != torch.bfloat16:
        raise RuntimeError(f"linear_bf16: a {tuple(a.shape)} {a.dtype} vs weight {tuple(weight.shape)}")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    op, ldc = _rows(out, "linear.out")
    chunks = _row_chunks(M, 2 * lda, out.element_size() * ldc)      # 32-bit byte offsets in the kernels (PANGU_E_RANGE)
    if chunks is not None:
        for m0, m1 in chunks:
            linear(a[m0:m1], weight, bias, act, out[m0:m1], aux[m0:m1] if aux is not None else None, out_dtype)
        return out
    with _timed("linear_bf16", 2.0 * M * N * K):
        _lib.check(lib.pangu_linear_fwd_bf16(_stream(a), ap, lda, _p(weight, "weight"),
                                             _p(bias, "bias", torch.float32) if bias is not None else None, op, ldc, M,
                                             N, K, act, _p(aux, "aux") if aux is not None else None,
                                             BF16 if out.dtype == torch.bfloat16 else F32), "linear_fwd_bf16")
    return out


def window_attention(qkv, qkv_bias, esb, Z, H, W, heads, shifted, want_lse=False):
    lib = _lib.load()
    N, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty((N, C), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((N, heads), dtype=torch.float32, device=qkv.device) if want_lse else None
    Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
    with _timed("attn_bf16", 4.0 * Np * 144 * C):
        _lib.check(lib.pangu_window_attn_fwd_bf16(_stream(qkv), _p(qkv, "qkv"), _p(qkv_bias, "qkv_bias"), _p(esb, "esb"),
                                                  out.data_ptr(), lse.data_ptr() if want_lse else None, Z, H, W, C, heads,
                                                  int(shifted)), "window_attn_fwd_bf16")
    return (out, lse) if want_lse else out


def window_attention_qkv(x, w_qkv, b_qkv, esb, Z, H, W, heads, shifted, want_lse=False):
    """Earth-specific window attention INCLUDING the QKV projection (reference layers.py:365-415 up to linear2): x (N, C)
    bf16 rows, w_qkv (3C, C) bf16, b_qkv (3C,) fp32, esb (types, heads, 144, 144) bf16 -> (N, C) bf16.  The qkv tensor is
    never written.  (The longitude-walking form of this operator lost its A/B and lives in experiments/.)"""
    lib = _lib.load()
    xp, ldx = _rows(x, "attn_qkv.x")
    N, C = x.shape
    if N != Z * H * W or tuple(w_qkv.shape) != (3 * C, C) or x.dtype != torch.bfloat16:
        raise RuntimeError(f"window_attention_qkv: x {tuple(x.shape)} {x.dtype}, w_qkv {tuple(w_qkv.shape)}, grid {(Z, H, W)}")
    out = torch.empty((N, C), dtype=torch.bfloat16, device=x.device)
    lse = torch.empty((N, heads), dtype=torch.float32, device=x.device) if want_lse else None
    Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
    with _timed("attn_qkv_bf16", 4.0 * Np * 144 * C + 6.0 * Np * C * C):
        _lib.check(lib.pangu_window_attn_qkv_fwd_bf16(_stream(x), xp, ldx, _p(w_qkv, "w_qkv"), _p(b_qkv, "b_qkv", torch.float32),
                                                      _p(esb, "esb"), out.data_ptr(), lse.data_ptr() if want_lse else None,
                                                      Z, H, W, C, heads, int(shifted)), "window_attn_qkv_fwd_bf16")
    return (out, lse) if want_lse else out


def linear_ln_residual(a, weight, bias, shortcut, gamma, beta, out=None):
    """out = shortcut + LayerNorm(a @ weight^T + bias) * gamma + beta in ONE launch (N = 192 or 384; inference path)."""
    lib = _lib.load()
    ap, lda = _rows(a, "linear_ln.a")
    M, K = a.shape
    N = weight.shape[0]
    if weight.shape[1] != K or a.dtype != torch.bfloat16 or shortcut.shape != (M, N) or not shortcut.is_contiguous():
        raise RuntimeError(f"linear_ln_residual: a {tuple(a.shape)} weight {tuple(weight.shape)} shortcut {tuple(shortcut.shape)}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    op, ldo = _rows(out, "linear_ln.out")
    chunks = _row_chunks(M, 2 * lda, 2 * ldo)
    if chunks is not None:
        for m0, m1 in chunks:
            linear_ln_residual(a[m0:m1], weight, bias, shortcut[m0:m1], gamma, beta, out[m0:m1])
        return out
    with _timed("linear_ln_bf16", 2.0 * M * N * K):
        _lib.check(lib.pangu_linear_ln_residual_fwd_bf16(
            _stream(a), ap, lda, _p(weight, "weight"), _p(bias, "bias", torch.float32) if bias is not None else None,
            _p(shortcut, "shortcut"), _p(gamma, "gamma", torch.float32), _p(beta, "beta", torch.float32), op, ldo, M, N, K),
            "linear_ln_residual_fwd_bf16")
    return out


_mlp_pack_index = {}      # (C, device) -> int64 gather index of the packed image into cat(w1.flatten(), w2.flatten())


def _mlp_pack_layout(w1, w2):
    """The chunk image as indexing operations on (w1 (4C, C), w2 (C, 4C)) of any dtype -> (2, 4C/32, 32*C)."""
    HID, C = w1.shape
    dev = w1.device
    nch = HID // 32
    hr = torch.arange(32, device=dev)
    pos = torch.arange(C // 8, device=dev)
    f = (hr & 15) if C == 384 else ((hr >> 1) & 7)
    src = pos[None, :] ^ f[:, None]                                           # logical chunk stored at (row, position)
    w1b = w1.reshape(nch, 32, C // 8, 8)
    w1img = w1b[:, hr[:, None], src, :]                                       # (nch, 32, C/8, 8)
    perm = torch.tensor([16 * s + 8 * (j >> 2) + 4 * h + (j & 3) for s in range(2) for h in range(2) for j in range(8)],
                        device=dev)
    w2img = w2.reshape(C, nch, 32)[:, :, perm].reshape(C, nch, 4, 8).permute(1, 2, 0, 3)
    return torch.stack([w1img.reshape(nch, 32 * C), w2img.reshape(nch, 32 * C)], 0).contiguous()


_mlp_pack_index32 = {}    # (C, device) -> the same index as int32 (the job table of pangu_shadow_refresh_bf16 gathers through it)


def mlp_pack_index32(C, device):
    """int32 form of the pack index (built by the first pack_mlp_weights call of that (C, device))."""
    key = (C, device)
    i32 = _mlp_pack_index32.get(key)
    if i32 is None:
        i32 = _mlp_pack_index32[key] = _mlp_pack_index[key].to(torch.int32)
    return i32


def pack_mlp_weights(w1, w2):
    """Chunk image of an Mlp's weights for `mlp_ln_residual` (layout: csrc/mlp_fused_bf16.hip header).  w1 (4C, C), w2 (C, 4C),
    any float dtype, C in (192, 384) -> bf16 (2, 4C/32, 32*C): plane 0 = per 32 hidden units the W1 rows (16-B chunks XOR-swizzled
    for conflict-free fragment reads), plane 1 = the W2 columns in the k order the first product's accumulators come in.
    The layout is a fixed permutation: it is built once per (C, device) as a gather index (by pushing element numbers through
    `_mlp_pack_layout`), so re-packing after an optimizer step is three launches (concatenate, cast, gather)."""
    HID, C = w1.shape
    if C not in (192, 384) or HID != 4 * C or tuple(w2.shape) != (C, HID):
        raise RuntimeError(f"pack_mlp_weights: w1 {tuple(w1.shape)} w2 {tuple(w2.shape)}")
    key = (C, w1.device)
    idx = _mlp_pack_index.get(key)
    if idx is None:
        n = HID * C
        e1 = torch.arange(n, device=w1.device).view(HID, C)
        e2 = torch.arange(n, 2 * n, device=w1.device).view(C, HID)
        idx = _mlp_pack_index[key] = _mlp_pack_layout(e1, e2).reshape(-1)
    flat = torch.cat((w1.reshape(-1), w2.reshape(-1))).to(torch.bfloat16)
    return flat[idx].view(2, HID // 32, 32 * C)


def mlp_ln_residual(x, w_packed, b1, b2, gamma, beta, out=None, branch_scale=1.0):
    """out = x + branch_scale * LayerNorm(GELU(x W1^T + b1) W2^T + b2) * gamma + beta in ONE launch (reference
    layers.py:251 with Mlp.forward :264-270 inside); the hidden activation never reaches memory.  x (M, C) bf16 rows,
    w_packed from pack_mlp_weights, biases / LayerNorm parameters fp32."""
    lib = _lib.load()
    xp, ldx = _rows(x, "mlp.x")
    M, C = x.shape
    if x.dtype != torch.bfloat16 or tuple(w_packed.shape) != (2, C // 8, 32 * C):
        raise RuntimeError(f"mlp_ln_residual: x {tuple(x.shape)} {x.dtype} vs packed weights {tuple(w_packed.shape)}")
    if out is None:
        out = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
    op, ldo = _rows(out, "mlp.out")
    chunks = _row_chunks(M, 2 * ldx, 2 * ldo)
    if chunks is not None:
        for m0, m1 in chunks:
            mlp_ln_residual(x[m0:m1], w_packed, b1, b2, gamma, beta, out[m0:m1], branch_scale)
        return out
    with _timed("mlp_fused_bf16", 16.0 * M * C * C):
        _lib.check(lib.pangu_mlp_ln_residual_fwd_bf16(
            _stream(x), xp, ldx, _p(w_packed, "w_packed"), _p(b1, "b1", torch.float32), _p(b2, "b2", torch.float32),
            _p(gamma, "gamma", torch.float32), _p(beta, "beta", torch.float32), op, ldo, M, C, float(branch_scale)),
            "mlp_ln_residual_fwd_bf16")
    return out


def mlp_ln_residual_train(x, w_packed, b1, b2, gamma, beta, branch_scale=1.0, out=None):
    """Training forward of the MLP branch in ONE launch: -> (out, pre, m) with out as `mlp_ln_residual`, pre (M, 4C) bf16 =
    x W1^T + b1 before the GELU and m (M, C) bf16 =
    GELU(pre) W2^T + b2 before the LayerNorm.  h = GELU(pre) is not stored (linear_gelu_bwd re-creates it)."""
    lib = _lib.load()
    xp, ldx = _rows(x, "mlp.x")
    M, C = x.shape
    if x.dtype != torch.bfloat16 or tuple(w_packed.shape) != (2, C // 8, 32 * C):
        raise RuntimeError(f"mlp_ln_residual_train: x {tuple(x.shape)} {x.dtype} vs packed weights {tuple(w_packed.shape)}")
    if _row_chunks(M, 2 * ldx, 8 * C) is not None:
        raise RuntimeError("mlp_ln_residual_train: more than 4 GB of pre-activation rows in one call")
    if out is None:
        out = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
    op, ldo = _rows(out, "mlp.out")              # may be one half of the (M, 2C) skip-concat buffer
    m = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
    pre = torch.empty((M, 4 * C), dtype=torch.bfloat16, device=x.device)
    with _timed("mlp_fused_bf16", 16.0 * M * C * C):
        _lib.check(lib.pangu_mlp_ln_residual_train_fwd_bf16(
            _stream(x), xp, ldx, _p(w_packed, "w_packed"), _p(b1, "b1", torch.float32), _p(b2, "b2", torch.float32),
            _p(gamma, "gamma", torch.float32), _p(beta, "beta", torch.float32), op, ldo,
            pre.data_ptr(), 4 * C, m.data_ptr(), C, M, C, float(branch_scale)),
            "mlp_ln_residual_train_fwd_bf16")
    return out, pre, m


def linear_gelu_bwd(dm, w2_t, pre, want_h=True):
    """Backward through Mlp.linear2 + GELU: -> (dpre, h) with dpre = (dm @ w2_t^T) * gelu'(pre) and h = GELU(pre) (None
    with want_h=False).  dm (M, C) bf16 rows, w2_t (4C, C) bf16 (= linear2.weight transposed), pre (M, 4C) bf16."""
    lib = _lib.load()
    ap, lda = _rows(dm, "gelu_bwd.dm")
    M, K = dm.shape
    N = w2_t.shape[0]
    if w2_t.shape[1] != K or tuple(pre.shape) != (M, N) or dm.dtype != torch.bfloat16:
        raise RuntimeError(f"linear_gelu_bwd: dm {tuple(dm.shape)} w2_t {tuple(w2_t.shape)} pre {tuple(pre.shape)}")
    dpre = torch.empty((M, N), dtype=torch.bfloat16, device=dm.device)
    h = torch.empty((M, N), dtype=torch.bfloat16, device=dm.device) if want_h else None
    chunks = _row_chunks(M, 2 * lda, 2 * N)
    for m0, m1 in (chunks or [(0, M)]):
        with _timed("linear_bf16", 2.0 * (m1 - m0) * N * K):
            _lib.check(lib.pangu_linear_gelu_bwd_bf16(_stream(dm), ap + m0 * lda * 2, lda, _p(w2_t, "w2_t"),
                                                      dpre[m0:m1].data_ptr(), N, m1 - m0, N, K, _p(pre, "pre") + m0 * N * 2,
                                                      h[m0:m1].data_ptr() if want_h else None), "linear_gelu_bwd_bf16")
    return dpre, h


def ln_residual(y, shortcut, gamma, beta, out=None, branch_scale=1.0):
    lib = _lib.load()
    N, C = y.shape
    sp, lds = _rows(shortcut, "shortcut")
    if out is None:
        out = torch.empty((N, C), dtype=torch.bfloat16, device=y.device)
    op, ldo = _rows(out, "out")
    _lib.check(lib.pangu_ln_residual_fwd_bf16(_stream(y), _p(y, "y"), sp, lds, _p(gamma, "gamma", torch.float32),
                                              _p(beta, "beta", torch.float32), op, ldo, N, C, float(branch_scale)),
               "ln_residual_fwd_bf16")
    return out


def downsample_ln(x, gamma, beta, Z, H, W):
    lib = _lib.load()
    xp, ldx = _rows(x, "x")
    C = x.shape[1]
    out = torch.empty((Z * ((H + 1) // 2) * (W // 2), 4 * C), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.pangu_downsample_ln_fwd_bf16(_stream(x), xp, ldx, _p(gamma, "gamma", torch.float32),
                                                _p(beta, "beta", torch.float32), out.data_ptr(), Z, H, W, C),
               "downsample_ln_fwd_bf16")
    return out


def upsample_ln(y, gamma, beta, Z, H2, W2, H):
    lib = _lib.load()
    Co = y.shape[1] // 4
    out = torch.empty((Z * H * 2 * W2, Co), dtype=torch.bfloat16, device=y.device)
    _lib.check(lib.pangu_upsample_ln_fwd_bf16(_stream(y), _p(y, "y"), _p(gamma, "gamma", torch.float32),
                                              _p(beta, "beta", torch.float32), out.data_ptr(), Z, H2, W2, H, Co),
               "upsample_ln_fwd_bf16")
    return out


def patch_embed_gather(inp, inp_surface, s_mean, s_std, u_mean, u_std, maps, const_h, levels_reversed=False):
    lib = _lib.load()
    LAT, LON = inp.shape[-2], inp.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    a_s = torch.empty((H4 * W4, 128), dtype=torch.bfloat16, device=inp.device)
    a_u = torch.empty((7 * H4 * W4, 192), dtype=torch.bfloat16, device=inp.device)
    f = torch.float32
    _lib.check(lib.pangu_patch_embed_gather_bf16(_stream(inp), _p(inp, "input", f), _p(inp_surface, "input_surface", f),
                                                 _p(s_mean, "s_mean", f), _p(s_std, "s_std", f), _p(u_mean, "u_mean", f),
                                                 _p(u_std, "u_std", f), _p(maps, "maps", f), _p(const_h, "const_h", f),
                                                 a_s.data_ptr(), a_u.data_ptr(), LAT, LON, int(bool(levels_reversed))),
               "patch_embed_gather_bf16")
    return a_s, a_u


# ---------------------------------------------------------------- backward
_WGRAD_WS_BYTES = ops._WGRAD_WS_BYTES
_wgrad_workspace = ops.wgrad_workspace      # one scratch buffer per device for both dtypes


def linear_wgrad(dc, a, want_bias=True, db_into=None):
    """dW[N,K] (fp32) = dc[M,N]^T @ a[M,K], db[N] = colsum(dc); bf16 operands (row-strided views allowed).  The partial tiles of
    the token slabs travel through a per-device scratch buffer (96 MB, allocated on first use; one per device: do not run weight gradients of one device on two streams at once; the few fp32 shapes whose slabs need 108 MB keep the atomic tail, measured level) instead of fp32 atomics."""
    lib = _lib.load()
    ws = _wgrad_workspace(dc.device)
    dp, lddc = _rows(dc, "wgrad.dc")
    ap, lda = _rows(a, "wgrad.a")
    M, N = dc.shape
    K = a.shape[1]
    # db_into: an fp32 (N,) buffer that already holds a partial bias gradient (the attention backward's pad-slot term): the column
    # sums are ADDED to it (the kernels accumulate atomically anyway) instead of a separate buffer + a torch add afterwards
    own_db = want_bias and db_into is None
    buf = _zeros((N * K + (N if own_db else 0),), dc.device)   # one fill launch (none inside a zero_arena)
    dw = buf[:N * K].view(N, K)
    db = (buf[N * K:] if own_db else db_into) if want_bias else None
    for m0, m1 in (_row_chunks(M, 2 * lddc, 2 * lda) or [(0, M)]):      # the kernel ADDS into dw / db
        with _timed("wgrad_bf16", 2.0 * (m1 - m0) * N * K):
            _lib.check(lib.pangu_linear_wgrad_bf16_ws(_stream(dc), dp + m0 * lddc * 2, lddc, ap + m0 * lda * 2, lda, dw.data_ptr(),
                                                      db.data_ptr() if want_bias else None, m1 - m0, N, K, ws.data_ptr(),
                                                      _WGRAD_WS_BYTES), "linear_wgrad_bf16")
    return dw, db


def window_attention_bwd(qkv, qkv_bias, esb, out, lse, dout, Z, H, W, heads, shifted, desb_out=None):
    lib = _lib.load()
    N, C3 = qkv.shape
    C = C3 // 3
    dqkv = torch.empty_like(qkv)
    dqb = _zeros((C3,), qkv.device)
    desb = torch.empty(esb.shape, dtype=torch.float32, device=qkv.device) if desb_out is None else desb_out.view(esb.shape)
    Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
    with _timed("attn_bwd_bf16", 14.0 * Np * 144 * C):
        _lib.check(lib.pangu_window_attn_bwd_bf16(_stream(qkv), _p(qkv, "qkv"), _p(qkv_bias, "qkv_bias"), _p(esb, "esb"),
                                                  _p(out, "out"), _p(lse, "lse", torch.float32), _p(dout, "dout"),
                                                  dqkv.data_ptr(), dqb.data_ptr(), desb.data_ptr(), Z, H, W, C, heads,
                                                  int(shifted)), "window_attn_bwd_bf16")
    return dqkv, dqb, desb


def ln_residual_bwd(dout, y, gamma, branch_scale=1.0):
    lib = _lib.load()
    N, C = y.shape
    dp, lddo = _rows(dout, "ln_bwd.dout")
    dy = torch.empty_like(y)
    dg, db = _zeros((2, C), y.device).unbind(0)
    _lib.check(lib.pangu_ln_residual_bwd_bf16(_stream(dout), dp, lddo, _p(y, "y"), _p(gamma, "gamma", torch.float32),
                                              dy.data_ptr(), dg.data_ptr(), db.data_ptr(), N, C, float(branch_scale)),
               "ln_residual_bwd_bf16")
    return dy, dg, db


def downsample_ln_bwd(dout, x, gamma, Z, H, W, add=None):
    """add (optional, dense (Z*H*W, C) bf16): a second gradient of the same tokens, summed into dx by the kernel."""
    lib = _lib.load()
    xp, ldx = _rows(x, "x")
    C = x.shape[1]
    dx = torch.empty((Z * H * W, C), dtype=torch.bfloat16, device=x.device)
    if add is not None and tuple(add.shape) != tuple(dx.shape):
        raise RuntimeError(f"downsample_ln_bwd: addend {tuple(add.shape)} != {tuple(dx.shape)}")
    dg, db = _zeros((2, 4 * C), x.device).unbind(0)
    _lib.check(lib.pangu_downsample_ln_bwd_bf16(_stream(dout), _p(dout, "dout"), xp, ldx, _p(gamma, "gamma", torch.float32),
                                                dx.data_ptr(), dg.data_ptr(), db.data_ptr(), Z, H, W, C,
                                                _p(add, "add") if add is not None else None), "downsample_ln_bwd_bf16")
    return dx, dg, db


def upsample_ln_bwd(dout, y, gamma, Z, H2, W2, H):
    lib = _lib.load()
    Co = y.shape[1] // 4
    dy = torch.empty_like(y)
    dg, db = _zeros((2, Co), y.device).unbind(0)
    _lib.check(lib.pangu_upsample_ln_bwd_bf16(_stream(dout), _p(dout, "dout"), _p(y, "y"), _p(gamma, "gamma", torch.float32),
                                              dy.data_ptr(), dg.data_ptr(), db.data_ptr(), Z, H2, W2, H, Co),
               "upsample_ln_bwd_bf16")
    return dy, dg, db


def patch_recover_gather_bwd(d_out, d_out_s):
    lib = _lib.load()
    LAT, LON = d_out.shape[-2], d_out.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    dy_u = torch.empty((7 * H4 * W4, 160), dtype=torch.bfloat16, device=d_out.device)
    dy_s = torch.empty((H4 * W4, 64), dtype=torch.bfloat16, device=d_out.device)
    _lib.check(lib.pangu_patch_recover_gather_bwd_bf16(_stream(d_out), _p(d_out, "d_output", torch.float32),
                                                       _p(d_out_s, "d_output_surface", torch.float32), dy_u.data_ptr(),
                                                       dy_s.data_ptr(), LAT, LON), "patch_recover_gather_bwd_bf16")
    return dy_u, dy_s

"""Forward compositions of the HIP kernels for each reference layer (inference path, no autograd graph).

Token tensors are (B, N, C); B is folded into rows for projections / LayerNorm and looped for the
geometry-dependent kernels (the reference itself is B=1 only, models/layers.py:219,227).
"""
import os

import torch

from . import ops
from .autograd import DownSampleFn, EarthBlockFn, MlpFn, PatchEmbedFn, PatchRecoverFn, PatchRecoverHalvesFn, UpSampleFn


def _train_path(module, *tensors):
    """Autograd path iff grad mode is on and something upstream (a parameter or an activation) needs a gradient."""
    if not torch.is_grad_enabled():
        return False
    return any(p.requires_grad for p in module.parameters()) or any(t.requires_grad for t in tensors)


def _stack(outs, B):
    return outs[0].unsqueeze(0) if B == 1 else torch.stack(outs, 0)


def _samples(x):
    """Per-sample contiguous tensors of a batched activation for the autograd path.  `x[b]` would put a SelectBackward
    into the graph (a zero fill + a copy of the whole activation per layer in backward: 2.8 ms of the fp32 step); the
    B = 1 case is a pure view, B > 1 goes through one unbind."""
    if x.shape[0] == 1:
        t = x.reshape(x.shape[1:])
        return [t if t.stride(-1) == 1 and t.dim() == 2 else t.contiguous()]      # row-strided rows are fine for every kernel
    return [t.contiguous() for t in x.unbind(0)]


def _tok2d(x):
    """(B,N,C) -> 2-D row view (B*N, C) (row-strided views stay views)."""
    B, N, C = x.shape
    if x.stride(2) != 1 or (B > 1 and x.stride(0) != N * x.stride(1)):
        x = x.contiguous()
    return x.as_strided((B * N, C), (x.stride(1), 1), x.storage_offset())


_FUSE_LN = os.environ.get("PANGU_F32_FUSE_LN", "1") != "0"       # A/B knob: 0 = separate GEMM + LN-residual launches

def mlp(m, x2d):
    """Mlp.forward on its own (reference layers.py:264-270; the block never comes through here): differentiable."""
    if _train_path(m, x2d):
        return MlpFn.apply(x2d.contiguous(), m.linear1.weight, m.linear1.bias, m.linear2.weight, m.linear2.bias)
    h = ops.linear(x2d, m.linear1.weight, m.linear1.bias, act=ops.ACT_GELU)
    return ops.linear(h, m.linear2.weight, m.linear2.bias)


def earth_block(blk, x, Z, H, W, roll, out=None):
    """x (B,N,C) -> (B,N,C).  reference layers.py:183-253 as 5 kernel launches per sample (qkv, attention core, proj+LN+residual, MLP-up+GELU, MLP-down+LN+residual)."""
    B, N, C = x.shape
    att = blk.attention
    dp = blk.drop_path
    if _train_path(blk, x):
        outs = []
        # out as a 2-D (N, C) row-strided tensor (B = 1): the block function writes its result there (a half of the
        # skip-concat buffer, PanguModel._forward_f32) -- no copy
        direct = out is not None and out.dim() == 2 and B == 1
        for xb in _samples(x):
            s1 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
            s2 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
            outs.append(EarthBlockFn.apply(
                xb, blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias,
                blk.linear.linear1.weight, blk.linear.linear1.bias, blk.linear.linear2.weight, blk.linear.linear2.bias,
                att.earth_specific_bias, att.linear1.weight, att.linear1.bias, att.linear2.weight, att.linear2.bias,
                (Z, H, W, att.head_number, bool(roll)), s1, s2, (out,) if direct else None))
        y = _stack(outs, B)
        if out is not None and not direct:
            out.copy_(y)
            return out
        return y
    x2 = _tok2d(x)
    s1 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
    s2 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
    if s1 != 0.0:
        qkv = ops.linear(x2, att.linear1.weight, att.linear1.bias)                      # (B*N, 3C)
        # inference on the paper's compact bias table (PanguModel.use_compact_bias): 10 MB instead of 62 MB per block
        esb_c = getattr(att, "_esb_compact", None)
        if esb_c is not None:                 # stale table (weights changed since it was folded): the expanded parameter
            p = att.earth_specific_bias
            if getattr(att, "_esb_compact_stamp", None) != ops.param_stamp(p) or esb_c.device != p.device:
                att._esb_compact = esb_c = None
        cp = esb_c is not None
        esb = esb_c if cp else att.earth_specific_bias[0]
        o = torch.cat([ops.window_attention(qkv[b * N:(b + 1) * N], att.linear1.bias, esb, Z, H, W,
                                            att.head_number, roll, compact=cp) for b in range(B)], 0) if B > 1 else \
            ops.window_attention(qkv, att.linear1.bias, esb, Z, H, W, att.head_number, roll, compact=cp)
        if _FUSE_LN and C in (192, 384):     # projection + post-norm residual in one launch (the GEMM tile spans the row)
            x1 = ops.linear_ln_residual(o, att.linear2.weight, att.linear2.bias, x2, blk.norm1.weight, blk.norm1.bias,
                                        branch_scale=s1)
        else:
            y = ops.linear(o, att.linear2.weight, att.linear2.bias)
            x1 = ops.ln_residual(y, x2, blk.norm1.weight, blk.norm1.bias, branch_scale=s1)
    else:
        x1 = x2
    # out may be the 2-D (N, C) row-strided half of the skip-concat buffer that PanguModel._forward_f32 hands to the LAST block
    # of layer 0 / 3 on the autograd path; a partially frozen fine-tune (nothing upstream of this block trains) lands here
    o2 = None if out is None else (out if out.dim() == 2 else _tok2d(out))
    if s2 != 0.0:
        if _FUSE_LN and C in (192, 384):
            h = ops.linear(x1, blk.linear.linear1.weight, blk.linear.linear1.bias, act=ops.ACT_GELU)
            x2o = ops.linear_ln_residual(h, blk.linear.linear2.weight, blk.linear.linear2.bias, x1, blk.norm2.weight,
                                         blk.norm2.bias, out=o2, branch_scale=s2)
        else:
            m = mlp(blk.linear, x1)
            x2o = ops.ln_residual(m, x1, blk.norm2.weight, blk.norm2.bias, out=o2, branch_scale=s2)
    else:
        x2o = x1
        if out is not None:
            o2.copy_(x1)
            x2o = o2
    if out is not None:
        return out.unsqueeze(0) if out.dim() == 2 else out
    return x2o.view(B, N, C)


def patch_embed(m, inp, inp_surface, statistics, maps, const_h, levels_reversed=False):
    """reference layers.py:40-93 -> (B, 8*181*360, 192)."""
    s_mean, s_std, u_mean, u_std = statistics
    B = inp.shape[0]
    LAT, LON = inp.shape[-2], inp.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    n_s, n_u = H4 * W4, 7 * H4 * W4
    dim = m.conv.weight.shape[0]
    x = torch.empty((B, n_s + n_u, dim), dtype=torch.float32, device=inp.device)
    f32 = lambda t: t.to(device=inp.device, dtype=torch.float32).contiguous()
    s_mean, s_std = f32(s_mean).reshape(-1), f32(s_std).reshape(-1)
    u_mean, u_std = f32(u_mean).reshape(13, 5), f32(u_std).reshape(13, 5)
    maps_c = f32(maps).reshape(3, 4 * H4, LON)
    const_c = f32(const_h).reshape(13, LAT, LON)
    if _train_path(m, inp, inp_surface):
        return _stack([PatchEmbedFn.apply(m.conv.weight, m.conv.bias, m.conv_surface.weight, m.conv_surface.bias,
                                          inp[b].contiguous(), inp_surface[b].contiguous(), s_mean, s_std, u_mean,
                                          u_std, maps_c, const_c, levels_reversed) for b in range(B)], B)
    for b in range(B):
        a_s, a_u = ops.patch_embed_gather(inp[b].contiguous(), inp_surface[b].contiguous(), s_mean, s_std, u_mean,
                                          u_std, maps_c, const_c, levels_reversed)
        ops.linear(a_s, m.conv_surface.weight, m.conv_surface.bias, out=x[b, :n_s])
        ops.linear(a_u, m.conv.weight, m.conv.bias, out=x[b, n_s:])
    return x


def down_sample(m, x, Z, H, W, skip_grad=None):
    B, N, C = x.shape
    if _train_path(m, x):
        return _stack([DownSampleFn.apply(xb, m.linear.weight, m.norm.weight, m.norm.bias, (Z, H, W), skip_grad if B == 1 else None)
                       for xb in _samples(x)], B)
    outs = []
    for b in range(B):
        g = ops.downsample_ln(_tok2d(x[b:b + 1]), m.norm.weight, m.norm.bias, Z, H, W)
        outs.append(ops.linear(g, m.linear.weight))
    return torch.stack(outs, 0) if B > 1 else outs[0].unsqueeze(0)


def up_sample(m, x, Z, H2, W2, H, out=None):
    B, N, C2 = x.shape
    if _train_path(m, x):
        y = _stack([UpSampleFn.apply(xb, m.linear1.weight, m.linear2.weight, m.norm.weight, m.norm.bias,
                                     (Z, H2, W2, H)) for xb in _samples(x)], B)
        if out is not None:
            out.copy_(y)
            return out
        return y
    y = ops.linear(_tok2d(x), m.linear1.weight)                          # (B*N, 4Co)
    Co = y.shape[1] // 4
    Nf = Z * H * 2 * W2
    if out is None:
        out = torch.empty((B, Nf, Co), dtype=torch.float32, device=x.device)
    for b in range(B):
        g = ops.upsample_ln(y[b * N:(b + 1) * N], m.norm.weight, m.norm.bias, Z, H2, W2, H)
        ops.linear(g, m.linear2.weight, out=_tok2d(out[b:b + 1]))
    return out


def patch_recover_halves(m, skip, x, Z, H, W, LAT=721, LON=1440, skip_grad=None):
    """Training path, B = 1: reference layers.py:511-545 on cat(skip, x) where skip / x (1, N, C) are the two halves of one
    (N, 2C) buffer (PanguModel._forward_f32)."""
    o, os_ = PatchRecoverHalvesFn.apply(skip[0], x[0], m.conv.weight, m.conv.bias, m.conv_surface.weight, m.conv_surface.bias,
                                        (H * W, LAT, LON), skip_grad)
    return o.unsqueeze(0), os_.unsqueeze(0)


def patch_recover(m, x, Z, H, W, LAT=721, LON=1440):
    """x (B, Z*H*W, C) (may be a row-strided view) -> (B,5,13,LAT,LON), (B,4,LAT,LON).  reference layers.py:511-545."""
    B, N, C = x.shape
    n_s = H * W
    if _train_path(m, x):
        res = [PatchRecoverFn.apply(xb, m.conv.weight, m.conv.bias, m.conv_surface.weight,
                                    m.conv_surface.bias, (n_s, LAT, LON)) for xb in _samples(x)]
        return _stack([r[0] for r in res], B), _stack([r[1] for r in res], B)
    outs, outs_s = [], []
    for b in range(B):
        xb = _tok2d(x[b:b + 1])
        y_s = ops.linear(xb[:n_s], m.conv_surface.weight, m.conv_surface.bias)
        y_u = ops.linear(xb[n_s:], m.conv.weight, m.conv.bias)
        o, os_ = ops.patch_recover_scatter(y_u, y_s, LAT, LON)
        outs.append(o)
        outs_s.append(os_)
    return _stack(outs, B), _stack(outs_s, B)

"""bf16 mixed-precision training path (BASELINE configs[2]): fp32 master parameters and fp32 parameter gradients,
bf16 weight shadows / activations / activation gradients, fp32 LayerNorm + softmax + accumulation.

Same kernel sequence as autograd.py on the bf16 entry points; saved activations are bf16 (32 GB instead of 64 GB)."""
import os

import torch

from . import ops
from . import ops_bf16 as ob

# MLP branch of the training forward (A/B knob PANGU_BF16_TRAIN_MLP): 1 (default) = ONE launch that keeps the hidden activation on
# chip and writes only what the backward needs (pre, m); the backward's data-gradient GEMM re-creates h = GELU(pre) for the W2
# weight gradient.  0 = three launches (MLP-up + GELU writing pre AND h, MLP-down, LayerNorm + residual): what widths other than
# 192 / 384 and row-strided inputs take anyway.  (Recomputing the MLP-up GEMM in the backward instead of saving `pre` -- the
# reference's answer to activation memory, layers.py:115-119 -- measured +2.7 ms per step and was removed in round 4; so were the
# QKV-inside-attention training forward, +0.3 ms, and weight gradients on a second stream, +0.5 ms.  DESIGN.md keeps the numbers.)
_TRAIN_MLP = int(os.environ.get("PANGU_BF16_TRAIN_MLP", "1"))


def _mlp_mode(C):
    return 1 if (C in (192, 384) and _TRAIN_MLP != 0) else 0


class EarthBlockFnBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n1w, n1b, n2w, n2b, m1w, m1b, m2w, m2b, esb, a1w, a1b, a2w, a2b, geom, s1, s2, sh, dst=None):
        # dst: optional 1-tuple holding a (N, C) row-strided view the block writes its result into (a half of the skip-concat
        # buffer of reference pangu_model.py:81) -- wrapped so that autograd does not see a tensor argument
        out = dst[0] if dst else None
        Z, H, W, heads, shifted = geom
        ctx.geom, ctx.s1, ctx.s2, ctx.sh = geom, s1, s2, sh
        ctx.params = (n1w, n2w, m1w, m2w, esb, a1w, a1b, a2w, m1b)
        saved = [x]
        x1 = x
        if s1 != 0.0:
            qkv = ob.linear(x, sh.get(a1w), a1b)
            o, lse = ob.window_attention(qkv, sh.get(a1b), sh.get(esb), Z, H, W, heads, shifted, want_lse=True)
            y = ob.linear(o, sh.get(a2w), a2b)
            # (a dropped MLP branch -- s2 == 0 -- makes x1 the block's result: written straight into `out`, no copy afterwards)
            x1 = ob.ln_residual(y, x, n1w, n1b, branch_scale=s1, out=out if s2 == 0.0 else None)
            saved += [qkv, o, lse, y]
        ctx.mlp_mode = mode = _mlp_mode(x.shape[1]) if x1.is_contiguous() else 0
        if s2 != 0.0 and mode:
            x2, pre, m = ob.mlp_ln_residual_train(x1, sh.get_mlp(m1w, m2w), m1b, m2b, n2w, n2b, branch_scale=s2, out=out)
            saved += [x1, pre, m]
        elif s2 != 0.0:
            pre = torch.empty((x.shape[0], m1w.shape[0]), dtype=torch.bfloat16, device=x.device)
            h = ob.linear(x1, sh.get(m1w), m1b, act=ob.ACT_GELU, aux=pre)
            m = ob.linear(h, sh.get(m2w), m2b)
            x2 = ob.ln_residual(m, x1, n2w, n2b, out=out, branch_scale=s2)
            saved += [x1, pre, h, m]
        elif out is not None:
            if x1 is not out:
                out.copy_(x1)                     # both branches dropped: the block is the identity
            x2 = out
        else:
            x2 = x1
        ctx.save_for_backward(*saved)
        return x2

    @staticmethod
    def backward(ctx, dout):
        # (every atomically accumulated gradient buffer of the whole backward pass comes out of ONE zero fill: ops._zeros)
        Z, H, W, heads, shifted = ctx.geom
        s1, s2, sh = ctx.s1, ctx.s2, ctx.sh
        n1w, n2w, m1w, m2w, esb, a1w, a1b, a2w, m1b = ctx.params
        sv = list(ctx.saved_tensors)
        x, rest = sv[0], sv[1:]
        if s1 != 0.0:
            qkv, o, lse, y = rest[:4]
            rest = rest[4:]
        g = {k: None for k in ("n1w", "n1b", "n2w", "n2b", "m1w", "m1b", "m2w", "m2b", "esb", "a1w", "a1b", "a2w", "a2b")}
        dqb_pad = None
        dx1 = dout
        if s2 != 0.0:
            mode = ctx.mlp_mode
            if mode == 1:
                x1, pre, m = rest
            else:
                x1, pre, h, m = rest
            dm, g["n2w"], g["n2b"] = ob.ln_residual_bwd(dout, m, n2w, s2)
            if mode == 1:                  # h = GELU(pre) comes out of the data-gradient GEMM's epilogue (never stored by the forward)
                dpre, h = ob.linear_gelu_bwd(dm, sh.get_t(m2w), pre)
            else:
                dpre = ob.linear(dm, sh.get_t(m2w), None, act=ob.ACT_GELU_BWD, aux=pre)
            g["m2w"], g["m2b"] = ob.linear_wgrad(dm, h)
            del dm, h
            g["m1w"], g["m1b"] = ob.linear_wgrad(dpre, x1)
            if dout.is_contiguous():      # residual gradient added in the GEMM epilogue (no extra pass over N x C)
                dx1 = ob.linear(dpre, sh.get_t(m1w), act=ob.ACT_ADD, aux=dout)
            else:
                dx1 = ob.linear(dpre, sh.get_t(m1w))
                dx1 += dout
            del dpre
        dx = dx1
        if s1 != 0.0:
            dy, g["n1w"], g["n1b"] = ob.ln_residual_bwd(dx1, y, n1w, s1)
            g["a2w"], g["a2b"] = ob.linear_wgrad(dy, o)
            do = ob.linear(dy, sh.get_t(a2w))
            del dy
            dqkv, dqb_pad, desb = ob.window_attention_bwd(qkv, sh.get(a1b), sh.get(esb), o, lse, do, Z, H, W, heads, shifted,
                                                          desb_out=ops.grad_slot(esb))       # straight into the DP flat buffer
            del do
            g["esb"] = desb.unsqueeze(0)
            # linear1's bias gradient = column sums of dqkv (real tokens) + the pad-slot term the attention backward already
            # accumulated into dqb_pad: the weight-gradient kernel adds its sums into that buffer
            fuse_db = dqb_pad is not None and dqb_pad.is_contiguous()
            g["a1w"], g["a1b"] = ob.linear_wgrad(dqkv, x, db_into=dqb_pad if fuse_db else None)
            if fuse_db:
                dqb_pad = None
            if dx1.is_contiguous():
                dx = ob.linear(dqkv, sh.get_t(a1w), act=ob.ACT_ADD, aux=dx1)
            else:
                dx = ob.linear(dqkv, sh.get_t(a1w))
                dx += dx1
        elif not dx.is_contiguous():
            dx = dx.contiguous()
        if dqb_pad is not None:
            g["a1b"] += dqb_pad
        ops.fill_dropped_grads(g, {"n1w": n1w, "n1b": n1w, "n2w": n2w, "n2b": n2w, "m1w": m1w, "m1b": m1b, "m2w": m2w, "m2b": n2w,
                                   "esb": esb, "a1w": a1w, "a1b": a1b, "a2w": a2w, "a2b": n1w})
        return (dx, g["n1w"], g["n1b"], g["n2w"], g["n2b"], g["m1w"], g["m1b"], g["m2w"], g["m2b"], g["esb"],
                g["a1w"], g["a1b"], g["a2w"], g["a2b"], None, None, None, None, None)


class PatchEmbedFnBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cw, cb, sw, sb, inp, inp_s, s_mean, s_std, u_mean, u_std, maps, const_h, sh, levels_reversed=False):
        a_s, a_u = ob.patch_embed_gather(inp, inp_s, s_mean, s_std, u_mean, u_std, maps, const_h, levels_reversed)
        n_s = a_s.shape[0]
        x = torch.empty((n_s + a_u.shape[0], cw.shape[0]), dtype=torch.bfloat16, device=inp.device)
        ob.linear(a_s, sh.get(sw, pad_k=128), sb, out=x[:n_s])
        ob.linear(a_u, sh.get(cw), cb, out=x[n_s:])
        ctx.save_for_backward(a_s, a_u, s_std, u_std)
        ctx.shapes, ctx.geom = (cw.shape, sw.shape), (inp.shape[-2], inp.shape[-1], bool(levels_reversed))
        ctx.sh, ctx.params = sh, (cw, sw)
        return x

    @staticmethod
    def backward(ctx, dx):
        a_s, a_u, s_std, u_std = ctx.saved_tensors
        n_s = a_s.shape[0]
        need = ctx.needs_input_grad
        dsw = dsb = dcw = dcb = None
        if any(need[:4]):
            dsw, dsb = ob.linear_wgrad(dx[:n_s], a_s)                 # (192, 128): columns 112.. are padding
            dcw, dcb = ob.linear_wgrad(dx[n_s:], a_u)
            k_s = ctx.shapes[1][1]
            dcw, dsw = dcw.reshape(ctx.shapes[0]), dsw[:, :k_s].reshape(ctx.shapes[1])
        d_in = d_in_s = None
        if need[4] or need[5]:
            # the raw fields asked for their gradient (reference layers.py:40-93 is plain autograd): dA for the columns with a field
            # behind them (bf16 operands, fp32 result), then the fp32 scatter adjoint of the gather, divided by the std
            LAT, LON, rev = ctx.geom
            cw, sw = ctx.params
            dx = dx.contiguous()
            da_s = ob.linear(dx[:n_s], ctx.sh.get_t(sw)[:64].contiguous(), out_dtype=torch.float32)
            da_u = ob.linear(dx[n_s:], ctx.sh.get_t(cw)[:160].contiguous(), out_dtype=torch.float32)
            d_in, d_in_s = ops.patch_embed_gather_bwd(da_s, da_u, s_std, u_std, LAT, LON, rev)
        return (dcw, dcb, dsw, dsb, d_in if need[4] else None, d_in_s if need[5] else None) + (None,) * 8


class DownSampleFnBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lw, nw, nb, geom, sh, skip_grad=None):
        # skip_grad: a one-slot list shared with PatchRecoverFnBF16 -- x is the skip connection (reference pangu_model.py:62, :81),
        # whose OTHER gradient (through the channel concat) that function leaves in the slot instead of handing it to autograd:
        # the backward below sums the two inside the down-sampling kernel (no elementwise add over the 200 MB)
        Z, H, W = geom
        g = ob.downsample_ln(x, nw, nb, Z, H, W)
        ctx.save_for_backward(x, g)
        ctx.geom, ctx.sh, ctx.params, ctx.skip_grad = geom, sh, (lw, nw), skip_grad
        if skip_grad is not None:
            skip_grad[1] = True                  # armed: this node's backward will consume the slot
        return ob.linear(g, sh.get(lw))

    @staticmethod
    def backward(ctx, dout):
        x, g = ctx.saved_tensors
        lw, nw = ctx.params
        Z, H, W = ctx.geom
        dout = dout.contiguous()
        dlw, _ = ob.linear_wgrad(dout, g, want_bias=False)
        dg = ob.linear(dout, ctx.sh.get_t(lw))
        add = None
        if ctx.skip_grad is not None:
            add, ctx.skip_grad[0] = ctx.skip_grad[0], None
        dx, dnw, dnb = ob.downsample_ln_bwd(dg, x, nw, Z, H, W, add=add)
        return dx, dlw, dnw, dnb, None, None, None


class UpSampleFnBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, l1w, l2w, nw, nb, geom, sh):
        Z, H2, W2, H = geom
        y = ob.linear(x, sh.get(l1w))
        g = ob.upsample_ln(y, nw, nb, Z, H2, W2, H)
        ctx.save_for_backward(x, y, g)
        ctx.geom, ctx.sh, ctx.params = geom, sh, (l1w, l2w, nw)
        return ob.linear(g, sh.get(l2w))

    @staticmethod
    def backward(ctx, dout):
        x, y, g = ctx.saved_tensors
        l1w, l2w, nw = ctx.params
        Z, H2, W2, H = ctx.geom
        dout = dout.contiguous()
        dl2w, _ = ob.linear_wgrad(dout, g, want_bias=False)
        dg = ob.linear(dout, ctx.sh.get_t(l2w))
        dy, dnw, dnb = ob.upsample_ln_bwd(dg, y, nw, Z, H2, W2, H)
        dl1w, _ = ob.linear_wgrad(dy, x, want_bias=False)
        dx = ob.linear(dy, ctx.sh.get_t(l1w))
        return dx, dl1w, dl2w, dnw, dnb, None, None


class PatchRecoverFnBF16(torch.autograd.Function):
    """reference layers.py:511-545 on the channel concat of pangu_model.py:81.  `skip` and `x` are the two (N, C) halves of ONE
    (N, 2C) buffer (the last blocks of layer 0 / layer 3 wrote them in place): no concat copy in the forward, and the backward
    hands each half its own CONTIGUOUS gradient (two N = C products instead of one N = 2C product whose halves would be
    row-strided views: the consumers' fast paths want dense rows)."""

    @staticmethod
    def forward(ctx, skip, x, cw, cb, sw, sb, geom, sh, skip_grad=None):
        ctx.skip_grad = skip_grad
        n_s, LAT, LON = geom
        N, C = skip.shape
        adjacent = (skip.stride() == (2 * C, 1) and x.stride() == (2 * C, 1) and x.data_ptr() == skip.data_ptr() + 2 * C
                    and skip.untyped_storage().data_ptr() == x.untyped_storage().data_ptr())
        cat = torch.as_strided(skip, (N, 2 * C), (2 * C, 1), skip.storage_offset()) if adjacent else torch.cat((skip, x), dim=-1)
        y_s = ob.linear(cat[:n_s], sh.get(sw), sb, out_dtype=torch.float32)
        y_u = ob.linear(cat[n_s:], sh.get(cw), cb, out_dtype=torch.float32)
        ctx.save_for_backward(cat)
        ctx.geom, ctx.sh, ctx.params = geom, sh, (cw, sw)
        return ops.patch_recover_scatter(y_u, y_s, LAT, LON)

    @staticmethod
    def backward(ctx, d_out, d_out_s):
        (cat,) = ctx.saved_tensors
        cw, sw = ctx.params
        n_s, LAT, LON = ctx.geom
        C = cat.shape[1] // 2
        dy_u, dy_s = ob.patch_recover_gather_bwd(d_out.contiguous(), d_out_s.contiguous())
        dcw, dcb = ob.linear_wgrad(dy_u, cat[n_s:])
        dsw, dsb = ob.linear_wgrad(dy_s, cat[:n_s])
        wt_s, wt_u = ctx.sh.get_t(sw), ctx.sh.get_t(cw)                    # (2C, 64), (2C, 160): rows = input channels
        d_skip = torch.empty((cat.shape[0], C), dtype=cat.dtype, device=cat.device)
        d_x = torch.empty_like(d_skip)
        for dst, rows in ((d_skip, slice(0, C)), (d_x, slice(C, 2 * C))):
            ob.linear(dy_s, wt_s[rows], out=dst[:n_s])
            ob.linear(dy_u, wt_u[rows], out=dst[n_s:])
        sg = ctx.skip_grad
        if sg is not None and sg[1] and ctx.needs_input_grad[0]:
            sg[0], d_skip = d_skip, None          # the down-sampling backward adds it in its own pass (DownSampleFnBF16)
            # (if autograd pruned that node -- `torch.autograd.grad(loss, inputs=[layer-3 parameters])` -- nothing consumes the
            # slot: it is emptied when this backward pass ends instead of pinning 200-400 MB until the next forward)
            torch.autograd.Variable._execution_engine.queue_callback(lambda sg=sg: sg.__setitem__(0, None))
        return d_skip, d_x, dcw.reshape(cw.shape), dcb, dsw.reshape(sw.shape), dsb, None, None, None


def forward_train(model, inp, inp_surface, statistics, maps, const_h, levels_reversed=False):
    """Autograd-enabled bf16 forward of the whole model (B looped; the reference is B = 1 per rank)."""
    sh = model._shadow
    s_mean, s_std, u_mean, u_std = statistics
    B = inp.shape[0]
    LAT, LON = inp.shape[-2], inp.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    dev = inp.device
    f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    s_mean, s_std = f32(s_mean).reshape(-1), f32(s_std).reshape(-1)
    u_mean, u_std = f32(u_mean).reshape(13, 5), f32(u_std).reshape(13, 5)
    maps_c, const_c = f32(maps).reshape(3, 4 * H4, LON), f32(const_h).reshape(13, LAT, LON)
    emb, rec, dn, up = model._input_layer, model._output_layer, model.downsample, model.upsample
    H2, W2 = (H4 + 1) // 2, W4 // 2

    def run_layer(layer, x, Z, H, W, out=None):
        last = len(layer.blocks) - 1
        for i, blk in enumerate(layer.blocks):
            att, dp = blk.attention, blk.drop_path
            s1 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
            s2 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
            x = EarthBlockFnBF16.apply(
                x, blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias, blk.linear.linear1.weight,
                blk.linear.linear1.bias, blk.linear.linear2.weight, blk.linear.linear2.bias, att.earth_specific_bias,
                att.linear1.weight, att.linear1.bias, att.linear2.weight, att.linear2.bias,
                (Z, H, W, att.head_number, i % 2 == 1), s1, s2, sh, (out,) if out is not None and i == last else None)
        return x

    outs, outs_s = [], []
    for b in range(B):
        x = PatchEmbedFnBF16.apply(emb.conv.weight, emb.conv.bias, emb.conv_surface.weight, emb.conv_surface.bias,
                                   inp[b].contiguous(), inp_surface[b].contiguous(), s_mean, s_std, u_mean, u_std, maps_c,
                                   const_c, sh, levels_reversed)
        # skip connection (reference pangu_model.py:81): layer 0 / layer 3 write their results straight into the two halves of
        # one (N, 2C) buffer -- no concat copy, as in the inference path
        Nn, Cc = x.shape
        cat = torch.empty((Nn, 2 * Cc), dtype=torch.bfloat16, device=dev)
        # the two halves as tensors that SHARE cat's storage without being autograd views of it (a view returned by a custom
        # Function whose base is written again -- the other half -- is refused by autograd)
        halves = [torch.empty(0, dtype=torch.bfloat16, device=dev).set_(cat.untyped_storage(), cat.storage_offset() + off, (Nn, Cc), (2 * Cc, 1))
                  for off in (0, Cc)]
        skip = run_layer(model.layers[0], x, 8, H4, W4, out=halves[0])
        skip_grad = [None, False]                 # [the concat path's gradient of `skip`, armed]: see DownSampleFnBF16
        x = DownSampleFnBF16.apply(skip, dn.linear.weight, dn.norm.weight, dn.norm.bias, (8, H4, W4), sh, skip_grad)
        x = run_layer(model.layers[1], x, 8, H2, W2)
        x = run_layer(model.layers[2], x, 8, H2, W2)
        x = UpSampleFnBF16.apply(x, up.linear1.weight, up.linear2.weight, up.norm.weight, up.norm.bias, (8, H2, W2, H4), sh)
        x = run_layer(model.layers[3], x, 8, H4, W4, out=halves[1])
        o, os_ = PatchRecoverFnBF16.apply(skip, x, rec.conv.weight, rec.conv.bias, rec.conv_surface.weight,
                                          rec.conv_surface.bias, (H4 * W4, LAT, LON), sh, skip_grad)
        outs.append(o)
        outs_s.append(os_)
    if B == 1:
        return outs[0].unsqueeze(0), outs_s[0].unsqueeze(0)
    return torch.stack(outs, 0), torch.stack(outs_s, 0)

"""torch.autograd.Function wrappers: one per reference layer, each a hand-scheduled chain of HIP kernels.

Per-sample 2-D token tensors (N, C).  Nothing of size (..,144,144) is saved: attention backward recomputes
the probabilities from q, k, v and the per-row log-sum-exp, so a whole training step keeps ~64 GB of fp32
activations and needs no block re-computation (the reference re-runs every block forward in backward,
models/layers.py:115-119).

Gradient of a projection y = a @ W^T + b:   da = dy @ W  (the forward GEMM with W^T),  dW, db = wgrad(dy, a).
"""
import torch

from . import ops


def _wt(w):
    """(out,in[,1]) weight -> contiguous (in,out): the `W` operand of ops.linear for the input gradient."""
    return w.reshape(w.shape[0], -1).t().contiguous()


class EarthBlockFn(torch.autograd.Function):
    """reference models/layers.py:183-253 (+ attention :360-421, Mlp :264-270) for one sample."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, n2w, n2b, m1w, m1b, m2w, m2b, esb, a1w, a1b, a2w, a2b, geom, s1, s2, dst=None):
        # dst: optional 1-tuple holding the (N, C) row-strided tensor the block writes its result into (one half of the
        # skip-concat buffer of reference pangu_model.py:81); wrapped so that autograd does not see a tensor argument
        out = dst[0] if dst else None
        Z, H, W, heads, shifted = geom
        ctx.geom, ctx.s1, ctx.s2 = geom, s1, s2
        saved = [x, n1w, n2w, m1w, m2w, esb, a1w, a1b, a2w]
        x1 = x
        if s1 != 0.0:
            qkv = ops.linear(x, a1w, a1b)
            o, lse = ops.window_attention(qkv, a1b, esb[0], Z, H, W, heads, shifted, want_lse=True)
            y = ops.linear(o, a2w, a2b)
            # (a dropped MLP branch -- s2 == 0 -- makes x1 the block's result: written straight into `out`, no copy afterwards)
            x1 = ops.ln_residual(y, x, n1w, n1b, branch_scale=s1, out=out if s2 == 0.0 else None)
            saved += [qkv, o, lse, y]
        if s2 != 0.0:
            pre = torch.empty((x.shape[0], m1w.shape[0]), dtype=x.dtype, device=x.device)
            h = ops.linear(x1, m1w, m1b, act=ops.ACT_GELU, aux=pre)
            m = ops.linear(h, m2w, m2b)
            x2 = ops.ln_residual(m, x1, n2w, n2b, out=out, branch_scale=s2)
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
        s1, s2 = ctx.s1, ctx.s2
        sv = list(ctx.saved_tensors)
        x, n1w, n2w, m1w, m2w, esb, a1w, a1b, a2w = sv[:9]
        rest = sv[9:]
        if s1 != 0.0:
            qkv, o, lse, y = rest[:4]
            rest = rest[4:]
        g = {k: None for k in ("n1w", "n1b", "n2w", "n2b", "m1w", "m1b", "m2w", "m2b", "esb", "a1w", "a1b", "a2w", "a2b")}
        dx1 = dout
        if s2 != 0.0:
            x1, pre, h, m = rest
            dm, g["n2w"], g["n2b"] = ops.ln_residual_bwd(dout, m, n2w, s2)
            g["m2w"], g["m2b"] = ops.linear_wgrad(dm, h)
            dpre = ops.linear(dm, _wt(m2w), None, act=ops.ACT_GELU_BWD, aux=pre)
            del dm
            g["m1w"], g["m1b"] = ops.linear_wgrad(dpre, x1)
            if dout.is_contiguous():      # residual gradient added in the GEMM epilogue (no extra pass over N x C)
                dx1 = ops.linear(dpre, _wt(m1w), act=ops.ACT_ADD, aux=dout)
            else:
                dx1 = ops.linear(dpre, _wt(m1w))
                dx1 += dout
            del dpre
        dx = dx1
        if s1 != 0.0:
            dy, g["n1w"], g["n1b"] = ops.ln_residual_bwd(dx1, y, n1w, s1)
            g["a2w"], g["a2b"] = ops.linear_wgrad(dy, o)
            do = ops.linear(dy, _wt(a2w))
            del dy
            dqkv, dqb_pad, desb = ops.window_attention_bwd(qkv, a1b, esb[0], o, lse, do, Z, H, W, heads, shifted,
                                                           desb_out=ops.grad_slot(esb))      # straight into the DP flat buffer
            del do
            g["esb"] = desb.unsqueeze(0)
            # linear1's bias gradient = column sums of dqkv + the pad-slot term already in dqb_pad: the kernel adds into that buffer
            g["a1w"], g["a1b"] = ops.linear_wgrad(dqkv, x, db_into=dqb_pad)
            if dx1.is_contiguous():
                dx = ops.linear(dqkv, _wt(a1w), act=ops.ACT_ADD, aux=dx1)
            else:
                dx = ops.linear(dqkv, _wt(a1w))
                dx += dx1
        elif not dx.is_contiguous():
            dx = dx.contiguous()
        ops.fill_dropped_grads(g, {"n1w": n1w, "n1b": n1w, "n2w": n2w, "n2b": n2w, "m1w": m1w, "m1b": m1w[:, 0], "m2w": m2w, "m2b": n2w,
                                   "esb": esb, "a1w": a1w, "a1b": a1b, "a2w": a2w, "a2b": n1w})
        return (dx, g["n1w"], g["n1b"], g["n2w"], g["n2b"], g["m1w"], g["m1b"], g["m2w"], g["m2b"], g["esb"],
                g["a1w"], g["a1b"], g["a2w"], g["a2b"], None, None, None, None)


class AttentionWindowsFn(torch.autograd.Function):
    """reference models/layers.py:360-421 (EarthAttention3D.forward taken on its own) on partitioned rows: xw (n_lon*types*144, C)
    in window-slot order, esb (1, types, heads, 144, 144), mask None | (n_lon, types, 144, 144) | (types, 144, 144)."""

    @staticmethod
    def forward(ctx, xw, w1, b1, w2, b2, esb, mask, geom):
        n_lon, types, heads = geom
        qkv = ops.linear(xw, w1, b1)
        o = ops.attention_windows(qkv, esb[0], mask, n_lon, types, heads)
        ctx.save_for_backward(xw, qkv, o, w1, w2, esb, *([mask] if mask is not None else []))
        ctx.geom = geom
        return ops.linear(o, w2, b2)

    @staticmethod
    def backward(ctx, dy):
        xw, qkv, o, w1, w2, esb, *rest = ctx.saved_tensors
        mask = rest[0] if rest else None
        n_lon, types, heads = ctx.geom
        dy = dy.contiguous()
        dw2, db2 = ops.linear_wgrad(dy, o)
        do = ops.linear(dy, _wt(w2))
        dqkv, desb = ops.attention_windows_bwd(qkv, esb[0], mask, do, n_lon, types, heads)
        dw1, db1 = ops.linear_wgrad(dqkv, xw)
        dx = ops.linear(dqkv, _wt(w1))
        return dx, dw1, db1, dw2, db2, desb.unsqueeze(0), None, None


class MlpFn(torch.autograd.Function):
    """reference models/layers.py:264-270 (Mlp.forward taken on its own: linear1 -> exact-erf GELU -> linear2) on (M, C) rows."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        pre = torch.empty((x.shape[0], w1.shape[0]), dtype=x.dtype, device=x.device)
        h = ops.linear(x, w1, b1, act=ops.ACT_GELU, aux=pre)
        ctx.save_for_backward(x, pre, h, w1, w2)
        return ops.linear(h, w2, b2)

    @staticmethod
    def backward(ctx, dm):
        x, pre, h, w1, w2 = ctx.saved_tensors
        dm = dm.contiguous()
        dw2, db2 = ops.linear_wgrad(dm, h)
        dpre = ops.linear(dm, _wt(w2), None, act=ops.ACT_GELU_BWD, aux=pre)
        dw1, db1 = ops.linear_wgrad(dpre, x)
        return ops.linear(dpre, _wt(w1)), dw1, db1, dw2, db2


def refuse_constant_grads(maps, const_h, statistics=()):
    """The constant operands of the patch embedding (maps, const_h, the normalisation statistics) get no gradient from this
    build; asking for one must not return None silently."""
    for name, t in (("maps", maps), ("const_h", const_h)) + tuple((f"statistics[{i}]", t) for i, t in enumerate(statistics)):
        if torch.is_tensor(t) and t.requires_grad:
            raise RuntimeError(f"PanguModel (MI355X build): {name}.requires_grad is set, but gradients with respect to the constant maps, "
                               "const_h and the normalisation statistics are not implemented (input / input_surface are)")


class PatchEmbedFn(torch.autograd.Function):
    """reference models/layers.py:40-93 for one sample.  The raw fields get their gradient when they ask for it
    (`input.requires_grad_()`, plain autograd in the reference): d_input = scatter-adjoint of the gather of (dx @ W) / std."""

    @staticmethod
    def forward(ctx, cw, cb, sw, sb, inp, inp_s, s_mean, s_std, u_mean, u_std, maps, const_h, levels_reversed=False):
        a_s, a_u = ops.patch_embed_gather(inp, inp_s, s_mean, s_std, u_mean, u_std, maps, const_h, levels_reversed)
        n_s = a_s.shape[0]
        x = torch.empty((n_s + a_u.shape[0], cw.shape[0]), dtype=torch.float32, device=inp.device)
        ops.linear(a_s, sw, sb, out=x[:n_s])
        ops.linear(a_u, cw, cb, out=x[n_s:])
        ctx.save_for_backward(a_s, a_u, cw, sw, s_std, u_std)
        ctx.shapes, ctx.geom = (cw.shape, sw.shape), (inp.shape[-2], inp.shape[-1], bool(levels_reversed))
        return x

    @staticmethod
    def backward(ctx, dx):
        a_s, a_u, cw, sw, s_std, u_std = ctx.saved_tensors
        n_s = a_s.shape[0]
        dx = dx.contiguous()
        need = ctx.needs_input_grad
        dsw = dsb = dcw = dcb = None
        if any(need[:4]):
            dsw, dsb = ops.linear_wgrad(dx[:n_s], a_s)
            dcw, dcb = ops.linear_wgrad(dx[n_s:], a_u)
            dcw, dsw = dcw.reshape(ctx.shapes[0]), dsw.reshape(ctx.shapes[1])
        d_in = d_in_s = None
        if need[4] or need[5]:
            LAT, LON, rev = ctx.geom
            # only the A-matrix columns with a field behind them: the first 64 of 112 (surface) / 160 of 192 (upper)
            da_s = ops.linear(dx[:n_s], _wt(sw)[:64].contiguous())
            da_u = ops.linear(dx[n_s:], _wt(cw)[:160].contiguous())
            d_in, d_in_s = ops.patch_embed_gather_bwd(da_s, da_u, s_std, u_std, LAT, LON, rev)
        return (dcw, dcb, dsw, dsb, d_in if need[4] else None, d_in_s if need[5] else None) + (None,) * 7


class DownSampleFn(torch.autograd.Function):
    """reference models/layers.py:432-459 for one sample."""

    @staticmethod
    def forward(ctx, x, lw, nw, nb, geom, skip_grad=None):
        # skip_grad: one-slot list shared with PatchRecoverHalvesFn (the skip connection's other gradient, summed inside the
        # down-sampling backward kernel instead of by autograd's elementwise add): see autograd_bf16.DownSampleFnBF16
        Z, H, W = geom
        g = ops.downsample_ln(x, nw, nb, Z, H, W)
        ctx.save_for_backward(x, g, lw, nw)
        ctx.geom, ctx.skip_grad = geom, skip_grad
        if skip_grad is not None:
            skip_grad[1] = True
        return ops.linear(g, lw)

    @staticmethod
    def backward(ctx, dout):
        x, g, lw, nw = ctx.saved_tensors
        Z, H, W = ctx.geom
        dout = dout.contiguous()
        dlw, _ = ops.linear_wgrad(dout, g, want_bias=False)
        dg = ops.linear(dout, _wt(lw))
        add = None
        if ctx.skip_grad is not None:
            add, ctx.skip_grad[0] = ctx.skip_grad[0], None
        dx, dnw, dnb = ops.downsample_ln_bwd(dg, x, nw, Z, H, W, add=add)
        return dx, dlw, dnw, dnb, None, None


class UpSampleFn(torch.autograd.Function):
    """reference models/layers.py:474-499 for one sample."""

    @staticmethod
    def forward(ctx, x, l1w, l2w, nw, nb, geom):
        Z, H2, W2, H = geom
        y = ops.linear(x, l1w)
        g = ops.upsample_ln(y, nw, nb, Z, H2, W2, H)
        ctx.save_for_backward(x, y, g, l1w, l2w, nw)
        ctx.geom = geom
        return ops.linear(g, l2w)

    @staticmethod
    def backward(ctx, dout):
        x, y, g, l1w, l2w, nw = ctx.saved_tensors
        Z, H2, W2, H = ctx.geom
        dout = dout.contiguous()
        dl2w, _ = ops.linear_wgrad(dout, g, want_bias=False)
        dg = ops.linear(dout, _wt(l2w))
        dy, dnw, dnb = ops.upsample_ln_bwd(dg, y, nw, Z, H2, W2, H)
        dl1w, _ = ops.linear_wgrad(dy, x, want_bias=False)
        dx = ops.linear(dy, _wt(l1w))
        return dx, dl1w, dl2w, dnw, dnb, None


class PatchRecoverFn(torch.autograd.Function):
    """reference models/layers.py:511-545 for one sample: x (N, C) -> (5,13,LAT,LON), (4,LAT,LON)."""

    @staticmethod
    def forward(ctx, x, cw, cb, sw, sb, geom):
        n_s, LAT, LON = geom
        y_s = ops.linear(x[:n_s], sw, sb)
        y_u = ops.linear(x[n_s:], cw, cb)
        ctx.save_for_backward(x, cw, sw)
        ctx.geom = geom
        return ops.patch_recover_scatter(y_u, y_s, LAT, LON)

    @staticmethod
    def backward(ctx, d_out, d_out_s):
        x, cw, sw = ctx.saved_tensors
        n_s, LAT, LON = ctx.geom
        dy_u, dy_s = ops.patch_recover_gather_bwd(d_out.contiguous(), d_out_s.contiguous())
        dcw, dcb = ops.linear_wgrad(dy_u, x[n_s:])
        dsw, dsb = ops.linear_wgrad(dy_s, x[:n_s])
        dx = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        ops.linear(dy_s, _wt(sw), out=dx[:n_s])
        ops.linear(dy_u, _wt(cw), out=dx[n_s:])
        return dx, dcw.reshape(cw.shape), dcb, dsw.reshape(sw.shape), dsb, None


class PatchRecoverHalvesFn(torch.autograd.Function):
    """The same layer on the channel concat of reference pangu_model.py:81 given as its two (N, C) halves, which are the two
    halves of ONE (N, 2C) buffer (layer 0 / layer 3 wrote them in place): no concat copy in the forward, and each half gets
    its own DENSE gradient in the backward (two N = C products instead of row-strided views of one N = 2C product)."""

    @staticmethod
    def forward(ctx, skip, x, cw, cb, sw, sb, geom, skip_grad=None):
        ctx.skip_grad = skip_grad
        n_s, LAT, LON = geom
        N, C = skip.shape
        assert skip.stride() == (2 * C, 1) and x.stride() == (2 * C, 1) and x.data_ptr() == skip.data_ptr() + 4 * C
        cat = torch.as_strided(skip, (N, 2 * C), (2 * C, 1), skip.storage_offset())
        y_s = ops.linear(cat[:n_s], sw, sb)
        y_u = ops.linear(cat[n_s:], cw, cb)
        ctx.save_for_backward(cat, cw, sw)
        ctx.geom = geom
        return ops.patch_recover_scatter(y_u, y_s, LAT, LON)

    @staticmethod
    def backward(ctx, d_out, d_out_s):
        cat, cw, sw = ctx.saved_tensors
        n_s, LAT, LON = ctx.geom
        C = cat.shape[1] // 2
        dy_u, dy_s = ops.patch_recover_gather_bwd(d_out.contiguous(), d_out_s.contiguous())
        dcw, dcb = ops.linear_wgrad(dy_u, cat[n_s:])
        dsw, dsb = ops.linear_wgrad(dy_s, cat[:n_s])
        wt_s, wt_u = _wt(sw), _wt(cw)                                      # (2C, 64), (2C, 160): rows = input channels
        d_skip = torch.empty((cat.shape[0], C), dtype=torch.float32, device=cat.device)
        d_x = torch.empty_like(d_skip)
        for dst, rows in ((d_skip, slice(0, C)), (d_x, slice(C, 2 * C))):
            ops.linear(dy_s, wt_s[rows], out=dst[:n_s])
            ops.linear(dy_u, wt_u[rows], out=dst[n_s:])
        sg = ctx.skip_grad
        if sg is not None and sg[1] and ctx.needs_input_grad[0]:
            sg[0], d_skip = d_skip, None          # the down-sampling backward adds it in its own pass (DownSampleFn)
            # (if autograd pruned that node -- `torch.autograd.grad(loss, inputs=[layer-3 parameters])` -- nothing consumes the
            # slot: it is emptied when this backward pass ends instead of pinning 200-400 MB until the next forward)
            torch.autograd.Variable._execution_engine.queue_callback(lambda sg=sg: sg.__setitem__(0, None))
        return d_skip, d_x, dcw.reshape(cw.shape), dcb, dsw.reshape(sw.shape), dsb, None, None

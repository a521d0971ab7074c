"""bf16 inference forward of the whole model (BASELINE configs[2]/[4]): bf16 activations + bf16 weight shadows,
fp32 LayerNorm / softmax / accumulation, fp32 output fields.  Same launch sequence as the fp32 path."""
import os

import torch

from . import ops, ops_bf16 as ob


def _stamp(p):
    """ops.param_stamp: (optimizer epoch, _version, data_ptr, shape, device)."""
    return ops.param_stamp(p)


class WeightShadow:
    """bf16 copies of the projection weights and bias tables (and the transposed / packed weight images of the backward GEMMs
    and the fused MLP kernel), re-made when a parameter's stamp changes (ops.param_stamp).  Never pickled / deep-copied with
    the model (PanguModel.__getstate__).

    After an optimizer step EVERY copy is stale.  Each copy made from contiguous fp32 parameters is also recorded as a job
    (cast / transposed cast / gather through the MLP pack index) of `pangu_shadow_refresh_bf16`; the first lookup of a new
    optimizer epoch re-makes all recorded copies IN PLACE with one launch over a device-resident job table (1.56 GB of HBM
    traffic, ~0.3 ms) instead of ~240 small torch launches (1.6 ms of GPU time, 10 ms of host time per training step)."""

    def __init__(self):
        ops.require_epoch_hook()
        self.cache = {}          # key -> (stamp, tensor)
        self.jobs = {}           # key -> (mode, params, dst, idx, n0, n1, blocks, signature of the params)
        self.table = None        # (device int64 job table, keys in table order, total blocks, keys left out)
        self.bulk_epoch = ops._weights_epoch[0]
        self.makers = {}         # key -> (params, make) of the copies that are NOT bulk-refresh jobs (padded casts): refresh_in_place
        self.fresh = {}          # key -> (epoch, param _version, param address) when an optimizer wrote that copy itself (train.HipAdam): skipped by the next refresh

    def clear(self):
        self.cache.clear()
        self.jobs.clear()
        self.makers.clear()
        self.fresh.clear()
        self.table = None

    def plain_image(self, p):
        """The recorded plain bf16 cast of parameter `p` (None when there is none or it no longer matches `p`): an optimizer that
        updates `p` may write this image in the same pass (train.HipAdam) and call `mark_fresh(p)`."""
        j = self.jobs.get(id(p))
        if j is None or j[0] != 0 or self._sig(j[1]) != j[7] or self.cache.get(id(p), (None, None))[1] is not j[2]:
            return None
        return j[2]

    def mark_fresh(self, p):
        """`plain_image(p)` was just re-written from the updated `p` by the optimizer step in progress (epoch = the current one;
        the post-step hook advances it): the refresh of the NEXT epoch -- and only that one -- leaves it out."""
        # (the parameter's `_version` and address at this moment: HipAdam's raw-pointer kernel bumps neither, so an in-place edit
        # before the next forward -- load_state_dict(best), an EMA swap, a weight clamp -- is what changes them, and the image
        # written here is then stale like every other copy)
        self.fresh[id(p)] = (ops._weights_epoch[0], p._version, p.data_ptr())

    @staticmethod
    def _sig(params):
        return tuple((p.data_ptr(), tuple(p.shape), p.device, p.dtype) for p in params)

    def _record(self, key, mode, params, dst, idx=None):
        """Remember how `dst` derives from `params` (all contiguous fp32 on dst's device), for the bulk refresh."""
        if not dst.is_cuda or not dst.is_contiguous() or dst.dtype != torch.bfloat16:
            return
        if any(p.dtype != torch.float32 or not p.is_contiguous() or p.device != dst.device for p in params):
            return
        if mode == 0:
            n0, n1 = params[0].numel(), 0
            blocks = (n0 + 4095) // 4096
        elif mode == 1:
            n0 = params[0].shape[0]
            n1 = params[0].numel() // n0
            blocks = ((n0 + 63) // 64) * ((n1 + 63) // 64)
        else:
            n0, n1 = params[0].numel(), dst.numel()
            blocks = (n1 + 4095) // 4096
        if blocks == 0 or dst.numel() != (n1 if mode == 2 else params[0].numel()):
            return
        self.jobs[key] = (mode, params, dst, idx, n0, n1, blocks, self._sig(params))
        self.table = None

    def _bulk_refresh(self):
        """One launch re-makes every recorded copy whose parameters still live where they did; the others are dropped and
        re-made lazily by their next lookup."""
        for key in [k for k, j in self.jobs.items() if self._sig(j[1]) != j[7] or self.cache.get(k, (None, None))[1] is not j[2]]:
            del self.jobs[key]
            self.cache.pop(key, None)
            self.table = None
        if not self.jobs:
            return
        from . import _lib
        # copies an optimizer wrote itself during the ONE step since the last epoch (exactly one: any other optimizer step in
        # between could have touched the parameter again)
        ep = ops._weights_epoch[0]
        skip = frozenset(k for k, (e, ver, ptr) in self.fresh.items()
                         if e == ep - 1 and k in self.jobs and self.jobs[k][1][0]._version == ver and self.jobs[k][1][0].data_ptr() == ptr)
        self.fresh.clear()
        if self.table is None or self.table[3] != skip:
            rows, keys, first = [], [], 0
            for key, (mode, params, dst, idx, n0, n1, blocks, _) in self.jobs.items():
                if key in skip:
                    continue
                rows.append([params[0].data_ptr(), params[1].data_ptr() if len(params) > 1 else 0, dst.data_ptr(),
                             idx.data_ptr() if idx is not None else 0, n0, n1, mode, first])
                keys.append(key)
                first += blocks
            rows.append([0, 0, 0, 0, 0, 0, 0, first])
            dev = next(iter(self.jobs.values()))[2].device
            self.table = (torch.tensor(rows, dtype=torch.int64).to(dev), keys, first, skip)
        table, keys, total, _ = self.table
        if keys:
            _lib.check(_lib.load().pangu_shadow_refresh_bf16(ob._stream(table), table.data_ptr(), len(keys), total), "shadow_refresh_bf16")
        for key in list(keys) + list(skip):
            self.cache[key] = (tuple(_stamp(p) for p in self.jobs[key][1]), self.jobs[key][2])

    def _lookup(self, key, params, make, mode=None, idx=None):
        ep = ops._weights_epoch[0]
        if self.bulk_epoch != ep:
            self.bulk_epoch = ep
            if self.jobs:
                self._bulk_refresh()
        stamp = tuple(_stamp(p) for p in params)
        hit = self.cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        w = make()
        self.cache[key] = (stamp, w)
        if self.jobs.pop(key, None) is not None:
            self.table = None
        if mode is not None:
            self._record(key, mode, params, w, idx() if callable(idx) else idx)
        if key not in self.jobs:
            self.makers[key] = (params, make)
        else:
            self.makers.pop(key, None)
        return w

    def refresh_in_place(self):
        """Re-make EVERY cached copy now, INTO the tensors it already lives in -- for a captured training step
        (train.GraphedTrainStep), whose graph holds the shadows' addresses and runs no Python that could notice a new optimizer
        epoch: call after `optimizer.step()`.  The recorded jobs take the one-launch bulk refresh, the few others (padded casts)
        an in-place copy."""
        self.bulk_epoch = ops._weights_epoch[0]
        if self.jobs:
            self._bulk_refresh()
        with torch.no_grad():
            for key, (params, make) in list(self.makers.items()):
                hit = self.cache.get(key)
                if hit is None or key in self.jobs:
                    continue
                hit[1].copy_(make())
                self.cache[key] = (tuple(_stamp(p) for p in params), hit[1])

    def get(self, p, pad_k=None):
        def make():
            w = p.detach().reshape(p.shape[0], -1) if p.dim() == 3 else p.detach()
            if p.dim() == 5:
                w = w[0]
            if pad_k is not None and w.shape[1] < pad_k:
                w = torch.nn.functional.pad(w, (0, pad_k - w.shape[1]))
            return w.to(torch.bfloat16).contiguous()
        padded = pad_k is not None and p.dim() >= 2 and p.numel() // p.shape[0] < pad_k
        return self._lookup(id(p), (p,), make, mode=None if padded else 0)

    def get_t(self, p):
        """Transposed bf16 shadow (in, out): the `W` operand of the input-gradient GEMM dA = dC @ W."""
        return self._lookup(("t", id(p)), (p,),
                            lambda: p.detach().reshape(p.shape[0], -1).t().to(torch.bfloat16).contiguous(), mode=1)

    def get_mlp(self, w1, w2):
        """Packed chunk image of an Mlp's two weights for the fused MLP kernel (ops_bf16.pack_mlp_weights)."""
        return self._lookup(("mlp", id(w1), id(w2)), (w1, w2), lambda: ob.pack_mlp_weights(w1.detach(), w2.detach()),
                            mode=2, idx=lambda: ob.mlp_pack_index32(w1.shape[1], w1.device))


_FUSE_LN = os.environ.get("PANGU_BF16_FUSE_LN", "1") != "0"      # A/B knob: 0 = separate GEMM + LN-residual launches
_FUSE_QKV = os.environ.get("PANGU_BF16_FUSE_QKV", "1") != "0"    # A/B knob: 0 = QKV projection as its own GEMM launch
_FUSE_MLP = os.environ.get("PANGU_BF16_FUSE_MLP", "1") != "0"    # A/B knob: 0 = MLP-up, MLP-down(+LN) as separate launches


def _block(blk, sh, x, Z, H, W, roll, out=None):
    """x (N,C) bf16 -> (N,C) bf16.  DropPath (reference layers.py:250-251) is the identity in eval(); in train() mode under
    no_grad each branch draws its per-sample keep factor like the fp32 path (a dropped branch is not computed)."""
    att = blk.attention
    dp = blk.drop_path
    s1 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
    s2 = dp.sample_scale(blk.training) if hasattr(dp, "sample_scale") else 1.0
    C = x.shape[1]
    # projection + post-norm residual in one launch (the branch never round-trips HBM); C = 384: the 8-wave 128x384 tile
    # loses what the fusion saves (measured), so stage 1/2 keeps the separate launches
    ok = _FUSE_LN and x.is_contiguous() and s1 == 1.0 and s2 == 1.0
    fuse = ok and C == 192
    fuse_proj = fuse or (ok and C == 384)      # C = 384: the attention projection only (-1.2 % on the forward); MLP-down: see below
    fuse_mlp = fuse
    if s1 != 0.0:
        if _FUSE_QKV and C in (192, 384):
            # QKV projection inside the attention kernel: the (N, 3C) qkv tensor never reaches HBM
            o = ob.window_attention_qkv(x, sh.get(att.linear1.weight), att.linear1.bias, sh.get(att.earth_specific_bias),
                                        Z, H, W, att.head_number, roll)
        else:
            qkv = ob.linear(x, sh.get(att.linear1.weight), att.linear1.bias)
            o = ob.window_attention(qkv, sh.get(att.linear1.bias), sh.get(att.earth_specific_bias), Z, H, W,
                                    att.head_number, roll)
        if fuse_proj:
            x1 = ob.linear_ln_residual(o, sh.get(att.linear2.weight), att.linear2.bias, x, blk.norm1.weight, blk.norm1.bias)
        else:
            y = ob.linear(o, sh.get(att.linear2.weight), att.linear2.bias)
            x1 = ob.ln_residual(y, x, blk.norm1.weight, blk.norm1.bias, branch_scale=s1)
    else:
        x1 = x
    if s2 == 0.0:
        if out is not None:
            out.copy_(x1)
            return out
        return x1
    if _FUSE_MLP and C in (192, 384):
        # whole MLP branch + LayerNorm + residual in one launch: the (N, 4C) hidden activation never reaches HBM
        return ob.mlp_ln_residual(x1, sh.get_mlp(blk.linear.linear1.weight, blk.linear.linear2.weight),
                                  blk.linear.linear1.bias, blk.linear.linear2.bias, blk.norm2.weight, blk.norm2.bias,
                                  out=out, branch_scale=s2)
    h = ob.linear(x1, sh.get(blk.linear.linear1.weight), blk.linear.linear1.bias, act=ob.ACT_GELU)
    if fuse_mlp:
        return ob.linear_ln_residual(h, sh.get(blk.linear.linear2.weight), blk.linear.linear2.bias, x1, blk.norm2.weight,
                                     blk.norm2.bias, out=out)
    m = ob.linear(h, sh.get(blk.linear.linear2.weight), blk.linear.linear2.bias)
    return ob.ln_residual(m, x1, blk.norm2.weight, blk.norm2.bias, out=out, branch_scale=s2)


def _layer(layer, sh, x, Z, H, W, out=None):
    n = len(layer.blocks)
    for i, blk in enumerate(layer.blocks):
        x = _block(blk, sh, x, Z, H, W, i % 2 == 1, out=out if i == n - 1 else None)
    return x


def forward(model, inp, inp_surface, statistics, maps, const_h, levels_reversed=False):
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
    emb, rec = model._input_layer, model._output_layer
    outs, outs_s = [], []
    n_s = H4 * W4
    N = 8 * n_s
    C = emb.conv.weight.shape[0]
    for b in range(B):
        a_s, a_u = ob.patch_embed_gather(inp[b].contiguous(), inp_surface[b].contiguous(), s_mean, s_std, u_mean, u_std,
                                         maps_c, const_c, levels_reversed)
        x = torch.empty((N, C), dtype=torch.bfloat16, device=dev)
        ob.linear(a_s, sh.get(emb.conv_surface.weight, pad_k=128), emb.conv_surface.bias, out=x[:n_s])
        ob.linear(a_u, sh.get(emb.conv.weight), emb.conv.bias, out=x[n_s:])
        cat = torch.empty((N, 2 * C), dtype=torch.bfloat16, device=dev)
        skip = _layer(model.layers[0], sh, x, 8, H4, W4, out=cat[:, :C])
        g = ob.downsample_ln(skip, model.downsample.norm.weight, model.downsample.norm.bias, 8, H4, W4)
        x = ob.linear(g, sh.get(model.downsample.linear.weight))
        H2, W2 = (H4 + 1) // 2, W4 // 2
        x = _layer(model.layers[1], sh, x, 8, H2, W2)
        x = _layer(model.layers[2], sh, x, 8, H2, W2)
        y = ob.linear(x, sh.get(model.upsample.linear1.weight))
        g = ob.upsample_ln(y, model.upsample.norm.weight, model.upsample.norm.bias, 8, H2, W2, H4)
        x = ob.linear(g, sh.get(model.upsample.linear2.weight))
        _layer(model.layers[3], sh, x, 8, H4, W4, out=cat[:, C:])
        y_s = ob.linear(cat[:n_s], sh.get(rec.conv_surface.weight), rec.conv_surface.bias, out_dtype=torch.float32)
        y_u = ob.linear(cat[n_s:], sh.get(rec.conv.weight), rec.conv.bias, out_dtype=torch.float32)
        o, os_ = ops.patch_recover_scatter(y_u, y_s, LAT, LON)
        outs.append(o)
        outs_s.append(os_)
    if B == 1:                                   # no 286 MB stack copy for the usual single sample
        return outs[0].unsqueeze(0), outs_s[0].unsqueeze(0)
    return torch.stack(outs, 0), torch.stack(outs_s, 0)

"""PanguModel — drop-in for the reference's models/pangu_model.py:8-87 on MI355X.

Same constructor (`depths, num_heads, dims, patch_size, device`), same 223 state_dict keys/shapes (the
onnx2torch layout of keys_all.csv), same `forward(input, input_surface, statistics, maps, const_h)`
returning `(output, output_surface)` in normalised units, `.device`, and a plain nn.Module tree that
survives `copy.deepcopy` / pickling.  The arithmetic runs on hand-written gfx950 kernels (C ABI in
include/pangu_hip.h); calling it with CPU tensors raises — there is no fallback path.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from .layers import (DownSample, EarthSpecificLayer, PatchEmbedding_pretrain, PatchRecovery_pretrain, UpSample,
                     _trunc_normal_)


def compact_bias_stamp(p):
    """What a compact bias table was derived from (ops.param_stamp: optimizer epoch, `_version`, storage, shape, device)."""
    return ops.param_stamp(p)


class _EvalRecomputeFn(torch.autograd.Function):
    """`model.eval()` called WITH gradients enabled -- the reference's own `test()` does exactly that (models/pangu_sample.py:197-202:
    no `torch.no_grad()`), and nothing ever calls backward there.  The forward runs the INFERENCE kernels under no_grad and keeps
    nothing but its inputs (no 30 / 66 GB of saved activations, the fused inference launches instead of the activation-saving
    training ones); if a backward does arrive, it re-runs the forward on the autograd path and differentiates that: whole-model
    activation checkpointing -- what the reference does per block on every call (layers.py:115-119, `use_checkpoint` is always on).
    Stochastic depth is off in eval(), so the two forwards see the same function.  The two fields go through save_for_backward:
    overwriting them in place between forward and backward (a rollout's static buffers) raises instead of differentiating a
    different function; they get their gradients when they ask for them."""

    @staticmethod
    def forward(ctx, model, consts, inp, inp_s, *params):
        # consts = (statistics, maps, const_h, want_bf16, levels_reversed): the compute dtype is decided ONCE, here (an enclosing
        # autocast is not active any more when the backward runs)
        ctx.model, ctx.consts = model, consts
        ctx.save_for_backward(inp, inp_s)
        statistics, maps, const_h, want_bf16, rev = consts
        with torch.no_grad():
            return model._forward_dispatch(inp, inp_s, statistics, maps, const_h, want_bf16, False, rev)

    @staticmethod
    def backward(ctx, d_out, d_out_s):
        model = ctx.model
        inp, inp_s = ctx.saved_tensors
        need_in, need = ctx.needs_input_grad[2:4], ctx.needs_input_grad[4:]
        statistics, maps, const_h, want_bf16, rev = ctx.consts
        fields = [t.detach().requires_grad_(n) for t, n in zip((inp, inp_s), need_in)]
        wrt = [t for t, n in zip(fields, need_in) if n] + [p for p, n in zip(model.parameters(), need) if n]
        with torch.enable_grad():
            out, out_s = model._forward_dispatch(fields[0], fields[1], statistics, maps, const_h, want_bf16, True, rev)
        grads = iter(torch.autograd.grad((out, out_s), wrt, (d_out, d_out_s), allow_unused=True))
        g_in = tuple(next(grads) if n else None for n in need_in)
        return (None, None) + g_in + tuple(next(grads) if n else None for n in need)


class PanguModel(nn.Module):
    def __init__(self, depths=[2, 6, 6, 2], num_heads=[6, 12, 12, 6], dims=[192, 384, 384, 192],
                 patch_size=(2, 4, 4), device=None):
        super().__init__()
        self.device = device
        self._input_layer = PatchEmbedding_pretrain(patch_size, dims[0])
        self.downsample = DownSample(dims[0])
        dpr = [x.item() for x in torch.linspace(0, 0.2, sum(depths))]          # reference pangu_model.py:19
        self.num_layers = len(depths)
        layer_list = OrderedDict()
        for i in range(self.num_layers):
            layer_list[f"EarthSpecificLayer{i}"] = EarthSpecificLayer(
                depth=depths[i], dim=dims[i], drop_path_ratio_list=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                heads=num_heads[i], use_checkpoint=False, device=device)
        self.layers = nn.Sequential(layer_list)
        self.upsample = UpSample(dims[-2], dims[-1])
        self._output_layer = PatchRecovery_pretrain(dims[-2])
        self.apply(self._init_weights)
        # optional default constants (NOT in state_dict): lets callers use forward(input, input_surface)
        self.register_buffer("_c_surface_mean", None, persistent=False)
        self.register_buffer("_c_surface_std", None, persistent=False)
        self.register_buffer("_c_upper_mean", None, persistent=False)
        self.register_buffer("_c_upper_std", None, persistent=False)
        self.register_buffer("_c_maps", None, persistent=False)
        self.register_buffer("_c_const_h", None, persistent=False)
        self.compute_dtype = torch.float32
        self._shadow = None
        self._compact_bias = False
        # eval() forward with gradients enabled: "recompute" (default: inference kernels now, the autograd forward again only if a
        # backward arrives -- _EvalRecomputeFn) or "save" (the activation-saving training forward at once: cheaper when every eval-mode
        # forward IS followed by a backward, e.g. train.GraphedTrainStep, deterministic fine-tuning with stochastic depth off)
        self.eval_grad_mode = "recompute"

    def set_compute_dtype(self, dtype):
        """torch.float32 (default; parity <= 1e-3 with the reference) or torch.bfloat16 (inference: bf16 activations and
        weight shadows, fp32 LayerNorm/softmax/accumulation).  bf16 is also selected by an enclosing
        `torch.autocast("cuda", dtype=torch.bfloat16)` — the switch the reference leaves commented out at
        models/pangu_sample.py:46-47."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        self.invalidate_shadows()
        return self

    def invalidate_shadows(self):
        """Drop the bf16 weight shadows / packed weight images (they are re-made on the next bf16 forward).  Called by
        load_state_dict, .to()/.cuda()/.half() (`_apply`) and set_compute_dtype; call it yourself after editing weights
        through `param.data` in place (`p.data.copy_(..)`), which leaves no trace the cache could check."""
        if self._shadow is not None:
            self._shadow.clear()
        for m in self.modules():              # compact bias tables are derived from the parameters too
            if hasattr(m, "_esb_compact"):
                m._esb_compact = None

    def use_compact_bias(self, enable=True):
        """fp32 INFERENCE on the paper's compact Earth-specific bias (reference layers.py:306-357, :384-391: a
        (3312, types, heads) table gathered through `position_index`; this model, like the reference, keeps the EXPANDED
        (1, types, heads, 144, 144) parameter of the ONNX export).  Each block's expanded parameter is folded back to the
        compact table once (weights.compact_bias_table) and the attention kernel gathers from it: 10 MB instead of 62 MB
        of bias per block, results bit-identical.  Raises ValueError when a block's parameter is not an expansion of a
        compact table (e.g. the reference's random initialisation of the expanded tensor): nothing is approximated.
        The tables are dropped whenever the weights may have changed (invalidate_shadows) and rebuilt on the next forward;
        training and the bf16 path always read the expanded parameter."""
        self._compact_bias = bool(enable)
        if not enable:
            for m in self.modules():
                if hasattr(m, "_esb_compact"):
                    m._esb_compact = None
            return self
        self._build_compact_bias()
        return self

    def _build_compact_bias(self):
        from . import weights
        for name, m in self.named_modules():
            p = getattr(m, "earth_specific_bias", None)
            if p is None:
                continue
            stamp = compact_bias_stamp(p)
            if getattr(m, "_esb_compact", None) is not None and getattr(m, "_esb_compact_stamp", None) == stamp:
                continue
            m._esb_compact = None
            table = weights.compact_bias_table(p.detach())
            if table is None:
                raise ValueError(f"{name}.earth_specific_bias is not an expansion of a compact (3312, types, heads) table: "
                                 "compact-bias inference would change the result; call use_compact_bias(False)")
            m._esb_compact, m._esb_compact_stamp = table, stamp

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_shadows()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_shadow", None) is not None:
            self._shadow.clear()
        for m in self.modules():              # derived from the parameters as well (device / dtype moves)
            if getattr(m, "_esb_compact", None) is not None:
                m._esb_compact = None
        return out

    def __getstate__(self):
        """Pickling (`torch.save(model)`) and `copy.deepcopy(model)` (reference models/pangu_sample.py:162-164) carry the
        module tree only, never the ~0.5 GB of derived bf16 shadows (keyed by id() of the ORIGINAL parameters)."""
        state = self.__dict__.copy()
        state["_shadow"] = None
        return state

    def _init_weights(self, m):                                                # reference pangu_model.py:41-48
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _assert_plain_children(self):
        """See layers.assert_plain_tree: wrappers around / hooks on sub-modules would be bypassed silently -- refuse."""
        from .layers import assert_plain_tree
        assert_plain_tree(self, "PanguModel")

    def set_constants(self, statistics, maps, const_h):
        """Register default `statistics, maps, const_h` so forward can be called with two arguments."""
        dev = next(self.parameters()).device
        self._c_surface_mean, self._c_surface_std, self._c_upper_mean, self._c_upper_std = (
            torch.as_tensor(t, dtype=torch.float32).to(dev) for t in statistics)
        self._c_maps = torch.as_tensor(maps, dtype=torch.float32).to(dev)
        self._c_const_h = torch.as_tensor(const_h, dtype=torch.float32).to(dev)

    def forward(self, input, input_surface, statistics=None, maps=None, const_h=None, levels_reversed=False):
        """reference pangu_model.py:50-87.  levels_reversed (keyword, not in the reference): `input` is stored with its level axis
        as on disk (ascending) and the reader's `[::-1]` (era5_data/utils_data.py:117) is done by the first kernel's addressing
        instead of a host- or device-side flip of the 270 MB field (data.DevicePrefetcher(fuse_flip=True) delivers such batches)."""
        if statistics is None or maps is None or const_h is None:
            if self._c_maps is None:
                raise TypeError("forward() needs statistics, maps, const_h (or call set_constants() first)")
            statistics = statistics if statistics is not None else (
                self._c_surface_mean, self._c_surface_std, self._c_upper_mean, self._c_upper_std)
            maps = maps if maps is not None else self._c_maps
            const_h = const_h if const_h is not None else self._c_const_h
        if not (input.is_cuda and input_surface.is_cuda):
            raise RuntimeError("PanguModel (MI355X build) needs its inputs on a HIP device; there is no CPU fallback "
                               f"(got {input.device})")
        pdev = self._input_layer.conv.weight.device
        if ops.same_device(input, input_surface) != pdev:
            raise RuntimeError(f"PanguModel: inputs on {input.device}, parameters on {pdev}")
        self._assert_plain_children()
        if input.device.index != torch.cuda.current_device():
            # an 8-GPU node driven from one process (or a caller that never called torch.cuda.set_device): every launch of this
            # forward goes to the INPUT's device and its current stream (ops._stream refuses anything else)
            with torch.cuda.device(input.device):
                return self.forward(input, input_surface, statistics, maps, const_h, levels_reversed)
        # the autograd path runs when a parameter OR one of the two fields asks for a gradient (the reference's patch embedding is
        # plain autograd, layers.py:40-93: `input.requires_grad_()` yields input.grad there); the constant operands are refused
        grad_path = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                 or input.requires_grad or input_surface.requires_grad)
        if torch.is_grad_enabled():
            from .autograd import refuse_constant_grads
            refuse_constant_grads(maps, const_h, statistics)
        want_bf16 = self.compute_dtype == torch.bfloat16 or (
            torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        rev = bool(levels_reversed)
        if grad_path and not self.training and getattr(self, "eval_grad_mode", "save") == "recompute":
            return _EvalRecomputeFn.apply(self, (statistics, maps, const_h, want_bf16, rev), input, input_surface, *self.parameters())
        return self._forward_dispatch(input, input_surface, statistics, maps, const_h, want_bf16, grad_path, rev)

    def _forward_dispatch(self, input, input_surface, statistics, maps, const_h, want_bf16, grad_path, levels_reversed=False):
        if want_bf16:
            from . import autograd_bf16, fused_bf16
            if self._shadow is None:
                self._shadow = fused_bf16.WeightShadow()
            if grad_path:
                return autograd_bf16.forward_train(self, input, input_surface, statistics, maps, const_h, levels_reversed)
            return fused_bf16.forward(self, input, input_surface, statistics, maps, const_h, levels_reversed)
        if self._compact_bias and not grad_path:
            self._build_compact_bias()            # no-op while the tables exist (dropped with the weight shadows)
        return self._forward_f32(input, input_surface, statistics, maps, const_h, grad_path, levels_reversed)

    def _forward_f32(self, input, input_surface, statistics, maps, const_h, grad_path, levels_reversed=False):
        B = input.shape[0]
        x = self._input_layer(input, input_surface, statistics, maps, const_h, levels_reversed)             # (B,521280,192)
        N, C = x.shape[1], x.shape[2]
        # skip connection: layer 0 writes its result into the left half, layer 3 into the right half of one
        # (B,N,2C) buffer, so the channel concat of reference pangu_model.py:81 costs no copy
        if grad_path and B == 1:
            # autograd path, one sample: layer 0 / layer 3 write straight into the two halves of one (N, 2C) buffer, given
            # to autograd as tensors that SHARE its storage without being views of it (a view returned by a custom Function
            # whose base is written again -- the other half -- is refused)
            from . import fused
            cat = torch.empty((N, 2 * C), dtype=x.dtype, device=x.device)
            halves = [torch.empty(0, dtype=x.dtype, device=x.device).set_(cat.untyped_storage(), cat.storage_offset() + off, (N, C), (2 * C, 1))
                      for off in (0, C)]
            skip = self.layers[0](x, 8, 181, 360, out=halves[0])
            skip_grad = [None, False]         # the skip connection's concat-path gradient, summed inside the down-sampling backward
            x = self.downsample(skip, 8, 181, 360, skip_grad=skip_grad)
            x = self.layers[1](x, 8, 91, 180)
            x = self.layers[2](x, 8, 91, 180)
            x = self.upsample(x)
            x = self.layers[3](x, 8, 181, 360, out=halves[1])
            return fused.patch_recover_halves(self._output_layer, skip, x, 8, 181, 360, skip_grad=skip_grad)
        if grad_path:
            skip = self.layers[0](x, 8, 181, 360)                 # autograd path, B > 1: plain concat
            x = self.downsample(skip, 8, 181, 360)
            x = self.layers[1](x, 8, 91, 180)
            x = self.layers[2](x, 8, 91, 180)
            x = self.upsample(x)
            x = self.layers[3](x, 8, 181, 360)
            return self._output_layer(torch.cat((skip, x), dim=-1), 8, 181, 360)
        cat = torch.empty((B, N, 2 * C), dtype=x.dtype, device=x.device)
        skip = self.layers[0](x, 8, 181, 360, out=cat[:, :, :C])
        x = self.downsample(skip, 8, 181, 360)
        x = self.layers[1](x, 8, 91, 180)
        x = self.layers[2](x, 8, 91, 180)
        x = self.upsample(x)
        self.layers[3](x, 8, 181, 360, out=cat[:, :, C:])
        return self._output_layer(cat, 8, 181, 360)

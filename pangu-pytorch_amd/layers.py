"""Layer modules of the Pangu-Weather network, MI355X-native.

Mirror of the reference's module tree (models/layers.py): same class names, constructor arguments,
sub-module names and parameter shapes, so `state_dict()` keys/shapes match the 223-row keys_all.csv and
`named_modules()` introspection (LoRA targets, torch_summarize) keeps working.  The arithmetic is NOT the
reference's op chain: every forward calls hand-written gfx950 kernels through the C ABI (`ops.py`), with
all view/pad/roll/permute/crop steps folded into kernel address arithmetic.  nn.Linear / nn.Conv1d /
nn.LayerNorm sub-modules are parameter containers only.
"""
from collections import OrderedDict

import torch
from torch import nn

from . import fused

WINDOW = (2, 6, 12)


def _trunc_normal_(t, std):
    return nn.init.trunc_normal_(t, std=std)      # stands in for timm's trunc_normal_ (reference layers.py:9)


class DropPath(nn.Module):
    """Stochastic depth (timm semantics, reference layers.py:140): in training each residual branch of a
    sample is dropped with prob p, else scaled 1/(1-p).  Only the per-sample factor is drawn here; it is
    applied inside the fused LayerNorm-residual kernel (and a dropped branch is not computed at all)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.n_dropped = 0           # branches dropped so far (diagnostic)
        self.n_dropped_branch = [0, 0]      # ... split by branch: [attention (layers.py:250), MLP (:251)]; bench.py prices the
                                            # work a training step really executed with these
        self._calls = 0

    def sample_scale(self, training):
        """The per-sample factor of the next residual branch; a block draws twice per sample, attention branch first."""
        which = self._calls & 1
        self._calls += 1
        if not training or self.drop_prob == 0.0:
            return 1.0
        keep = 1.0 - self.drop_prob
        if torch.rand(()).item() < keep:
            return 1.0 / keep
        self.n_dropped += 1
        self.n_dropped_branch[which] += 1
        return 0.0

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:.3f}"


_ALLOWED_MODULE_TYPES = set()      # filled on first use (assert_plain_tree)


def assert_plain_tree(root, what):
    """The kernels read the parameters of the nn.Linear / nn.Conv1d / nn.LayerNorm children directly and never call their
    `forward` (nor, on the bf16 path, the forward of ANY sub-module of the model): a wrapper that replaces such a child (LoRA /
    peft `lora.Linear`, parametrizations, quantisation stubs -- reference finetune/lora_tune.py:124-135 wraps `linear1`) or a
    forward (pre-)hook on a sub-module would be ignored silently and e.g. train nothing.  Refuse instead: every sub-module of
    `root` must be one of this file's classes or the plain torch.nn class, without forward hooks (hooks on `root` itself run)."""
    allowed = _ALLOWED_MODULE_TYPES
    if not allowed:
        allowed.update({nn.Linear, nn.Conv1d, nn.LayerNorm, nn.GELU, nn.Dropout, nn.Identity, nn.Sequential, DropPath,
                        PatchEmbedding_pretrain, Mlp, EarthAttention3D, EarthSpecificBlock, EarthSpecificLayer, DownSample, UpSample,
                        PatchRecovery_pretrain})
    for m in root.modules():
        if m is not root and (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks
                              or type(m) not in allowed):
            break
    else:
        return
    name = next(n for n, q in root.named_modules() if q is m)
    if type(m) in allowed:
        raise RuntimeError(f"{what} (MI355X build): sub-module '{name}' carries forward hooks or backward hooks, but its forward is "
                           "never called (the HIP kernels read the parameters directly): the hook would be ignored")
    raise RuntimeError(f"{what} (MI355X build): sub-module '{name}' is {type(m).__module__}.{type(m).__name__}, not the plain "
                       "module the kernels read their parameters from; a wrapper's own arithmetic (LoRA adapters, "
                       "parametrizations) would be bypassed and e.g. train nothing")


class PatchEmbedding_pretrain(nn.Module):
    """reference layers.py:12-93."""

    def __init__(self, patch_size, dim):
        super().__init__()
        self.conv = nn.Conv1d(in_channels=192, out_channels=dim, kernel_size=1, stride=1)
        self.conv_surface = nn.Conv1d(in_channels=112, out_channels=dim, kernel_size=1, stride=1)
        self.window_size = WINDOW

    def forward(self, input, input_surface, statistics, maps, const_h, levels_reversed=False):
        return fused.patch_embed(self, input, input_surface, statistics, maps, const_h, levels_reversed)


class Mlp(nn.Module):
    """reference layers.py:255-270 (dropout p=0 is the identity and is not materialised)."""

    def __init__(self, dim, dropout_rate):
        super().__init__()
        self.linear1 = nn.Linear(dim, dim * 4)
        self.linear2 = nn.Linear(dim * 4, dim)
        self.activation = nn.GELU()
        self.drop = nn.Dropout(dropout_rate)

    def forward(self, x):
        assert_plain_tree(self, "Mlp")
        if self.drop.p > 0.0 and self.training:
            raise RuntimeError("Mlp (MI355X build): dropout_rate > 0 in train() mode is not implemented by the kernels (the "
                               "reference's model builds every Mlp with rate 0, layers.py:143)")
        shp = x.shape
        return fused.mlp(self, x.reshape(-1, shp[-1])).reshape(shp)


class EarthAttention3D(nn.Module):
    """reference layers.py:272-421.  Inside a block the arithmetic is the fused window-attention kernel (the block never calls
    this module's forward); `forward(x_window, mask)` keeps the module usable on its own."""

    def __init__(self, dim, heads, dropout_rate, window_size, device=None):
        super().__init__()
        self.device = device
        self.linear1 = nn.Linear(dim, dim * 3, bias=True)
        self.linear2 = nn.Linear(dim, dim)
        self.dropout_rate = float(dropout_rate)                # reference layers.py:284 (nn.Dropout; the model passes 0)
        self.head_number = heads
        self.dim = dim
        self.scale = (dim // heads) ** -0.5
        self.window_size = window_size
        input_shape = {192: (8, 186), 384: (8, 96)}[dim]                       # reference layers.py:298-301
        self.type_of_windows = (input_shape[0] // window_size[0]) * (input_shape[1] // window_size[1])
        wtok = window_size[0] * window_size[1] * window_size[2]
        self.earth_specific_bias = nn.Parameter(torch.zeros(1, self.type_of_windows, heads, wtok, wtok, device=device))
        _trunc_normal_(self.earth_specific_bias, std=0.02)
        self._construct_index()

    def forward(self, x, mask):
        """reference layers.py:360-421, the module's own calling convention: x (nLon, types, 144, C) ALREADY partitioned into
        windows (as EarthSpecificBlock.forward :216-221 hands it over), mask None or (nLon, types, 144, 144) (gen_mask, :153-181)
        -> (nLon, types, 144, C).  EarthSpecificBlock does not come through here -- its kernels fold the partition into their
        addressing -- this is for callers that use the module on its own: linear1 and linear2 on the GEMM kernel, the core on
        `pangu_attn_windows_fwd` / `_bwd` (explicit mask tensor, every slot an ordinary token), differentiable like the
        reference's module (autograd.AttentionWindowsFn)."""
        from . import ops
        from .autograd import AttentionWindowsFn
        assert_plain_tree(self, "EarthAttention3D")
        if getattr(self, "dropout_rate", 0.0) > 0.0 and self.training:
            raise RuntimeError("EarthAttention3D (MI355X build): dropout_rate > 0 in train() mode is not implemented by the kernels "
                               "(the reference's model builds every attention with rate 0, layers.py:144)")
        if x.dim() != 4 or x.shape[1] != self.type_of_windows or x.shape[2] != 144 or x.shape[3] != self.dim:
            raise RuntimeError(f"EarthAttention3D: expected (nLon, {self.type_of_windows}, 144, {self.dim}) windows, got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError(f"EarthAttention3D (MI355X build) needs its input on a HIP device (got {x.device}); there is no CPU fallback")
        n_lon = x.shape[0]
        xw = x.to(torch.float32).contiguous().view(-1, self.dim)
        m = None if mask is None else mask.detach().to(device=x.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(x.device):
            if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
                y = AttentionWindowsFn.apply(xw, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                             self.earth_specific_bias, m, (n_lon, self.type_of_windows, self.head_number))
            else:
                qkv = ops.linear(xw, self.linear1.weight, self.linear1.bias)
                o = ops.attention_windows(qkv, self.earth_specific_bias[0], m, n_lon, self.type_of_windows, self.head_number)
                y = ops.linear(o, self.linear2.weight, self.linear2.bias)
        return y.view(x.shape)

    def _construct_index(self):
        """reference layers.py:319-357: `self.position_index`, int64 (20736,) in [0, 3312) -- a plain attribute there too
        (not a buffer, not in state_dict); the reference builds it and never reads it (the gather is commented out,
        :384-391).  Here it is the closed form of weights.position_index (bit-exact against the reference's loops,
        tests/test_weights.py) and what weights.expand_bias / compact_bias index with."""
        from . import weights
        self.position_index = weights.position_index()


class EarthSpecificBlock(nn.Module):
    """reference layers.py:127-253."""

    def __init__(self, dim, drop_path_ratio, heads, device=None):
        super().__init__()
        self.device = device
        self.window_size = WINDOW
        self.drop_path = DropPath(drop_path_ratio) if drop_path_ratio > 0.0 else nn.Identity()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.linear = Mlp(dim, 0)
        self.attention = EarthAttention3D(dim, heads, 0, self.window_size, device=device)
        self.padding_front, self.padding_back = 0, 5
        self.type_of_windows = self.attention.type_of_windows

    def forward(self, x, Z, H, W, roll, out=None):
        assert_plain_tree(self, "EarthSpecificBlock")
        return fused.earth_block(self, x, Z, H, W, roll, out=out)

    def gen_mask(self, x):
        """reference layers.py:153-181: the shifted-window attention mask for a padded activation `x` (1, Z, Hp, W, C) ->
        (W/12, types, 144, 144) fp32 in {0, -100}.  The attention kernels never materialise it (closed form per lane,
        csrc/common.h win_mask); this returns the same kernel-side closed form, exported by `pangu_window_mask_export`
        (bit-exact against the reference's slicing construction, tests/test_gpu_parity.py), expanded over the longitude
        windows as the reference's tensor is (its values do not depend on the longitude window: no W slicing)."""
        from . import ops
        _, Z, Hp, W, _ = x.shape
        m = ops.window_mask(Z, Hp - self.padding_back - self.padding_front, W, x.device)      # (types, 144, 144)
        return m.unsqueeze(0).expand(W // self.window_size[2], -1, -1, -1)


class EarthSpecificLayer(nn.Module):
    """reference layers.py:96-125.  `use_checkpoint` is accepted for signature parity and ignored: with the
    fused attention nothing of size (..,144,144) is kept, so activations fit HBM without recompute."""

    def __init__(self, depth, dim, drop_path_ratio_list, heads, use_checkpoint=False, device=None):
        super().__init__()
        self.device = device
        self.depth = depth
        blocks = OrderedDict()
        for i in range(depth):
            blocks[f"EarthSpecificBlock{i}"] = EarthSpecificBlock(dim, drop_path_ratio_list[i], heads, device=device)
        self.blocks = nn.Sequential(blocks)
        self.use_checkpoint = use_checkpoint

    def forward(self, x, Z, H, W, out=None):
        n = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            x = blk(x, Z, H, W, roll=(i % 2 == 1), out=out if i == n - 1 else None)
        return x


class DownSample(nn.Module):
    """reference layers.py:423-459."""

    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(in_features=4 * dim, out_features=2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, Z, H, W, skip_grad=None):
        return fused.down_sample(self, x, Z, H, W, skip_grad=skip_grad)


class UpSample(nn.Module):
    """reference layers.py:461-499."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.linear1 = nn.Linear(input_dim, output_dim * 4, bias=False)
        self.linear2 = nn.Linear(output_dim, output_dim, bias=False)
        self.norm = nn.LayerNorm(output_dim)

    def forward(self, x, Z=8, H2=91, W2=180, H=181, out=None):
        return fused.up_sample(self, x, Z, H2, W2, H, out=out)


class PatchRecovery_pretrain(nn.Module):
    """reference layers.py:501-545."""

    def __init__(self, dim):
        super().__init__()
        self.patch_size = (2, 4, 4)
        self.dim = dim
        self.conv = nn.Conv1d(in_channels=dim, out_channels=160, kernel_size=1, stride=1)
        self.conv_surface = nn.Conv1d(in_channels=dim, out_channels=64, kernel_size=1, stride=1)

    def forward(self, x, Z, H, W):
        return fused.patch_recover(self, x, Z, H, W)

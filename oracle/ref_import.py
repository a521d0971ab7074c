"""Import the reference (zhaoshan2/pangu-pytorch, read-only at /root/reference) in THIS container only.

TEST INFRASTRUCTURE. The reference needs `timm.models.layers.{DropPath, trunc_normal_}` (models/layers.py:9),
which is not installed here: a two-symbol stand-in module is registered (DropPath restated from its
published definition: per-sample Bernoulli keep mask scaled by 1/keep_prob, identity in eval).
Nothing from /root/reference is copied; it is imported in place and never travels to the GPU box.
"""
import os
import sys
import types
import torch

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


class _DropPath(torch.nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * m / keep


def load():
    """Returns (layers_module, pangu_model_module) of the reference."""
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        tm = types.ModuleType("timm.models")
        tl = types.ModuleType("timm.models.layers")
        tl.DropPath = _DropPath
        tl.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models, tm.layers = tm, tl
        sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import models.layers as L
    import models.pangu_model as M
    return L, M

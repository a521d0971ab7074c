"""pangu-pytorch_amd — MI355X-native Pangu-Weather forward/backward (drop-in for zhaoshan2/pangu-pytorch's
models/pangu_model.py + models/layers.py).  Import as `pangu_pytorch_amd` (see pangu_pytorch_amd.py)."""
from .pangu_model import PanguModel   # noqa: F401
from . import layers, ops, fused, _lib, rollout, train, dist, score, weights, data   # noqa: F401

__all__ = ["PanguModel", "layers", "ops", "fused"]

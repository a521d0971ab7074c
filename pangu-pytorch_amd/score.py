"""Latitude-weighted RMSE / ACC on the device (reference era5_data/score.py:80-135).

The 286 MB prediction/target fields never leave HBM: one HIP reduction pass (`pangu_lat_weighted_sums`) produces the
four weighted sums per (sample, channel) plane, from which both scores follow.  The weights reproduce the reference's
torch versions, including its `3.1416` literal for pi (score.py:89,98 — the numpy versions use np.pi)."""
import torch

from . import _lib
from .ops import _chk, _stream


def latitude_weights(num_lat, device):
    """reference score.py:82-88: num_lat * cos(3.1416/180 * lat_j) / sum_j cos(...), lat_j = 90 - j*180/(num_lat-1)."""
    j = torch.arange(0, num_lat, device=device)
    lat = 90.0 - j * 180.0 / float(num_lat - 1)
    c = torch.cos(3.1416 / 180.0 * lat)
    return (num_lat * c / torch.sum(c)).to(torch.float32).contiguous()


def _sums(pred, target):
    if pred.shape != target.shape or pred.dim() not in (3, 4, 5):
        raise RuntimeError("pred/target must have equal shape (.., H, W)")
    H, W = pred.shape[-2], pred.shape[-1]
    planes = pred.numel() // (H * W)
    out = torch.zeros((planes, 4), dtype=torch.float32, device=pred.device)
    w = latitude_weights(H, pred.device)
    lib = _lib.load()
    _lib.check(lib.pangu_lat_weighted_sums(_stream(pred), _chk(pred.contiguous(), "pred"), _chk(target.contiguous(), "target"),
                                           w.data_ptr(), out.data_ptr(), planes, H, W), "lat_weighted_sums")
    return out.view(pred.shape[:-2] + (4,)), H * W


def weighted_rmse_channels(pred, target):
    """reference weighted_rmse_torch_channels (score.py:92-105): sqrt(mean_{h,w} w_h (p-t)^2) per leading index."""
    s, n = _sums(pred, target)
    return torch.sqrt(s[..., 0] / n)


def weighted_acc_channels(pred, target):
    """reference weighted_acc_torch_channels (score.py:123-135)."""
    s, _ = _sums(pred, target)
    return s[..., 1] / torch.sqrt(s[..., 2] * s[..., 3])


def weighted_rmse(pred, target):
    return weighted_rmse_channels(pred, target).mean(dim=0)


def weighted_acc(pred, target):
    return weighted_acc_channels(pred, target).mean(dim=0)

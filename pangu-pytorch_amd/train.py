"""Training-step pieces around the model (counterpart of reference models/pangu_sample.py:45-77).

The loss is the caller's code in the reference (plain torch ops on the two output fields); it is restated here so
bench/tests/finetune loops need nothing from the reference's era5_data package.
"""
import torch

UPPER_WEIGHTS = (3.00, 0.60, 1.50, 0.77, 0.54)       # reference era5_data/config.py:45
SURFACE_WEIGHTS = (1.50, 0.77, 0.66, 3.00)           # reference era5_data/config.py:46


def norm_data(target, target_surface, stats_last):
    """reference era5_data/utils_data.py:315-321; stats_last = (s_mean(1,4,1,1), s_std, u_mean(1,5,13,1,1), u_std)."""
    s_mean, s_std, u_mean, u_std = stats_last
    return (target - u_mean) / u_std, (target_surface - s_mean) / s_std


_loss_weights = {}      # (device, dtype) -> (w_upper, w_surface) on the device


def _weights_on(device, dtype):
    """The two weight vectors, uploaded once per (device, dtype): torch.tensor(list, device=cuda) is a synchronous
    host-to-device copy, i.e. one that waits for the whole forward in front of it -- 14 ms of host stall per call in the
    bf16 training step, after which the backward launches start from an empty queue."""
    key = (device, dtype)
    w = _loss_weights.get(key)
    if w is None:
        w = (torch.tensor(UPPER_WEIGHTS, dtype=dtype).view(1, 5, 1, 1, 1).to(device),
             torch.tensor(SURFACE_WEIGHTS, dtype=dtype).view(1, 4, 1, 1).to(device))
        _loss_weights[key] = w
    return w


def weighted_l1_loss(output, output_surface, target, target_surface):
    """reference models/pangu_sample.py:61-67: mean(|o-t| * w_upper) + 0.25 * mean(|o_s-t_s| * w_surface)."""
    wu, ws = _weights_on(output.device, output.dtype)
    loss_surface = torch.mean(torch.abs(output_surface - target_surface) * ws)
    loss_upper = torch.mean(torch.abs(output - target) * wu)
    return loss_upper + loss_surface * 0.25


def make_optimizer(model, lr=5e-6, weight_decay=3e-6):
    """The reference's optimiser (finetune_fully.py:121: Adam(lr=5e-6, weight_decay=3e-6)) in torch's single-kernel
    multi-tensor form when the parameters live on a HIP device: the same update rule, one launch instead of ~10 per
    parameter tensor (223 tensors -> 2 200 launches of ~6 us per step otherwise)."""
    params = [p for p in model.parameters() if p.requires_grad]
    fused = all(p.is_cuda for p in params)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=fused)


def train_step(model, optimizer, batch, statistics, maps, const_h, stats_last=None, grad_sync=None):
    """One optimisation step (reference pangu_sample.py:45-77). batch = (input, input_surface, target, target_surface).
    `grad_sync` (optional callable) runs between backward and optimizer.step(): the data-parallel gradient
    all-reduce (the reference's intended `gather_grad`, era5_data/utils_dist.py:125-134)."""
    inp, inp_s, tgt, tgt_s = batch
    optimizer.zero_grad(set_to_none=True)
    out, out_s = model(inp, inp_s, statistics, maps, const_h)
    if stats_last is not None:
        tgt, tgt_s = norm_data(tgt, tgt_s, stats_last)
    loss = weighted_l1_loss(out, out_s, tgt, tgt_s)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    else:
        # a DropPath-dropped branch is not computed here, so its parameters come back without a gradient; the reference
        # computes the branch, multiplies by zero and hands Adam ZERO gradients (moments decay, weight decay applies)
        for group in optimizer.param_groups:
            for p in group["params"]:
                if p.requires_grad and p.grad is None:
                    p.grad = torch.zeros_like(p)
    optimizer.step()
    return loss.detach()

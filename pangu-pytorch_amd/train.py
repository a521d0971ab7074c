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


def _weighted_l1_loss_torch(output, output_surface, target, target_surface, target_levels_reversed=False, stats_last=None):
    if target_levels_reversed:
        target = target.flip(-3)                       # the reader's `[::-1]` (reference era5_data/utils_data.py:117)
    if stats_last is not None:
        target, target_surface = norm_data(target, target_surface, stats_last)
    wu, ws = _weights_on(output.device, output.dtype)
    loss_surface = torch.mean(torch.abs(output_surface - target_surface) * ws)
    loss_upper = torch.mean(torch.abs(output - target) * wu)
    return loss_upper + loss_surface * 0.25


_stats_flat = {}      # (ids of the four statistics tensors) -> (the tensors, their flat fp32 device copies)


def _flat_stats(stats_last, device, Vu, L, Vs):
    """stats_last = (s_mean (1,Vs,1,1), s_std, u_mean (1,Vu,L,1,1), u_std) -> four flat contiguous fp32 device tensors, converted
    once per set of statistics tensors (they are constants of a training run)."""
    key = tuple(id(t) for t in stats_last) + (str(device),)
    hit = _stats_flat.get(key)
    if hit is not None and all(a is b for a, b in zip(hit[0], stats_last)):
        return hit[1]
    s_mean, s_std, u_mean, u_std = stats_last
    if u_mean.numel() != Vu * L or u_std.numel() != Vu * L or s_mean.numel() != Vs or s_std.numel() != Vs:
        raise RuntimeError(f"weighted_l1_loss: stats_last shapes {[tuple(t.shape) for t in stats_last]} for {Vu} x {L} upper / {Vs} surface planes")
    flat = tuple(t.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous() for t in (u_mean, u_std, s_mean, s_std))
    if len(_stats_flat) > 8:
        _stats_flat.clear()
    _stats_flat[key] = (tuple(stats_last), flat)
    return flat


class WeightedL1LossFn(torch.autograd.Function):
    """The loss and its gradient as one HIP pass each (csrc/loss.hip): the torch expression reads / writes the 286 MB fields
    fourteen times per step (0.9 ms); the forward here reads them once, the backward once more and writes the two gradients.
    The target side of the reference's loop body is folded into the same two passes: `normData` of the targets
    (era5_data/utils_data.py:315-321, called at models/pangu_sample.py:57) when `stats_last` is given, and the reader's level
    reversal (utils_data.py:117) when the target arrives with ascending levels (`target_levels_reversed`)."""

    @staticmethod
    def forward(ctx, output, output_surface, target, target_surface, target_levels_reversed=False, stats_last=None):
        from . import _lib
        lib = _lib.load()
        wu, ws = _weights_on(output.device, torch.float32)
        with torch.cuda.device(output.device):
            return WeightedL1LossFn._forward(ctx, lib, wu, ws, output, output_surface, target, target_surface,
                                             bool(target_levels_reversed), stats_last)

    @staticmethod
    def _forward(ctx, lib, wu, ws, output, output_surface, target, target_surface, rev, stats_last):
        from . import _lib
        from .ops import _stream
        B, Vu, L = output.shape[0], output.shape[1], output.shape[2]
        Vs = output_surface.shape[1]
        geom = (B, Vu, output[0, 0].numel(), Vs, output_surface[0, 0].numel(), L)
        nblk = lib.pangu_weighted_l1_loss_blocks(*geom)
        if nblk <= 0:
            raise RuntimeError(f"weighted_l1_loss: unsupported shapes {tuple(output.shape)} {tuple(output_surface.shape)}")
        st = _flat_stats(stats_last, output.device, Vu, L, Vs) if stats_last is not None else ()
        sp = tuple(t.data_ptr() for t in st) if st else (None,) * 4
        partial = torch.empty((nblk,), dtype=torch.float32, device=output.device)
        loss = torch.empty((3,), dtype=torch.float32, device=output.device)
        _lib.check(lib.pangu_weighted_l1_loss_fwd(_stream(output), output.data_ptr(), target.data_ptr(), output_surface.data_ptr(),
                                                  target_surface.data_ptr(), wu.data_ptr(), ws.data_ptr(), partial.data_ptr(),
                                                  loss.data_ptr(), *geom, int(rev), *sp), "weighted_l1_loss_fwd")
        ctx.save_for_backward(output, output_surface, target, target_surface, *st)
        ctx.geom, ctx.rev = geom, rev
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        from .ops import _stream
        output, output_surface, target, target_surface, *st = ctx.saved_tensors
        wu, ws = _weights_on(output.device, torch.float32)
        d_o, d_os = torch.empty_like(output), torch.empty_like(output_surface)
        g = g.to(torch.float32).contiguous()
        sp = tuple(t.data_ptr() for t in st) if st else (None,) * 4
        with torch.cuda.device(output.device):
            _lib.check(_lib.load().pangu_weighted_l1_loss_bwd(
                _stream(output), output.data_ptr(), target.data_ptr(), output_surface.data_ptr(), target_surface.data_ptr(),
                wu.data_ptr(), ws.data_ptr(), g.data_ptr(), d_o.data_ptr(), d_os.data_ptr(), *ctx.geom, int(ctx.rev), *sp),
                "weighted_l1_loss_bwd")
        return d_o, d_os, None, None, None, None


def _hip_loss_ok(output, output_surface, target, target_surface):
    ts = (output, output_surface, target, target_surface)
    return (all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == output.device for t in ts)
            and output.dim() == 5 and output_surface.dim() == 4 and output.shape == target.shape
            and output_surface.shape == target_surface.shape and output.shape[0] == output_surface.shape[0]
            and output.shape[1] == len(UPPER_WEIGHTS) and output_surface.shape[1] == len(SURFACE_WEIGHTS)
            and not target.requires_grad and not target_surface.requires_grad)


def weighted_l1_loss(output, output_surface, target, target_surface, target_levels_reversed=False, stats_last=None):
    """reference models/pangu_sample.py:61-67: mean(|o-t| * w_upper) + 0.25 * mean(|o_s-t_s| * w_surface).  Contiguous fp32
    fields on a HIP device take the two-pass HIP form (WeightedL1LossFn); anything else (other dtypes, CPU tensors of the
    host-side tests, targets that need gradients) is the reference's own torch expression.
    stats_last: the targets are in physical units and are normalised first (`normData`, :57); target_levels_reversed: the upper-air
    target is stored with ascending levels (as on disk) -- both are folded into the HIP passes."""
    if _hip_loss_ok(output, output_surface, target, target_surface):
        return WeightedL1LossFn.apply(output, output_surface, target, target_surface, target_levels_reversed, stats_last)
    return _weighted_l1_loss_torch(output, output_surface, target, target_surface, target_levels_reversed, stats_last)


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam's update (L2 weight decay, no amsgrad / maximize) for fp32 parameters on a HIP device, ONE launch per
    parameter group over a device-resident job table (csrc/adam.hip: the arithmetic of torch's fused Adam, bit-identical; 14
    launches of <= 320 workgroups there, 1.8 ms for the model's 7.7 GB).  State layout and hyper-parameter names are
    torch.optim.Adam's (`step`, `exp_avg`, `exp_avg_sq`; lr / betas / eps / weight_decay per group), so schedulers work unchanged.
    Optimizer steps advance ops' weights epoch through the global post-step hook like any torch optimizer."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, shadow_of=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}        # group index -> (signature, device table, n_jobs, total blocks)
        # shadow_of: a PanguModel whose bf16 weight shadows (fused_bf16.WeightShadow) this optimizer keeps current for the plain
        # casts -- the Earth-specific bias tables, 94 % of the bytes: the kernel writes the bf16 image next to the updated fp32
        # value instead of the refresh launch re-reading 1.04 GB
        self._shadow_of = shadow_of

    # the device job tables hold raw pointers into the optimizer state: anything that can replace state tensors drops them
    def load_state_dict(self, state_dict):
        self._tables = {}
        return super().load_state_dict(state_dict)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    def add_param_group(self, param_group):
        self._tables = {}
        return super().add_param_group(param_group)

    @torch.no_grad()
    def step(self, closure=None, missing_as_zero=False):
        """missing_as_zero: parameters without a gradient are stepped with a ZERO gradient (moments decay, weight decay applies)
        instead of being skipped -- what the reference's DropPath-dropped branches get (the branch is computed and multiplied by
        zero there) -- without materialising the zeros."""
        import math
        import struct

        from . import _lib
        from .ops import _stream
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            rows, first, dev = [], 0, None
            beta1, beta2 = group["betas"]
            ws = getattr(self._shadow_of, "_shadow", None) if self._shadow_of is not None else None
            imaged = []
            for p in group["params"]:
                if p.grad is None and not (missing_as_zero and p.requires_grad):
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and
                        (g is None or (g.dtype == torch.float32 and g.device == p.device))):
                    raise RuntimeError("HipAdam: parameters and gradients must be contiguous float32 tensors on one HIP device")
                if g is not None and not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if torch.is_tensor(st["step"]):          # a state_dict written by torch.optim.Adam keeps `step` as a tensor
                    st["step"] = int(st["step"].item())
                st["step"] += 1
                n = p.numel()
                if n == 0:
                    continue
                img = ws.plain_image(p) if ws is not None else None
                if img is not None and (img.numel() != n or img.device != p.device):
                    img = None
                if img is not None:
                    imaged.append(p)
                rows.append((p.data_ptr(), g.data_ptr() if g is not None else 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             img.data_ptr() if img is not None else 0, n, st["step"], first))
                first += (n + 4095) // 4096
                if dev is not None and p.device != dev:
                    raise RuntimeError("HipAdam: the parameters of one group must live on one device")
                dev = p.device
            if not rows:
                continue
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("HipAdam implements torch.optim.Adam without amsgrad / maximize (a loaded state_dict asked for them)")
            # ATen's fused Adam (ATen/native/cuda/fused_adam_utils.cuh:130-137, torch 2.10): both corrections AND the square root
            # are formed in double (`std::pair<double, double>`), then narrowed to the float arguments of adam_math -- checked bit
            # for bit over 40 steps by tests/test_gpu_hardening.py (rounding to float before the root, as ADVICE r3 suggested, breaks it)
            bias = lambda k: (1.0 - beta1 ** k, math.sqrt(1.0 - beta2 ** k))
            uniform = all(r[6] == rows[0][6] for r in rows)
            # the table holds pointers (and per-tensor bias corrections only when the step counts differ): with gradients that keep
            # their addresses -- FlatGradSync's flat buffer, or the caching allocator handing the same blocks back every step -- it
            # is uploaded once; otherwise through pinned memory, asynchronously (no host sync in the training loop)
            # (the moments' addresses are part of the signature: `load_state_dict` / a rollback that replaces `exp_avg` while the
            # parameters and the flat gradient buffer stay put must not leave the kernel updating the old, freed tensors)
            sig = tuple((r[0], r[1], r[2], r[3], r[4], r[5], 0 if uniform else r[6]) for r in rows)
            hit = self._tables.get(gi)
            if hit is None or hit[0] != sig:
                tab = []
                for (pp, gp, mp, vp, sp, n, k, fb) in rows:
                    bits = 0
                    if not uniform:
                        b1, b2 = bias(k)
                        bits = struct.unpack("<I", struct.pack("<f", b1))[0] | (struct.unpack("<I", struct.pack("<f", b2))[0] << 32)
                        if bits >= 1 << 63:
                            bits -= 1 << 64
                    tab.append([pp, gp, mp, vp, sp, n, bits, fb])
                tab.append([0, 0, 0, 0, 0, 0, 0, first])
                hit = self._tables[gi] = (sig, torch.tensor(tab, dtype=torch.int64).pin_memory().to(dev, non_blocking=True), len(rows), first)
            b1, b2 = bias(rows[0][6])
            with torch.cuda.device(dev):      # the launch goes to the parameters' device whatever the caller's current device is
                _lib.check(lib.pangu_adam_step_multi(_stream(hit[1]), hit[1].data_ptr(), hit[2], hit[3], float(group["lr"]), float(beta1),
                                                     float(beta2), float(group["weight_decay"]), float(group["eps"]), b1, b2), "adam_step_multi")
            for p in imaged:
                ws.mark_fresh(p)
        return loss


def make_optimizer(model, lr=5e-6, weight_decay=3e-6):
    """The reference's optimiser (finetune_fully.py:121: Adam(lr=5e-6, weight_decay=3e-6)).  Parameters on a HIP device: the
    one-launch HipAdam above (bit-identical to torch's fused Adam); CPU parameters (host-side tests): torch.optim.Adam."""
    params = [p for p in model.parameters() if p.requires_grad]
    on_gpu = all(p.is_cuda for p in params)
    if on_gpu and all(p.dtype == torch.float32 and p.is_contiguous() for p in params):
        return HipAdam(params, lr=lr, weight_decay=weight_decay, shadow_of=model)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=on_gpu)


class GraphedTrainStep:
    """OPT-IN: forward + loss + backward of one fixed-shape batch captured in a hipGraph, Adam stepped eagerly after each replay
    (its bias corrections are host scalars that change every step; it is ONE launch).  For a model WITHOUT active stochastic depth
    only (model.eval(), or every DropPath at rate 0): a captured step cannot skip a dropped branch -- it would have to compute it
    and scale it by a zero read from device memory, as the reference does, which costs the ~10 % of block work that not computing
    dropped branches saves.  Measured (tools/profile_train.py <dtype> <steps> <warmup> graph): the eager step is
    not host-bound, so replaying it from a graph recovers next to nothing -- this class exists to make that a number.
    The bf16 weight shadows the graph reads are re-made IN PLACE after every optimizer step (WeightShadow.refresh_in_place)."""

    def __init__(self, model, optimizer, batch, statistics, maps, const_h, stats_last=None, warmup=2):
        from .layers import DropPath
        if model.training and any(isinstance(m, DropPath) and m.drop_prob > 0.0 for m in model.modules()):
            raise RuntimeError("GraphedTrainStep: stochastic depth is active (model.train() with DropPath rates > 0): a captured step "
                               "cannot skip dropped branches; call model.eval() or use train.train_step")
        self.model, self.optimizer = model, optimizer
        self.batch = [t.clone() for t in batch]
        self.consts, self.stats_last = (statistics, maps, const_h), stats_last
        dev = self.batch[0].device
        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):            # outside capture: weight shadows, loss weights, allocator, lazy LDS attributes
                    optimizer.zero_grad(set_to_none=True)
                    self._fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            optimizer.zero_grad(set_to_none=True)      # the gradients are (re-)allocated INSIDE the capture: static addresses
            from . import ops
            ops.reset_pass_arenas()                    # never a zero buffer filled outside the capture (e.g. left by a failed backward)
            ops.release_stream_workspaces()            # the capture allocates its own weight-gradient scratch, inside its own pool
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = self._fwd_bwd()
            ops.release_stream_workspaces()            # ... which stays this graph's: nothing else may be handed a pointer into the pool
        # the graph writes THESE gradient tensors; `optimizer.zero_grad(set_to_none=True)` between replays (the usual loop idiom)
        # detaches them from the parameters, and HipAdam would then skip every parameter silently: step() re-attaches them
        self._grads = [(p, p.grad) for group in optimizer.param_groups for p in group["params"] if p.grad is not None]

    def _fwd_bwd(self):
        inp, inp_s, tgt, tgt_s = self.batch
        # every forward here IS followed by its backward: the activation-saving forward at once, not inference kernels + a recompute
        # (PanguModel.eval_grad_mode; this class needs stochastic depth off, i.e. usually model.eval())
        mode, self.model.eval_grad_mode = getattr(self.model, "eval_grad_mode", "save"), "save"
        try:
            out, out_s = self.model(inp, inp_s, *self.consts)
        finally:
            self.model.eval_grad_mode = mode
        loss = weighted_l1_loss(out, out_s, tgt, tgt_s, stats_last=self.stats_last)      # normData folded into the loss passes
        loss.backward()
        return loss.detach()

    def step(self, batch=None):
        """Replay fwd + loss + bwd on `batch` (copied into the static buffers; None = the captured batch again), then one optimizer
        step; returns the (static) loss tensor."""
        if batch is not None:
            for dst, src in zip(self.batch, batch):
                dst.copy_(src)
        self.graph.replay()
        for p, g in self._grads:            # (see __init__: a zero_grad(set_to_none=True) in the caller's loop detached them)
            if p.grad is not g:
                p.grad = g
        self.optimizer.step()
        sh = getattr(self.model, "_shadow", None)
        if sh is not None and sh.cache:
            sh.refresh_in_place()
        return self.loss


def _owns_dropped_branches(grad_sync):
    """True when `grad_sync` is a bound method of dist.FlatGradSync (its flat buffer holds a zero for every parameter whose
    branch was dropped, and every rank launches every bucket whatever it drew)."""
    from .dist import FlatGradSync
    return isinstance(getattr(grad_sync, "__self__", None), FlatGradSync)


def train_step(model, optimizer, batch, statistics, maps, const_h, stats_last=None, grad_sync=None, levels_reversed=False):
    """One optimisation step (reference pangu_sample.py:45-77). batch = (input, input_surface, target, target_surface).
    `grad_sync` (optional callable) runs between backward and optimizer.step(): the data-parallel gradient
    all-reduce (the reference's intended `gather_grad`, era5_data/utils_dist.py:125-134).
    `stats_last`: the targets are in physical units (`normData`, :57, folded into the loss kernel).
    `levels_reversed`: input and target carry their level axis as on disk (ascending); the reader's reversal
    (era5_data/utils_data.py:117) is done by the first / last kernel's addressing (data.DevicePrefetcher(fuse_flip=True)).

    Dropped-branch contract: a DropPath-dropped branch is not computed here, so the backward leaves its parameters WITHOUT a
    gradient and the arm that steps the optimizer hands them the reference's zero-gradient step (moments decay, weight decay
    applies).  That holds for grad_sync=None and for a dist.FlatGradSync method (flat buffer: zeros are already there, every rank
    launches every bucket).  Any OTHER grad_sync callable (e.g. dist.gather_grad, which all-reduces parameter by parameter and
    skips `p.grad is None`) sees materialised ZERO gradients instead: ranks that drew different DropPath patterns would
    otherwise issue different numbers of collectives (a hang) and the optimizer would skip those parameters."""
    inp, inp_s, tgt, tgt_s = batch
    optimizer.zero_grad(set_to_none=True)
    out, out_s = model(inp, inp_s, statistics, maps, const_h, levels_reversed=levels_reversed)
    loss = weighted_l1_loss(out, out_s, tgt, tgt_s, target_levels_reversed=levels_reversed, stats_last=stats_last)
    from . import ops
    lean = grad_sync is None or _owns_dropped_branches(grad_sync)
    with ops.dropped_branch_grads("none" if lean else "zeros"):
        loss.backward()
    if grad_sync is not None:
        grad_sync()
        optimizer.step()
    elif isinstance(optimizer, HipAdam):
        # a DropPath-dropped branch is not computed here, so its parameters come back without a gradient; the reference
        # computes the branch, multiplies by zero and hands Adam ZERO gradients (moments decay, weight decay applies)
        optimizer.step(missing_as_zero=True)
    else:
        for group in optimizer.param_groups:
            for p in group["params"]:
                if p.requires_grad and p.grad is None:
                    p.grad = torch.zeros_like(p)
        optimizer.step()
    return loss.detach()

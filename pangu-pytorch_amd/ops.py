"""Thin Python wrappers over the C ABI: device pointers in, torch-allocated outputs out.

Every function requires CUDA(HIP) fp32 contiguous tensors and launches on torch's current stream.
"""
import threading

import torch

from . import _lib

ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_ADD = 0, 1, 2, 3

# Optional live kernel timing (bench.py): HIP events recorded on the launch stream around each launch.
_timing = None


def timing_start():
    global _timing
    _timing = []


def timing_stop(kind, also=()):
    """-> (total ms, total algorithmic FLOP (or bytes), launches) of the launches tagged `kind`; stops timing.
    With `also` (more tags) a dict {tag: (ms, work, launches)} for those tags is appended to the tuple."""
    global _timing
    rec, _timing = _timing or [], None
    torch.cuda.synchronize()

    def agg(tag):
        sel = [(a.elapsed_time(b), w) for k, a, b, w in rec if k == tag]
        return sum(t for t, _ in sel), sum(w for _, w in sel), len(sel)

    if also:
        return agg(kind) + ({t: agg(t) for t in also},)
    return agg(kind)


class _timed:
    def __init__(self, kind, work):
        self.kind, self.work = kind, work

    def __enter__(self):
        if _timing is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if _timing is not None:
            self.b.record(torch.cuda.current_stream())
            _timing.append((self.kind, self.a, self.b, self.work))



def _stream(t):
    """The stream a launch on `t`'s device goes to: torch's current stream of the CURRENT device -- which must be `t`'s.
    A tensor on cuda:1 launched while the current device is cuda:0 would put device-1 pointers on a device-0 stream (silent
    peer access over xGMI, or a fault): refused here, once per launch.  PanguModel.forward enters `torch.cuda.device(input.device)`
    itself and autograd runs a backward under the device of its forward, so only hand-made calls of single ops can trip this."""
    dev = t.device if isinstance(t, torch.Tensor) else torch.device(t)
    if dev.type != "cuda":
        raise RuntimeError(f"the Pangu HIP path needs tensors on an MI355X device (got {dev}); there is no CPU fallback")
    cur = torch.cuda.current_device()
    if dev.index is not None and dev.index != cur:
        raise RuntimeError(f"Pangu HIP op called with tensors on {dev} while the current device is cuda:{cur}: wrap the call in "
                           f"`with torch.cuda.device({dev.index}):` (PanguModel.forward does this for its own launches)")
    return torch.cuda.current_stream().cuda_stream


def same_device(*tensors):
    """All tensors on one HIP device (None entries skipped), else RuntimeError -- the model-level check behind `_stream`'s
    first-tensor guard."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"Pangu HIP path: tensors on different devices ({dev} and {t.device})")
    return dev


def _chk(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the Pangu HIP path needs tensors on an MI355X device (got {t.device}); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


def _rows(t, name):
    """2-D tensor with unit inner stride (row stride may exceed the width)."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{name}: expected a CUDA float32 2-D tensor with unit inner stride")
    return t.data_ptr(), t.stride(0)


# What a block's backward returns for the parameters of a DropPath-dropped branch.  The branch is not computed here; in the
# reference it is computed and multiplied by zero (layers.py:250-251), so autograd hands the optimizer ZERO gradients there and
# e.g. `torch.optim.Adam` still steps those parameters (weight decay, moment decay, `step` advanced).
#   "zeros" (DEFAULT since round 5: the reference's semantics under ANY optimizer / training loop -- the drop-in contract of
#           models/pangu_sample.py:45-77 is "unchanged"): explicit zero tensors (the 62-MB bias table's zeros are written
#           straight into its flat data-parallel slot when there is one),
#   "none"  what train.train_step selects around its own backward: it gives those parameters the zero-gradient step itself
#           (HipAdam.step(missing_as_zero=True), dist.FlatGradSync) without materialising zeros.
_dropped_default = "zeros"
_dropped_scopes = {"none": 0, "zeros": 0}      # active `dropped_branch_grads` scopes per mode (any thread)
_dropped_lock = threading.Lock()


def set_dropped_branch_grads(mode):
    """The process-wide default policy (outside any `dropped_branch_grads` scope)."""
    global _dropped_default
    if mode not in ("none", "zeros"):
        raise ValueError("set_dropped_branch_grads: 'none' or 'zeros'")
    _dropped_default = mode


def dropped_branch_policy():
    """The policy in force: "zeros" while ANY "zeros" scope is active (always safe: it only materialises what the reference
    materialises), else "none" while any "none" scope is active, else the default.  Scopes are COUNTED, not saved-and-restored:
    two loops training two models from two threads may enter and leave in any order without leaving the process at "none"."""
    if _dropped_scopes["zeros"]:
        return "zeros"
    if _dropped_scopes["none"]:
        return "none"
    return _dropped_default


class dropped_branch_grads:
    """`with ops.dropped_branch_grads("none"): loss.backward()` -- the policy for the backward passes run inside the context
    (process-wide, like the setter; backward runs on autograd's threads, so a thread-local would not reach it)."""

    def __init__(self, mode):
        if mode not in ("none", "zeros"):
            raise ValueError("dropped_branch_grads: 'none' or 'zeros'")
        self.mode = mode

    def __enter__(self):
        with _dropped_lock:
            _dropped_scopes[self.mode] += 1
        return self

    def __exit__(self, *exc):
        with _dropped_lock:
            _dropped_scopes[self.mode] -= 1


def fill_dropped_grads(g, like):
    """g: name -> gradient or None; like: name -> parameter of that shape.  Under the "zeros" policy every None becomes zeros."""
    if dropped_branch_policy() == "zeros":
        for k, p in like.items():
            if g.get(k) is None:
                # matrices (`like` holds the parameter itself for every >= 2-D entry; vectors are shape stand-ins): with a
                # dist.FlatGradSync registered, zero the parameter's flat-buffer slot in place (no zeros tensor, no copy pass)
                slot = grad_slot(p) if p.dim() >= 2 else None
                if slot is not None:
                    g[k] = slot.zero_().view(p.shape)
                else:
                    g[k] = torch.zeros_like(p)
    return g


# Derived copies of parameters (bf16 weight shadows, packed weight images, compact bias tables) are keyed by a STAMP of the
# parameter they were made from.  `_version` alone is not enough: torch's fused optimizers (Adam(fused=True), the form
# train.make_optimizer uses) update the parameters through one multi-tensor kernel that does NOT bump `_version` (measured on
# torch 2.10: 0 before and after three steps), so a shadow keyed by it would stay at the first step's weights for the whole run.
# Every torch optimizer step therefore advances a process-wide epoch that is part of every stamp (conservative: any optimizer
# stepping anything re-makes every derived copy once).
_weights_epoch = [0]


def bump_weights_epoch(*_):
    """Invalidate every derived copy of every parameter (registered below as a global optimizer post-step hook; call it
    yourself after updating weights by other means than a torch optimizer or an op that bumps `_version`)."""
    _weights_epoch[0] += 1


def param_stamp(p):
    """What identifies the CONTENT of a parameter as far as it can be observed cheaply: optimizer steps advance the epoch,
    in-place updates through the parameter bump `_version` (`copy_` under no_grad), `param.data = w` (the reference's own import
    idiom, models/onnx2torch.py:37-52) swaps the storage, i.e. `data_ptr()`.  In-place edits made THROUGH `param.data`
    (`p.data.copy_(..)`) change none of them: call `model.invalidate_shadows()` or `ops.bump_weights_epoch()` after such an edit."""
    return (_weights_epoch[0], p._version, p.data_ptr(), tuple(p.shape), p.device)


def _install_epoch_hook():
    import torch.optim.optimizer as _o
    if getattr(_o, "_pangu_epoch_hook", None) is None and hasattr(_o, "register_optimizer_step_post_hook"):
        _o._pangu_epoch_hook = _o.register_optimizer_step_post_hook(bump_weights_epoch)
    return getattr(_o, "_pangu_epoch_hook", None) is not None


_epoch_hook_installed = _install_epoch_hook()


def require_epoch_hook():
    """Called where derived copies of parameters are first made: without the global optimizer hook (torch < 2.0) a fused
    optimizer step would leave them stale silently -- refuse instead."""
    if not _epoch_hook_installed:
        raise RuntimeError("torch.optim.optimizer.register_optimizer_step_post_hook is missing (torch >= 2.0 required): the bf16 "
                           "weight shadows could not be invalidated after optimizer steps")

# Gradient slots (dist.FlatGradSync): parameter storage address -> the fp32 view of the flat gradient buffer that will hold
# that parameter's gradient.  Backward kernels that WRITE (not accumulate) a parameter gradient -- the Earth-specific bias
# tables, 94 % of the 1.107 GB -- store straight into the slot, so the flat buffer is filled without a copy pass.
_grad_slots = {}          # parameter storage address -> (parameter, flat view, owner)
_grad_claimed = set()     # addresses whose slot was handed to a kernel since the parameter's gradient was last accumulated


def register_grad_slots(slots, owner=None):
    """slots: {parameter: flat-buffer view}, MERGED into the registry under `owner` (a dist.FlatGradSync); a parameter
    registered before moves to the new owner."""
    for p, v in slots.items():
        _grad_slots[p.data_ptr()] = (p, v, owner)
        _grad_claimed.discard(p.data_ptr())


def unregister_grad_slots(owner):
    """Drop the entries registered under `owner` (only those: another FlatGradSync's slots stay)."""
    for k in [k for k, hit in _grad_slots.items() if hit[2] is owner]:
        del _grad_slots[k]
        _grad_claimed.discard(k)


def release_grad_slot(param):
    """The parameter's gradient has been accumulated (or the step is over): its slot may be handed out again."""
    _grad_claimed.discard(param.data_ptr())


def grad_slot(param):
    """The flat-buffer view for this parameter's gradient, or None.  None when: no registry entry; the parameter already
    holds a gradient (accumulation over several backward passes: autograd must ADD, so a kernel may not overwrite the slot);
    or the slot was ALREADY handed out in this backward pass -- one parameter can feed several autograd nodes (per-GPU batch
    B > 1 calls the block function once per sample; a model applied twice in one graph) and autograd's input buffer still
    holds the first node's result as an alias of the slot until every edge has arrived: a second kernel writing the same
    memory would turn g1 + g2 into 2*g2.  Later nodes get None, write a fresh tensor, and autograd adds it to the slot."""
    k = param.data_ptr()
    hit = _grad_slots.get(k)
    if hit is None or k in _grad_claimed:
        return None
    p, v, _ = hit
    if p.grad is not None or v.numel() != param.numel() or v.device != param.device:
        return None
    _grad_claimed.add(k)
    return v



_PASS_ARENA_NUMEL = 28 << 20      # fp32 elements: every atomically accumulated gradient buffer of one backward pass of the
                                  # model (16 blocks x (12 C^2 + 16 C) + the resampling / embedding / recovery weights = 24 M) fits
# One arena per (device, autograd thread): autograd runs one worker thread per device, so two models on two GPUs driven from one
# process run their backward passes concurrently and must not see each other's arena (the single global then thrashed,
# a fresh 112-MB fill per buffer).  Every arena carries the id of the backward pass (autograd graph task) it was made in: an
# arena left armed by a backward that RAISED (its release callback never ran) is never handed to a later pass -- in particular
# not to a GraphedTrainStep capture, whose graph would otherwise accumulate onto a buffer zeroed outside the capture.
_pass_arenas = {}                 # (device index, thread id) -> [flat fp32 zeros, used elements, graph task id]


def _release_pass_arena(key):
    _pass_arenas.pop(key, None)


def reset_pass_arenas():
    """Drop every arena (train.GraphedTrainStep calls this before capturing; harmless at any time outside a backward pass)."""
    _pass_arenas.clear()


def _graph_task_id():
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def _zeros(shape, device):
    """fp32 zeros of `shape` for a gradient the kernels ACCUMULATE into (weight / bias / LayerNorm / pad-slot gradients).  Inside
    an autograd backward pass they are 16-B aligned slices of ONE buffer zeroed by ONE fill at its first use (112 MB: ~25 us)
    and dropped by an end-of-backward callback -- the per-buffer fills were 27 launches per training step (round 3: one per
    block).  Outside a backward pass (direct op calls, tests) a plain torch.zeros."""
    import functools
    import threading
    n = 1
    for d in shape:
        n *= int(d)
    dev = torch.device(device)
    if dev.index is None and dev.type == "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    task = _graph_task_id()
    if task < 0 or n > _PASS_ARENA_NUMEL // 4:      # not inside a backward pass / too large for the arena
        return torch.zeros(shape, dtype=torch.float32, device=dev)
    key = (dev.index, threading.get_ident())
    a = _pass_arenas.get(key)
    if a is None or a[2] != task or a[0].device != dev or a[1] + n > a[0].numel():
        try:
            torch.autograd.Variable._execution_engine.queue_callback(functools.partial(_release_pass_arena, key))
        except RuntimeError:      # not inside a backward pass
            return torch.zeros(shape, dtype=torch.float32, device=dev)
        a = _pass_arenas[key] = [torch.zeros((_PASS_ARENA_NUMEL,), dtype=torch.float32, device=dev), 0, task]
    t = a[0][a[1]:a[1] + n].view(shape)
    a[1] += (n + 3) & ~3
    return t


_U32_BYTES = (1 << 32) - (1 << 24)


def _row_chunks(M, *row_bytes):
    """The GEMM-family kernels address operands with 32-bit byte offsets (PANGU_E_RANGE): row ranges [(m0, m1), ..] that
    keep every operand below 4 GB, or None when the call fits as it is (every shape of the B <= 2 model does)."""
    lim = _U32_BYTES // max(row_bytes) - 256          # the C entries keep 256 rows of slack for their tile tails
    if M <= lim:
        return None
    if lim < 1:
        raise RuntimeError(f"row stride of {max(row_bytes)} bytes is too large for the 32-bit addressed GEMM kernels")
    if lim >= 64:
        lim = lim // 64 * 64
    return [(m, min(M, m + lim)) for m in range(0, M, lim)]


def window_index(Z, H, W, shifted, device):
    lib = _lib.load()
    Hp = H + 5
    out = torch.empty((W // 12, (Z // 2) * (Hp // 6), 144), dtype=torch.int32, device=device)
    _lib.check(lib.pangu_window_index_export(_stream(device), out.data_ptr(), Z, H, W, int(shifted)), "window_index_export")
    return out


def window_mask(Z, H, W, device):
    lib = _lib.load()
    Hp = H + 5
    out = torch.empty(((Z // 2) * (Hp // 6), 144, 144), dtype=torch.float32, device=device)
    _lib.check(lib.pangu_window_mask_export(_stream(device), out.data_ptr(), Z, H, W), "window_mask_export")
    return out


def linear(a, weight, bias=None, act=ACT_NONE, out=None, aux=None):
    """out[M,N] = act(a[M,K] @ weight[N,K]^T + bias). `a`/`out` may be row-strided views.
    act=GELU: aux (optional, dense [M,N]) receives the pre-activation; act=GELU_BWD: out = (a@w^T) * gelu'(aux);
    act=ADD: out = a@w^T + bias + aux (residual-gradient accumulation fused into the data-gradient GEMM)."""
    lib = _lib.load()
    ap, lda = _rows(a, "linear.a")
    M, K = a.shape
    w2 = weight.reshape(weight.shape[0], -1)          # Conv1d(k=1) weights are (out,in,1)
    N = w2.shape[0]
    if w2.shape[1] != K:
        raise RuntimeError(f"linear: weight {tuple(weight.shape)} does not match input width {K}")
    wp = _chk(w2, "linear.weight")
    bp = _chk(bias, "linear.bias") if bias is not None else None
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    op, ldc = _rows(out, "linear.out")
    chunks = _row_chunks(M, 4 * lda, 4 * ldc)
    if chunks is not None:
        for m0, m1 in chunks:
            linear(a[m0:m1], weight, bias, act, out[m0:m1], aux[m0:m1] if aux is not None else None)
        return out
    with _timed("linear", 2.0 * M * N * K):
        _lib.check(lib.pangu_linear_fwd(_stream(a), ap, lda, wp, bp, op, ldc, M, N, K, act,
                      _chk(aux, "linear.aux") if aux is not None else None), "linear_fwd")
    return out


def linear_ln_residual(a, weight, bias, shortcut, gamma, beta, out=None, branch_scale=1.0):
    """out = shortcut + branch_scale * (LayerNorm(a @ weight^T + bias) * gamma + beta) in ONE launch (N = 192; inference)."""
    lib = _lib.load()
    ap, lda = _rows(a, "linear_ln.a")
    M, K = a.shape
    N = weight.shape[0]
    if weight.shape[1] != K or tuple(shortcut.shape) != (M, N):
        raise RuntimeError(f"linear_ln_residual: a {tuple(a.shape)} weight {tuple(weight.shape)} shortcut {tuple(shortcut.shape)}")
    sp, lds = _rows(shortcut, "linear_ln.shortcut")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    op, ldo = _rows(out, "linear_ln.out")
    chunks = _row_chunks(M, 4 * lda, 4 * ldo, 4 * lds)
    if chunks is not None:
        for m0, m1 in chunks:
            linear_ln_residual(a[m0:m1], weight, bias, shortcut[m0:m1], gamma, beta, out[m0:m1], branch_scale)
        return out
    with _timed("linear_ln", 2.0 * M * N * K):      # its own bucket: not the plain GEMM kernel of bench.py's roofline
        _lib.check(lib.pangu_linear_ln_residual_fwd(
            _stream(a), ap, lda, _chk(weight, "linear_ln.weight"), _chk(bias, "linear_ln.bias") if bias is not None else None,
            sp, lds, _chk(gamma, "gamma"), _chk(beta, "beta"), op, ldo, M, N, K, float(branch_scale)), "linear_ln_residual_fwd")
    return out


_WGRAD_WS_BYTES = 96 << 20
_wgrad_ws = {}      # (device, stream) -> fp32 scratch buffer of the two-stage weight-gradient reduction (shared with ops_bf16)


def release_stream_workspaces(stream=None):
    """Drop the scratch buffers keyed by `stream` (a torch.cuda.Stream about to go away: its handle may be recycled), or all of
    them (stream=None; the next launch on a stream allocates its buffer again).  train.GraphedTrainStep calls this around its
    capture: a buffer allocated while capturing lives in THAT graph's memory pool and must not be handed to anything else."""
    for key in [k for k in _wgrad_ws if stream is None or k[1] == stream.cuda_stream]:
        del _wgrad_ws[key]


def wgrad_workspace(device):
    """One 96-MB scratch buffer per (device, stream): launches of one stream are ordered, so they can share it; a backward
    running on a side stream gets its own instead of racing on the default stream's."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = torch.empty(_WGRAD_WS_BYTES // 4, dtype=torch.float32, device=device)
        _wgrad_ws[key] = ws
    return ws


def linear_wgrad(dc, a, want_bias=True, db_into=None):
    """dW[N,K] = dc[M,N]^T @ a[M,K], db[N] = colsum(dc), ADDED into zero-initialised buffers (db_into: add the column sums into this
    fp32 (N,) buffer, which already holds a partial bias gradient, instead); the token slabs' partial tiles
    travel through a per-(device, stream) scratch buffer (96 MB, allocated on first use; the few fp32 shapes whose slabs need 108 MB keep the atomic tail, measured level) and one reduce launch instead of fp32 atomics."""
    lib = _lib.load()
    ws = wgrad_workspace(dc.device)
    dp, lddc = _rows(dc, "wgrad.dc")
    ap, lda = _rows(a, "wgrad.a")
    M, N = dc.shape
    K = a.shape[1]
    own_db = want_bias and db_into is None
    buf = _zeros((N * K + (N if own_db else 0),), dc.device)   # one fill launch (none inside a zero_arena)
    dw = buf[:N * K].view(N, K)
    db = (buf[N * K:] if own_db else db_into) if want_bias else None
    esz = dc.element_size()
    for m0, m1 in (_row_chunks(M, esz * lddc, esz * lda) or [(0, M)]):      # the kernel ADDS into dw / db
        with _timed("wgrad", 2.0 * (m1 - m0) * N * K):
            _lib.check(lib.pangu_linear_wgrad_ws(_stream(dc), dp + m0 * lddc * esz, lddc, ap + m0 * lda * esz, lda, dw.data_ptr(),
                                                 db.data_ptr() if want_bias else None, m1 - m0, N, K, ws.data_ptr(),
                                                 _WGRAD_WS_BYTES), "linear_wgrad")
    return dw, db


def window_attention(qkv, qkv_bias, esb, Z, H, W, heads, shifted, want_lse=False, compact=False):
    """esb: the expanded (types, heads, 144, 144) table, or with compact=True the paper's compact table laid out
    (types, heads, 3312) (weights.compact_bias_table): same result bit for bit, 6.3x fewer bias bytes."""
    lib = _lib.load()
    N, C3 = qkv.shape
    C = C3 // 3
    if N != Z * H * W:
        raise RuntimeError(f"window_attention: {N} tokens != {Z}x{H}x{W}")
    types = (Z // 2) * ((H + 5) // 6)
    want = (types, heads, 3312) if compact else (types, heads, 144, 144)
    if tuple(esb.shape) != want:
        raise RuntimeError(f"window_attention: bias table {tuple(esb.shape)} != {want}")
    out = torch.empty((N, C), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((N, heads), dtype=torch.float32, device=qkv.device) if want_lse else None
    Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144          # padded token count: the core's FLOPs (4*Np*144*C)
    with _timed("attn", 4.0 * Np * 144 * C):
        fn = lib.pangu_window_attn_fwd_compact if compact else lib.pangu_window_attn_fwd
        _lib.check(fn(_stream(qkv), _chk(qkv, "qkv"), _chk(qkv_bias, "qkv_bias"), _chk(esb, "esb"), out.data_ptr(),
                      lse.data_ptr() if want_lse else None, Z, H, W, C, heads, int(shifted)), "window_attn_fwd")
    return (out, lse) if want_lse else out


def attention_windows(qkv, esb, mask, n_lon, types, heads):
    """EarthAttention3D's core on a PARTITIONED tensor (reference layers.py:368-415): qkv (n_lon*types*144, 3C) in window-slot
    order, esb (types, heads, 144, 144), mask None | (n_lon, types, 144, 144) | (types, 144, 144) fp32 -> (n_lon*types*144, C)."""
    lib = _lib.load()
    M, C3 = qkv.shape
    C = C3 // 3
    if M != n_lon * types * 144 or tuple(esb.shape) != (types, heads, 144, 144):
        raise RuntimeError(f"attention_windows: qkv {tuple(qkv.shape)}, esb {tuple(esb.shape)} vs {n_lon} x {types} windows")
    stride = 0
    if mask is not None:
        if tuple(mask.shape) == (n_lon, types, 144, 144):
            stride = types * 144 * 144
        elif tuple(mask.shape) != (types, 144, 144):
            raise RuntimeError(f"attention_windows: mask {tuple(mask.shape)} is neither ({n_lon}, {types}, 144, 144) nor ({types}, 144, 144)")
    out = torch.empty((M, C), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.pangu_attn_windows_fwd(_stream(qkv), _chk(qkv, "qkv"), _chk(esb, "esb"),
                                          _chk(mask, "mask") if mask is not None else None, stride, out.data_ptr(), n_lon, types,
                                          heads, C), "attn_windows_fwd")
    return out


def attention_windows_bwd(qkv, esb, mask, dout, n_lon, types, heads):
    """Backward of `attention_windows`: -> (dqkv (rows, 3C), d_esb (types, heads, 144, 144)); the mask gets no gradient."""
    lib = _lib.load()
    M, C3 = qkv.shape
    C = C3 // 3
    if M != n_lon * types * 144 or tuple(esb.shape) != (types, heads, 144, 144) or tuple(dout.shape) != (M, C):
        raise RuntimeError(f"attention_windows_bwd: qkv {tuple(qkv.shape)}, esb {tuple(esb.shape)}, dout {tuple(dout.shape)}")
    stride = 0
    if mask is not None:
        if tuple(mask.shape) == (n_lon, types, 144, 144):
            stride = types * 144 * 144
        elif tuple(mask.shape) != (types, 144, 144):
            raise RuntimeError(f"attention_windows_bwd: mask {tuple(mask.shape)}")
    dqkv = torch.empty_like(qkv)
    desb = torch.empty_like(esb)
    _lib.check(lib.pangu_attn_windows_bwd(_stream(qkv), _chk(qkv, "qkv"), _chk(esb, "esb"),
                                          _chk(mask, "mask") if mask is not None else None, stride, _chk(dout, "dout"),
                                          dqkv.data_ptr(), desb.data_ptr(), n_lon, types, heads, C), "attn_windows_bwd")
    return dqkv, desb


def window_attention_bwd(qkv, qkv_bias, esb, out, lse, dout, Z, H, W, heads, shifted, desb_out=None):
    """-> (dqkv [N,3C], dqkv_bias [3C] (pad-slot part of linear1.bias' gradient), d_esb like esb; written into
    `desb_out` (contiguous fp32, esb.numel() elements) when given)."""
    lib = _lib.load()
    N, C3 = qkv.shape
    C = C3 // 3
    dqkv = torch.empty_like(qkv)
    dqb = _zeros((C3,), qkv.device)
    desb = torch.empty_like(esb) if desb_out is None else desb_out.view(esb.shape)
    Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
    with _timed("attn_bwd", 14.0 * Np * 144 * C):
        _lib.check(lib.pangu_window_attn_bwd(_stream(qkv), _chk(qkv, "qkv"), _chk(qkv_bias, "qkv_bias"), _chk(esb, "esb"),
                                             _chk(out, "out"), _chk(lse, "lse"), _chk(dout, "dout"), dqkv.data_ptr(),
                                             dqb.data_ptr(), desb.data_ptr(), Z, H, W, C, heads, int(shifted)),
                   "window_attn_bwd")
    return dqkv, dqb, desb


def ln_residual_bwd(dout, y, gamma, branch_scale=1.0):
    """-> (dy, dgamma, dbeta) of out = shortcut + scale*LN(y); dout may be row-strided."""
    lib = _lib.load()
    N, C = y.shape
    dp, lddo = _rows(dout, "ln_bwd.dout")
    dy = torch.empty_like(y)
    dg, db = _zeros((2, C), y.device).unbind(0)
    _lib.check(lib.pangu_ln_residual_bwd(_stream(dout), dp, lddo, _chk(y, "ln_bwd.y"), _chk(gamma, "gamma"), dy.data_ptr(),
                                         dg.data_ptr(), db.data_ptr(), N, C, float(branch_scale)), "ln_residual_bwd")
    return dy, dg, db


def downsample_ln_bwd(dout, x, gamma, Z, H, W, add=None):
    """add (optional, dense (Z*H*W, C)): a second gradient of the same tokens, summed into dx by the kernel."""
    lib = _lib.load()
    xp, ldx = _rows(x, "downsample_bwd.x")
    C = x.shape[1]
    dx = torch.empty((Z * H * W, C), dtype=torch.float32, device=x.device)
    if add is not None and tuple(add.shape) != tuple(dx.shape):
        raise RuntimeError(f"downsample_ln_bwd: addend {tuple(add.shape)} != {tuple(dx.shape)}")
    dg, db = _zeros((2, 4 * C), x.device).unbind(0)
    _lib.check(lib.pangu_downsample_ln_bwd(_stream(dout), _chk(dout, "dout"), xp, ldx, _chk(gamma, "gamma"), dx.data_ptr(),
                                           dg.data_ptr(), db.data_ptr(), Z, H, W, C,
                                           _chk(add, "downsample_bwd.add") if add is not None else None), "downsample_ln_bwd")
    return dx, dg, db


def upsample_ln_bwd(dout, y, gamma, Z, H2, W2, H):
    lib = _lib.load()
    Co = y.shape[1] // 4
    dy = torch.empty_like(y)
    dg, db = _zeros((2, Co), y.device).unbind(0)
    _lib.check(lib.pangu_upsample_ln_bwd(_stream(dout), _chk(dout, "dout"), _chk(y, "y"), _chk(gamma, "gamma"),
                                         dy.data_ptr(), dg.data_ptr(), db.data_ptr(), Z, H2, W2, H, Co),
               "upsample_ln_bwd")
    return dy, dg, db


def patch_recover_gather_bwd(d_out, d_out_s):
    lib = _lib.load()
    LAT, LON = d_out.shape[-2], d_out.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    dy_u = torch.empty((7 * H4 * W4, 160), dtype=torch.float32, device=d_out.device)
    dy_s = torch.empty((H4 * W4, 64), dtype=torch.float32, device=d_out.device)
    _lib.check(lib.pangu_patch_recover_gather_bwd(_stream(d_out), _chk(d_out, "d_output"), _chk(d_out_s, "d_output_surface"),
                                                  dy_u.data_ptr(), dy_s.data_ptr(), LAT, LON),
               "patch_recover_gather_bwd")
    return dy_u, dy_s


def ln_residual(y, shortcut, gamma, beta, out=None, branch_scale=1.0, want_stats=False):
    lib = _lib.load()
    N, C = y.shape
    sp, lds = _rows(shortcut, "ln_residual.shortcut")
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=y.device)
    op, ldo = _rows(out, "ln_residual.out")
    stats = torch.empty((N, 2), dtype=torch.float32, device=y.device) if want_stats else None
    _lib.check(lib.pangu_ln_residual_fwd(_stream(y), _chk(y, "ln_residual.y"), sp, lds, _chk(gamma, "gamma"),
                                         _chk(beta, "beta"), op, ldo, stats.data_ptr() if want_stats else None, N, C,
                                         float(branch_scale)), "ln_residual_fwd")
    return (out, stats) if want_stats else out


def downsample_ln(x, gamma, beta, Z, H, W, want_stats=False):
    lib = _lib.load()
    xp, ldx = _rows(x, "downsample.x")
    C = x.shape[1]
    rows = Z * ((H + 1) // 2) * (W // 2)
    out = torch.empty((rows, 4 * C), dtype=torch.float32, device=x.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if want_stats else None
    _lib.check(lib.pangu_downsample_ln_fwd(_stream(x), xp, ldx, _chk(gamma, "gamma"), _chk(beta, "beta"), out.data_ptr(),
                                           stats.data_ptr() if want_stats else None, Z, H, W, C), "downsample_ln_fwd")
    return (out, stats) if want_stats else out


def upsample_ln(y, gamma, beta, Z, H2, W2, H, want_stats=False):
    lib = _lib.load()
    Co = y.shape[1] // 4
    rows = Z * H * 2 * W2
    out = torch.empty((rows, Co), dtype=torch.float32, device=y.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=y.device) if want_stats else None
    _lib.check(lib.pangu_upsample_ln_fwd(_stream(y), _chk(y, "upsample.y"), _chk(gamma, "gamma"), _chk(beta, "beta"),
                                         out.data_ptr(), stats.data_ptr() if want_stats else None, Z, H2, W2, H, Co),
               "upsample_ln_fwd")
    return (out, stats) if want_stats else out


def patch_embed_gather(inp, inp_surface, s_mean, s_std, u_mean, u_std, maps, const_h, levels_reversed=False):
    """One sample: inp (5,13,LAT,LON), inp_surface (4,LAT,LON) -> (a_surface [H4*W4,112], a_upper [7*H4*W4,192]).
    levels_reversed: `inp` is stored with ascending levels (as on disk); the reader's reversal (reference
    era5_data/utils_data.py:117) is done by the kernel's addressing."""
    lib = _lib.load()
    LAT, LON = inp.shape[-2], inp.shape[-1]
    H4, W4 = (LAT + 3) // 4, LON // 4
    a_s = torch.empty((H4 * W4, 112), dtype=torch.float32, device=inp.device)
    a_u = torch.empty((7 * H4 * W4, 192), dtype=torch.float32, device=inp.device)
    _lib.check(lib.pangu_patch_embed_gather(_stream(inp), _chk(inp, "input"), _chk(inp_surface, "input_surface"),
                                            _chk(s_mean, "surface_mean"), _chk(s_std, "surface_std"),
                                            _chk(u_mean, "upper_mean"), _chk(u_std, "upper_std"), _chk(maps, "maps"),
                                            _chk(const_h, "const_h"), a_s.data_ptr(), a_u.data_ptr(), LAT, LON,
                                            int(bool(levels_reversed))), "patch_embed_gather")
    return a_s, a_u


def patch_embed_gather_bwd(da_s, da_u, s_std, u_std, LAT, LON, levels_reversed=False):
    """Adjoint of patch_embed_gather w.r.t. the raw fields: da_s [H4*W4, 64], da_u [7*H4*W4, 160] fp32 (the A-matrix columns that
    came from the fields) -> (d_input (5,13,LAT,LON), d_input_surface (4,LAT,LON)), each divided by the std the forward divided by
    (autograd of reference models/layers.py:48-55,71-76)."""
    lib = _lib.load()
    H4, W4 = (LAT + 3) // 4, LON // 4
    if da_s.shape != (H4 * W4, 64) or da_u.shape != (7 * H4 * W4, 160):
        raise RuntimeError(f"patch_embed_gather_bwd: da_s {tuple(da_s.shape)} da_u {tuple(da_u.shape)} for {LAT}x{LON}")
    d_in = torch.empty((5, 13, LAT, LON), dtype=torch.float32, device=da_u.device)
    d_in_s = torch.empty((4, LAT, LON), dtype=torch.float32, device=da_u.device)
    _lib.check(lib.pangu_patch_embed_gather_bwd(_stream(da_u), _chk(da_s, "da_surface"), _chk(da_u, "da_upper"),
                                                _chk(s_std, "surface_std"), _chk(u_std, "upper_std"), d_in.data_ptr(),
                                                d_in_s.data_ptr(), LAT, LON, int(bool(levels_reversed))), "patch_embed_gather_bwd")
    return d_in, d_in_s


_denorm_target = None      # (phys_upper (B,5,13,LAT,LON), phys_surface (B,4,LAT,LON), (u_mean, u_std, s_mean, s_std) flat fp32, [sample counter])


class scatter_denorm:
    """Rollout: inside this context every `patch_recover_scatter` (the last kernel of a forward) ALSO writes its fields in
    physical units (reference era5_data/utils_data.py:324-330 `normBackData`: out * std + mean, a multiply then an add) into
    `phys_upper[b]` / `phys_surface[b]` -- the next step's input buffers -- instead of four elementwise passes over the 286 MB
    afterwards.  stats_last = (s_mean (1,4,1,1), s_std, u_mean (1,5,13,1,1), u_std)."""

    def __init__(self, phys_upper, phys_surface, stats_last):
        s_mean, s_std, u_mean, u_std = stats_last
        f = lambda t: t.to(device=phys_upper.device, dtype=torch.float32).reshape(-1).contiguous()
        self.target = [phys_upper, phys_surface, (f(u_mean), f(u_std), f(s_mean), f(s_std)), 0]
        if self.target[2][0].numel() != 65 or self.target[2][2].numel() != 4:
            raise RuntimeError("scatter_denorm: stats_last must be (s_mean (1,4,1,1), s_std, u_mean (1,5,13,1,1), u_std)")

    def __enter__(self):
        global _denorm_target
        self.prev, _denorm_target = _denorm_target, self.target
        self.target[3] = 0
        return self

    def __exit__(self, *exc):
        global _denorm_target
        _denorm_target = self.prev


def patch_recover_scatter(y_upper, y_surface, LAT, LON):
    lib = _lib.load()
    out = torch.empty((5, 13, LAT, LON), dtype=torch.float32, device=y_upper.device)
    out_s = torch.empty((4, LAT, LON), dtype=torch.float32, device=y_upper.device)
    tgt = _denorm_target
    if tgt is not None:
        b = tgt[3]
        tgt[3] += 1
        pu, ps = tgt[0][b], tgt[1][b]
        if tuple(pu.shape) != (5, 13, LAT, LON) or not pu.is_contiguous() or not ps.is_contiguous():
            raise RuntimeError(f"scatter_denorm target {tuple(pu.shape)} does not match the (5, 13, {LAT}, {LON}) fields")
        um, us, sm, ss = tgt[2]
        _lib.check(lib.pangu_patch_recover_scatter_denorm(
            _stream(y_upper), _chk(y_upper, "y_upper"), _chk(y_surface, "y_surface"), out.data_ptr(), out_s.data_ptr(), pu.data_ptr(),
            ps.data_ptr(), um.data_ptr(), us.data_ptr(), sm.data_ptr(), ss.data_ptr(), LAT, LON), "patch_recover_scatter_denorm")
        return out, out_s
    _lib.check(lib.pangu_patch_recover_scatter(_stream(y_upper), _chk(y_upper, "y_upper"), _chk(y_surface, "y_surface"),
                                               out.data_ptr(), out_s.data_ptr(), LAT, LON), "patch_recover_scatter")
    return out, out_s

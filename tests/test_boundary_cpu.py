"""Host-side hardening of the drop-in boundary (no GPU): the launch-device guard and the refusal of module wrappers / hooks
that the kernels would bypass (reference finetune/lora_tune.py:124-135 wraps `linear1`; the HIP path never calls it)."""
import pytest
import torch
from torch import nn

import pangu_pytorch_amd as P
from pangu_pytorch_amd import ops


def test_stream_guard_refuses_cpu_tensor():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops._stream(torch.zeros(4))


def test_stream_guard_refuses_other_device(monkeypatch):
    """A tensor on cuda:1 while the current device is cuda:0 (an 8-GPU node driven from one process) must not be launched."""
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    with pytest.raises(RuntimeError, match="current device is cuda:0"):
        ops._stream(torch.device("cuda", 1))


def test_same_device():
    a, b = torch.zeros(1), torch.zeros(1, device="meta")
    assert ops.same_device(a, None, a) == a.device
    with pytest.raises(RuntimeError, match="different devices"):
        ops.same_device(a, b)


@pytest.fixture(scope="module")
def model():
    return P.PanguModel()


def test_plain_tree_passes(model):
    model._assert_plain_children()
    h = model.register_forward_hook(lambda *a: None)         # a hook on the model itself DOES run: allowed
    model._assert_plain_children()
    h.remove()


def test_forward_hook_on_child_is_refused(model):
    lin = model.layers[1].blocks[2].linear.linear1
    h = lin.register_forward_hook(lambda *a: None)
    try:
        with pytest.raises(RuntimeError, match="forward hooks"):
            model._assert_plain_children()
    finally:
        h.remove()
    h = model.layers[0].blocks[0].attention.register_forward_pre_hook(lambda *a: None)
    try:
        with pytest.raises(RuntimeError, match="forward hooks"):
            model._assert_plain_children()
    finally:
        h.remove()
    # full backward hooks on a sub-module are bypassed just the same (its forward never runs) -- refused too
    h = model.layers[2].blocks[1].norm1.register_full_backward_hook(lambda *a: None)
    try:
        with pytest.raises(RuntimeError, match="backward hooks"):
            model._assert_plain_children()
    finally:
        h.remove()
    model._assert_plain_children()


class _LoraLinear(nn.Module):
    """Shape of a peft `lora.Linear`: wraps the base layer and adds a low-rank path in ITS forward."""

    def __init__(self, base, r=4):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.Linear(base.in_features, r, bias=False)
        self.lora_B = nn.Linear(r, base.out_features, bias=False)

    def forward(self, x):
        return self.base_layer(x) + self.lora_B(self.lora_A(x))


def test_lora_style_wrapper_is_refused():
    m = P.PanguModel()
    att = m.layers[0].blocks[1].attention
    att.linear1 = _LoraLinear(att.linear1)
    with pytest.raises(RuntimeError, match="LoRA"):
        m._assert_plain_children()
    blk = P.layers.EarthSpecificBlock(192, 0.0, 6)
    blk.linear.linear2 = _LoraLinear(blk.linear.linear2)
    with pytest.raises(RuntimeError, match="EarthSpecificBlock"):
        blk(torch.zeros(1, 8 * 181 * 12, 192), 8, 181, 12, False)


class _MyLinear(nn.Linear):
    def forward(self, x):
        return super().forward(x) * 2


def test_subclassed_linear_is_refused():
    blk = P.layers.EarthSpecificBlock(384, 0.0, 12)
    blk.attention.linear2 = _MyLinear(384, 384)
    with pytest.raises(RuntimeError, match="_MyLinear"):
        P.layers.assert_plain_tree(blk, "EarthSpecificBlock")


def test_earth_attention_forward_checks_its_windows():
    att = P.layers.EarthAttention3D(192, 6, 0, (2, 6, 12))
    with torch.no_grad(), pytest.raises(RuntimeError, match="windows"):
        att(torch.zeros(1, 100, 144, 192), None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # right shape, CPU tensor: refused at the first launch
        att(torch.zeros(1, 124, 144, 192), None)


def test_bench_gpus_n_without_a_launcher_spawns_child_ranks_and_relays_failure():
    """VERDICT r4 item 1a, the part that runs without a GPU: a plain `python bench.py --gpus 2` (no WORLD_SIZE) must not die with
    "launch with torch.distributed.run": the parent (torch not imported, GPU untouched) starts the ranks as a child process and relays
    their return code.  Here the ranks fail (no HIP device in this container), so: rc != 0, no JSON line, and the failure is the
    ranks', not the old SystemExit text."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PANGU_DIST_BACKEND"] = "gloo"
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the launcher-less 2-rank run itself is tests/test_gpu_bench_dist.py")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-train",
                        "--no-bf16", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "launch with" not in r.stderr and "torch.distributed" in r.stderr      # the child launcher ran and reported its ranks


def test_eval_recompute_function_plumbs_gradients_like_plain_autograd():
    """pangu_model._EvalRecomputeFn on a stand-in module (CPU): the forward runs `_forward_dispatch(.., grad_path=False)` under no_grad
    and returns outputs that require grad; the backward re-runs it with grad_path=True and hands every trainable parameter the gradient
    plain autograd gives, frozen parameters None -- also when only ONE of the two outputs is used by the loss; the two FIELDS get
    their gradients when they ask for them, and overwriting a field in place between forward and backward raises
    (the recompute would differentiate a different function)."""
    import pytest
    import torch
    from pangu_pytorch_amd.pangu_model import _EvalRecomputeFn

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.tensor([1.5, -2.0, 0.5]))
            self.b = torch.nn.Parameter(torch.tensor([0.25]))
            self.frozen = torch.nn.Parameter(torch.tensor([3.0]), requires_grad=False)
            self.calls = []

        def _forward_dispatch(self, x, xs, stats, maps, const_h, want_bf16, grad_path, levels_reversed=False):
            self.calls.append((grad_path, torch.is_grad_enabled(), levels_reversed))
            return (x * self.a).sin() * self.b + self.frozen, (xs * self.a.sum()).cos()

    m = Stub().eval()
    x, xs = torch.tensor([0.3, 0.7, -1.1]), torch.tensor([0.2, 0.9])
    consts = (None, None, None, False, True)
    out, out_s = _EvalRecomputeFn.apply(m, consts, x, xs, *m.parameters())
    assert m.calls == [(False, False, True)] and out.requires_grad and out_s.requires_grad
    (out.sum() * 2.0 + (out_s ** 2).sum()).backward()
    assert m.calls == [(False, False, True), (True, True, True)]
    got = [p.grad.clone() if p.grad is not None else None for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    o, o_s = m._forward_dispatch(x, xs, None, None, None, False, True)
    (o.sum() * 2.0 + (o_s ** 2).sum()).backward()
    for g, p in zip(got, m.parameters()):
        if p.requires_grad:
            assert torch.allclose(g, p.grad, rtol=1e-6, atol=1e-7)
        else:
            assert g is None and p.grad is None
    # one output unused: its cotangent is materialised as zeros
    m.zero_grad(set_to_none=True)
    out, _ = _EvalRecomputeFn.apply(m, consts, x, xs, *m.parameters())
    out.sum().backward()
    ga = m.a.grad.clone()
    m.zero_grad(set_to_none=True)
    m._forward_dispatch(x, xs, None, None, None, False, True)[0].sum().backward()
    assert torch.allclose(ga, m.a.grad, rtol=1e-6, atol=1e-7)
    # the fields' own gradients: only the one that asks gets one
    m.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    out, out_s = _EvalRecomputeFn.apply(m, consts, xg, xs, *m.parameters())
    (out.sum() * 2.0 + (out_s ** 2).sum()).backward()
    xr = x.clone().requires_grad_(True)
    o, o_s = m._forward_dispatch(xr, xs, None, None, None, False, True)
    (gx,) = torch.autograd.grad(o.sum() * 2.0 + (o_s ** 2).sum(), xr)
    assert torch.allclose(xg.grad, gx, rtol=1e-6, atol=1e-7) and xs.grad is None
    # a field overwritten in place before the backward: refused
    m.zero_grad(set_to_none=True)
    x2 = x.clone()
    out, _ = _EvalRecomputeFn.apply(m, consts, x2, xs, *m.parameters())
    x2.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()


def test_dropped_branch_policy_scopes_are_counted_not_saved_and_restored():
    """`ops.dropped_branch_grads` scopes entered from two training loops (two threads) in interleaved order must leave the
    process at the DEFAULT policy ("zeros": the reference's autograd result for a DropPath-dropped branch), and a "zeros" scope wins
    over a concurrent "none" scope (it only materialises what the reference materialises)."""
    from pangu_pytorch_amd import ops
    assert ops.dropped_branch_policy() == "zeros"
    a, b = ops.dropped_branch_grads("none"), ops.dropped_branch_grads("none")
    a.__enter__()
    assert ops.dropped_branch_policy() == "none"
    b.__enter__()
    a.__exit__(None, None, None)                      # A leaves while B is still inside
    assert ops.dropped_branch_policy() == "none"
    b.__exit__(None, None, None)
    assert ops.dropped_branch_policy() == "zeros"     # (save-and-restore would have left "none" here)
    with ops.dropped_branch_grads("none"):
        with ops.dropped_branch_grads("zeros"):
            assert ops.dropped_branch_policy() == "zeros"
        assert ops.dropped_branch_policy() == "none"
    assert ops.dropped_branch_policy() == "zeros"


def test_train_step_keeps_zero_gradients_for_a_foreign_grad_sync():
    """train_step may drop the explicit zero gradients of DropPath-dropped branches only when the gradient
    sync is None or a dist.FlatGradSync method (flat buffer: the zeros are there, every rank launches every bucket).  Any other
    callable -- the repo's own API-parity dist.gather_grad all-reduces parameter by parameter and skips `p.grad is None` -- must
    see materialised zeros, or ranks that drew different DropPath patterns issue different numbers of collectives."""
    import torch
    from pangu_pytorch_amd import dist, train
    m = torch.nn.Linear(4, 4)
    sync = dist.FlatGradSync(m)
    try:
        assert train._owns_dropped_branches(sync.finish)
        assert not train._owns_dropped_branches(lambda: dist.gather_grad(m.parameters(), 1))
        assert not train._owns_dropped_branches(dist.gather_grad)
        assert not train._owns_dropped_branches(None)
    finally:
        sync.remove()
    # bucket quantum: 16 bytes per rank whatever the element size (4 ELEMENTS are 8 bytes for bf16 parameters)
    mb = torch.nn.Linear(5, 3).to(torch.bfloat16)
    sb = dist.FlatGradSync(mb)
    try:
        assert all((end - start) * 2 % 16 == 0 and start * 2 % 16 == 0 for start, end, _ in sb.buckets)
    finally:
        sb.remove()

"""Round-4 hardening of the drop-in boundary on the GPU: EarthAttention3D's own forward, the launch-device guard, partially
frozen fine-tunes, optimizer-state replacement under HipAdam, in-place weight edits after an optimizer step."""
import copy
import os

import numpy as np
import pytest
import torch

import cases
import pangu_oracle as O
import synth

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def P():
    import pangu_pytorch_amd as P
    assert torch.cuda.is_available()
    P._lib.load()
    return P


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("roll", [False, True])
def test_earth_attention3d_forward_on_windows(P, golden_dir, C, roll):
    """`blk.attention(x_window, mask)` -- the reference module's own calling convention (layers.py:360-421) -- through the
    kernels: == the oracle's restatement and the reference's own output (tests/golden/attn_windows.npz), every slot an
    ordinary token (non-zero data in what would be the block's pad rows), mask = the tensor gen_mask returns."""
    g = np.load(os.path.join(golden_dir, "attn_windows.npz"))
    st = cases.STAGES[C]
    blk = P.layers.EarthSpecificBlock(C, 0.0, st["heads"], device="cuda").cuda().eval()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, s, "cuda") for k, s in cases.block_param_shapes(C).items()})
    xw = cases.attention_window_input(C, 2, "cuda")
    mask = blk.gen_mask(torch.zeros(1, st["Z"], st["H"] + 5, 24, C, device="cuda")) if roll else None
    with torch.no_grad():
        y = blk.attention(xw, mask)
    assert y.shape == xw.shape and y.dtype == torch.float32
    assert cases.compare_summary(y, g, f"attn_windows_{C}_{int(roll)}.out", 1e-4) < 1e-4
    p = cases.block_params(C, roll)
    ref = O.attention_windows(xw.cpu(), p[pre + "attention.linear1.weight"], p[pre + "attention.linear1.bias"],
                              p[pre + "attention.linear2.weight"], p[pre + "attention.linear2.bias"],
                              p[pre + "attention.earth_specific_bias"], mask.cpu() if roll else None)
    assert ((y.cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-4
    # a FOREIGN mask (not the Earth-specific one) is honoured too: the kernel reads the tensor it is given
    if roll:
        m2 = synth.uniform(tuple(mask.shape), 77, 3.0)
        with torch.no_grad():
            y2 = blk.attention(xw, m2.cuda())
        ref2 = O.attention_windows(xw.cpu(), p[pre + "attention.linear1.weight"], p[pre + "attention.linear1.bias"],
                                   p[pre + "attention.linear2.weight"], p[pre + "attention.linear2.bias"],
                                   p[pre + "attention.earth_specific_bias"], m2)
        assert ((y2.cpu() - ref2).abs().max() / ref2.abs().max()).item() < 1e-4


@pytest.mark.parametrize("C,roll", [(192, True), (384, False)])
def test_earth_attention3d_module_backward(P, C, roll):
    """The module on its own is differentiable like the reference's (layers.py:360-421 under autograd): gradients of the input
    windows and of all five parameters == torch autograd over the oracle's restatement."""
    st = cases.STAGES[C]
    blk = P.layers.EarthSpecificBlock(C, 0.0, st["heads"], device="cuda").cuda().eval()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, s, "cuda") for k, s in cases.block_param_shapes(C).items()})
    att = blk.attention
    xw = cases.attention_window_input(C, 2, "cuda").requires_grad_(True)
    mask = blk.gen_mask(torch.zeros(1, st["Z"], st["H"] + 5, 24, C, device="cuda")) if roll else None
    y = att(xw, mask)
    cot = cases.cotangent("attn_windows", y.shape, "cuda")
    (y * cot).sum().backward()
    p = {k: v.requires_grad_(True) for k, v in cases.block_params(C, roll).items()}
    xr = xw.detach().cpu().requires_grad_(True)
    names = ("attention.linear1.weight", "attention.linear1.bias", "attention.linear2.weight", "attention.linear2.bias",
             "attention.earth_specific_bias")
    ref = O.attention_windows(xr, *(p[pre + n] for n in names), mask.cpu() if roll else None)
    (ref * cot.cpu()).sum().backward()
    rel = lambda a, b: ((a.detach().cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(y, ref.detach()) < 1e-4 and rel(xw.grad, xr.grad) < 1e-4
    for n in names:
        q = dict(blk.named_parameters())[n]
        assert q.grad is not None and rel(q.grad, p[pre + n].grad) < 1e-4, n


def test_mlp_module_backward(P):
    """`blk.linear(x)` (Mlp.forward on its own, reference layers.py:264-270) under autograd == torch on the same weights."""
    C = 192
    m = P.layers.Mlp(C, 0).cuda()
    x = synth.uniform((3, 50, C), 5, 1.5).cuda().requires_grad_(True)
    y = m(x)
    cot = synth.uniform(tuple(y.shape), 6).cuda()
    (y * cot).sum().backward()
    xr = x.detach().clone().requires_grad_(True)
    w1, b1, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (m.linear1.weight, m.linear1.bias, m.linear2.weight, m.linear2.bias))
    ref = torch.nn.functional.gelu(xr @ w1.t() + b1) @ w2.t() + b2
    (ref * cot).sum().backward()
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(y.detach(), ref.detach()) < 1e-4 and rel(x.grad, xr.grad) < 1e-4
    for q, r in ((m.linear1.weight, w1), (m.linear1.bias, b1), (m.linear2.weight, w2), (m.linear2.bias, b2)):
        assert rel(q.grad, r.grad) < 1e-4


def test_block_matches_module_pieces(P):
    """The block's fused path == the reference's own composition partition -> attention module -> reverse, with the module
    forward above in the middle (pad rows zero, as the block hands them over)."""
    C, W, roll = 192, 24, True
    st = cases.STAGES[C]
    Z, H = st["Z"], st["H"]
    blk = P.layers.EarthSpecificBlock(C, 0.0, st["heads"], device="cuda").cuda().eval()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, s, "cuda") for k, s in cases.block_param_shapes(C).items()})
    x = cases.block_input(C, W, "cuda")
    from pangu_pytorch_amd import ops
    idx = ops.window_index(Z, H, W, roll, "cuda").long()                    # (nLon, types, 144), -1 = pad
    xp = torch.cat((x[0], torch.zeros(1, C, device="cuda")), 0)
    xw = xp[torch.where(idx < 0, torch.full_like(idx, x.shape[1]), idx)]    # (nLon, types, 144, C)
    mask = blk.gen_mask(torch.zeros(1, Z, H + 5, W, C, device="cuda"))
    with torch.no_grad():
        yw = blk.attention(xw, mask)
        a = torch.zeros(x.shape[1] + 1, C, device="cuda")
        a[torch.where(idx < 0, torch.full_like(idx, x.shape[1]), idx).flatten()] = yw.view(-1, C)
        a = a[:-1]
        x1 = x[0] + torch.nn.functional.layer_norm(a, (C,), blk.norm1.weight, blk.norm1.bias)
        m = torch.nn.functional.gelu(x1 @ blk.linear.linear1.weight.t() + blk.linear.linear1.bias) @ blk.linear.linear2.weight.t() \
            + blk.linear.linear2.bias
        want = x1 + torch.nn.functional.layer_norm(m, (C,), blk.norm2.weight, blk.norm2.bias)
        got = blk(x, Z, H, W, roll)[0]
    assert ((got - want).abs().max() / want.abs().max()).item() < 1e-4


def test_launch_on_non_current_device(P):
    """a model on cuda:1 in a process whose current device is cuda:0.  PanguModel / the block enter the
    tensors' device themselves; a bare op call is refused."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    from pangu_pytorch_amd import ops
    assert torch.cuda.current_device() == 0
    a = synth.uniform((256, 192), 5).to("cuda:1")
    w = synth.uniform((192, 192), 6).to("cuda:1")
    with pytest.raises(RuntimeError, match="current device is cuda:0"):
        ops.linear(a, w)
    with torch.cuda.device(1):
        got = ops.linear(a, w)
    assert ((got.cpu() - a.cpu() @ w.cpu().t()).abs().max()).item() < 1e-3
    m = P.PanguModel(device="cuda:1").to("cuda:1").eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda:1")
    with torch.no_grad():
        out, _ = m(inp, inp_s, stats, maps, const_h)
        with torch.cuda.device(1):
            ref, _ = m(inp, inp_s, stats, maps, const_h)
    assert out.device == inp.device and torch.equal(out, ref)
    assert torch.cuda.current_device() == 0


def test_inputs_on_cpu_or_other_device_are_refused(P):
    m = P.PanguModel(device="cuda").cuda().eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(inp, inp_s, stats, maps, const_h)


@pytest.mark.parametrize("trainable", ["_output_layer", "layers.EarthSpecificLayer3.blocks.EarthSpecificBlock1.linear", "upsample"])
def test_partially_frozen_finetune_f32(P, trainable):
    """B = 1 fp32 with everything frozen except one late module -- the last blocks of layer 0 (and, for the
    output-layer case, layer 3) take the INFERENCE path while the model is on its autograd route and hands them the 2-D halves
    of the skip-concat buffer.  The trainable parameters' gradients equal those of the fully trainable model."""
    from pangu_pytorch_amd import train
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    tgt, tgt_s = cases.model_targets("cuda")
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    train.weighted_l1_loss(out, out_s, tgt, tgt_s).backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if k.startswith(trainable)}
    ref_out = out.detach().clone()
    assert ref
    m.zero_grad(set_to_none=True)
    del out, out_s
    for k, p in m.named_parameters():
        p.requires_grad_(k.startswith(trainable))
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    # (the frozen blocks run the inference kernels -- projection + LayerNorm fused -- not the training path's: same values, not same bits)
    assert ((out.detach() - ref_out).abs().max() / ref_out.abs().max()).item() < 1e-5
    train.weighted_l1_loss(out, out_s, tgt, tgt_s).backward()
    for k, p in m.named_parameters():
        if k.startswith(trainable):
            assert p.grad is not None and ((p.grad - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)).item() < 1e-4, k
        else:
            assert p.grad is None, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_forward_with_grad_is_the_inference_forward_and_recomputes_for_backward(P, dtype):
    """The reference's `test()` calls the model in eval() WITHOUT no_grad (models/pangu_sample.py:197-202).  Default
    `eval_grad_mode = "recompute"`: that call runs the inference kernels (bit-identical to the no_grad forward, no saved activations)
    and a backward, should one arrive, re-runs the autograd forward: gradients equal those of `eval_grad_mode = "save"` (the
    activation-saving forward at once) -- identical kernels on identical inputs, accumulation order of the fp32 atomics aside."""
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    m.set_compute_dtype(dtype)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        ref, ref_s = m(inp, inp_s, stats, maps, const_h)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    assert m.eval_grad_mode == "recompute"
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    assert out.requires_grad and torch.equal(out, ref) and torch.equal(out_s, ref_s)
    held = (torch.cuda.memory_allocated() - base) / 2**30
    assert held < 1.0, held                      # the two output fields (0.29 GB), not 30 / 66 GB of activations
    # a FIXED cotangent (not the L1 loss, whose sign(o - t) would be taken on two slightly different outputs): the backward is then
    # the same graph on the same inputs in both modes -- the autograd forward is what gets differentiated either way
    cot, cot_s = cases.cotangent("evalrc", out.shape, "cuda"), cases.cotangent("evalrc_s", out_s.shape, "cuda")
    ((out * cot).sum() + (out_s * cot_s).sum()).backward()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(g is not None for g in got.values())
    m.zero_grad(set_to_none=True)
    del out, out_s
    m.eval_grad_mode = "save"
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    ((out * cot).sum() + (out_s * cot_s).sum()).backward()
    worst, worst_l2 = (0.0, ""), (0.0, "")
    for k, p in m.named_parameters():
        e = ((p.grad - got[k]).abs().max() / p.grad.abs().max().clamp_min(1e-30)).item()
        l2 = ((p.grad - got[k]).double().norm() / p.grad.double().norm().clamp_min(1e-30)).item()
        worst, worst_l2 = max(worst, (e, k)), max(worst_l2, (l2, k))
    print("eval recompute vs save, worst gradient difference (max-abs / max, rel-L2):", worst, worst_l2)
    assert worst_l2[0] < 1e-4 and worst[0] < 1e-3, (worst, worst_l2)      # identical kernels; only the fp32 atomics' order differs
    m.set_compute_dtype(torch.float32)


def test_hip_adam_follows_replaced_state(P):
    """optimizer.load_state_dict(snapshot) after a step replaces exp_avg / exp_avg_sq while the parameters and
    (FlatGradSync-style) gradient buffers keep their addresses; the cached device job table must not keep the old moments."""
    from pangu_pytorch_amd import train
    torch.manual_seed(0)
    shapes = [(1000,), (33, 7), (4096,), (5,)]
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    grads = [torch.randn(s, device="cuda") for s in shapes]           # fixed gradient storage, as the flat buffer gives
    a = train.HipAdam(ps, lr=1e-2, weight_decay=1e-3)
    b = torch.optim.Adam(qs, lr=1e-2, weight_decay=1e-3, fused=True)

    def step(k):
        for p, q, g in zip(ps, qs, grads):
            g.copy_(torch.randn(g.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(k)))
            p.grad = g
            q.grad = g.clone()
        a.step()
        b.step()

    step(1)
    snap_a, snap_b = copy.deepcopy(a.state_dict()), copy.deepcopy(b.state_dict())
    snap_p = [p.detach().clone() for p in ps]
    step(2)
    step(3)
    a.load_state_dict(snap_a)                   # rollback: NEW moment tensors, same parameter / gradient addresses
    b.load_state_dict(snap_b)
    with torch.no_grad():
        for p, q, s in zip(ps, qs, snap_p):
            p.copy_(s)
            q.copy_(s)
    step(4)
    step(5)
    for p, q in zip(ps, qs):
        assert torch.equal(p.detach(), q.detach())
    for p, q in zip(ps, qs):
        assert torch.equal(a.state[p]["exp_avg"], b.state[q]["exp_avg"])
        assert torch.equal(a.state[p]["exp_avg_sq"], b.state[q]["exp_avg_sq"])
    # direct replacement of a moment tensor
    a.state[ps[0]]["exp_avg"] = a.state[ps[0]]["exp_avg"].clone()
    b.state[qs[0]]["exp_avg"] = b.state[qs[0]]["exp_avg"].clone()
    step(6)
    assert torch.equal(ps[0].detach(), qs[0].detach()) and torch.equal(a.state[ps[0]]["exp_avg"], b.state[qs[0]]["exp_avg"])
    # amsgrad / maximize coming in through a loaded state_dict are refused, not ignored
    sd = a.state_dict()
    sd["param_groups"][0]["amsgrad"] = True
    a.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="amsgrad"):
        a.step()


def test_hip_adam_bit_identical_over_many_steps(P):
    """ADVICE r3 (low) checked: sqrt(bias_correction2) is formed as ATen forms it (in double, fused_adam_utils.cuh:130-137,
    then narrowed to float): 40 steps, bit-identical."""
    from pangu_pytorch_amd import train
    torch.manual_seed(1)
    p = torch.nn.Parameter(torch.randn(5000, device="cuda"))
    q = torch.nn.Parameter(p.detach().clone())
    a = train.HipAdam([p], lr=1e-3, weight_decay=3e-6)
    b = torch.optim.Adam([q], lr=1e-3, weight_decay=3e-6, fused=True)
    for k in range(40):
        g = torch.randn(5000, device="cuda")
        p.grad, q.grad = g, g.clone()
        a.step()
        b.step()
        assert torch.equal(p.detach(), q.detach()), k


def test_bf16_shadows_follow_inplace_edit_after_hip_adam_step(P):
    """HipAdam writes the bf16 images of the plainly cast parameters itself and the next refresh skips them --
    unless the parameter was modified in place in between (`copy_` under no_grad: an EMA swap, a clamp), which bumps
    `_version`: then the image is re-made like every other copy."""
    from pangu_pytorch_amd import train
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    m.set_compute_dtype(BF)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    tgt, tgt_s = cases.model_targets("cuda")
    opt = train.HipAdam([p for p in m.parameters()], lr=1e-3, shadow_of=m)
    for _ in range(2):
        train.train_step(m, opt, (inp, inp_s, tgt, tgt_s), stats, maps, const_h)
    esb = m.layers[2].blocks[3].attention.earth_specific_bias
    w = m.layers[0].blocks[1].linear.linear1.weight
    with torch.no_grad():
        esb.copy_(esb * 0.5 + 0.25)             # after the last optimizer step, before the next forward
        w.mul_(1.5)
        out = m(inp, inp_s, stats, maps, const_h)[0].clone()
        assert torch.equal(m._shadow.get(esb), esb.detach()[0].to(BF))
        m.invalidate_shadows()
        fresh = m(inp, inp_s, stats, maps, const_h)[0]
    assert torch.equal(out, fresh)


@pytest.mark.parametrize("dt", [torch.float32, BF])
def test_graphed_train_step_matches_eager(P, dt):
    """train.GraphedTrainStep (fwd + loss + bwd replayed from a hipGraph, Adam eager; DropPath off) == the eager train_step: same
    losses over three optimisation steps (lr large enough that the weights move), and in bf16 the weight shadows the graph reads
    are re-made in place after every step (a forward on freshly rebuilt shadows gives the same bits)."""
    from pangu_pytorch_amd import train
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    tgt, tgt_s = cases.model_targets("cuda")
    losses = {}
    for how in ("eager", "graph"):
        m = P.PanguModel(device="cuda").cuda().eval()
        m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda", spec="refinit"))
        m.set_compute_dtype(dt)
        opt = train.HipAdam([p for p in m.parameters()], lr=1e-4, shadow_of=m)
        if how == "graph":
            g = train.GraphedTrainStep(m, opt, (inp, inp_s, tgt, tgt_s), stats, maps, const_h)
            losses[how] = [float(g.step()) for _ in range(3)]
            if dt == BF:
                with torch.no_grad():
                    a = m(inp, inp_s, stats, maps, const_h)[0].clone()
                    m.invalidate_shadows()
                    b = m(inp, inp_s, stats, maps, const_h)[0]
                assert torch.equal(a, b)
        else:
            losses[how] = [float(train.train_step(m, opt, (inp, inp_s, tgt, tgt_s), stats, maps, const_h)) for _ in range(3)]
        del m, opt
        torch.cuda.empty_cache()
    print(dt, losses)
    assert losses["eager"][2] < losses["eager"][0]                      # the steps really optimise
    for a, b in zip(losses["eager"], losses["graph"]):
        assert abs(a - b) < (2e-5 if dt == torch.float32 else 2e-3) * abs(a)
    m = P.PanguModel(device="cuda").cuda().train()
    with pytest.raises(RuntimeError, match="stochastic depth"):
        train.GraphedTrainStep(m, train.make_optimizer(m), (inp, inp_s, tgt, tgt_s), stats, maps, const_h)

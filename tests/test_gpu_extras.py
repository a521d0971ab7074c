"""SURVEY.md 8(f) widenings on the GPU: device-side latitude-weighted scores and the double-buffered input pipeline."""
import os

import numpy as np
import pytest
import torch

import cases
import pangu_oracle as O
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pangu_pytorch_amd as P
    P._lib.load()
    return P


def test_scores_match_reference_golden(P, golden_dir):
    g = np.load(os.path.join(golden_dir, "extras.npz"))
    pred = synth.uniform((2, 5, 721, 1440), synth.name_seed("score_pred"), device="cuda")
    tgt = pred * 0.7 + 0.5 * synth.uniform((2, 5, 721, 1440), synth.name_seed("score_tgt"), device="cuda")
    rm, ac = P.score.weighted_rmse_channels(pred, tgt), P.score.weighted_acc_channels(pred, tgt)
    assert rm.shape == (2, 5) and np.allclose(rm.cpu().numpy(), g["rmse"], rtol=2e-5)
    assert np.allclose(ac.cpu().numpy(), g["acc"], rtol=2e-5)
    assert np.allclose(P.score.weighted_rmse(pred, tgt).cpu().numpy(), g["rmse_mean"], rtol=2e-5)
    assert np.allclose(P.score.weighted_acc(pred, tgt).cpu().numpy(), g["acc_mean"], rtol=2e-5)
    # 5-D upper-air fields and ragged shapes against the oracle
    p5 = synth.uniform((1, 2, 3, 37, 24), 5, device="cuda")
    t5 = synth.uniform((1, 2, 3, 37, 24), 6, device="cuda")
    assert torch.allclose(P.score.weighted_rmse_channels(p5, t5).cpu(), O.weighted_rmse_channels(p5.cpu(), t5.cpu()), rtol=2e-5)
    assert torch.allclose(P.score.weighted_acc_channels(p5, t5).cpu(), O.weighted_acc_channels(p5.cpu(), t5.cpu()), rtol=2e-4, atol=1e-6)


def test_device_prefetcher(P):
    batches = [(torch.full((1, 5, 13, 8, 16), float(i)) + torch.arange(13.0).view(1, 1, 13, 1, 1),
                torch.full((1, 4, 8, 16), float(i)), torch.full((1, 5, 13, 8, 16), -float(i)), torch.zeros(1, 4, 8, 16), i)
               for i in range(5)]
    got = list(P.data.DevicePrefetcher(batches, "cuda", flip_levels=True))
    assert len(got) == 5
    for i, (a, b, c, d, tag) in enumerate(got):
        assert a.is_cuda and tag == i
        assert torch.equal(a.cpu(), batches[i][0].flip(-3)) and torch.equal(b.cpu(), batches[i][1])
        assert torch.equal(c.cpu(), batches[i][2].flip(-3))
    plain = list(P.data.DevicePrefetcher(batches, "cuda"))
    assert torch.equal(plain[3][0].cpu(), batches[3][0])


def test_onnx_route_and_checkpoint_forward_golden(golden_dir, tmp_path):
    """Weight import end to end on the GPU: initialisers in the ONNX layout through the 223-row key table
    (weights.load_onnx_initializers, reference models/onnx2torch.py:23-52), then saved / re-loaded as the
    {'model': state_dict} checkpoint of reference finetune_fully.py:115-116 (weights.load_checkpoint), then the forward
    == the reference's forward on those weights (tests/golden/model_fwd.npz) within 1e-3."""
    import json
    import os
    import numpy as np
    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import weights as Wt
    table = json.load(open(os.path.join(golden_dir, "keys_table.json")))
    want = synth.fill_state_dict(cases.model_param_shapes())
    onnx_weights = {table[k]: (v.t().contiguous() if v.dim() == 2 else v).numpy() for k, v in want.items()}
    m = P.PanguModel(device="cuda").cuda().eval()
    assert Wt.load_onnx_initializers(m, onnx_weights, table) == []
    path = tmp_path / "onnx2torch_ckpt.pth"
    torch.save({"model": m.state_dict(), "epoch": 0}, path)
    m2 = P.PanguModel(device="cuda").cuda().eval()
    Wt.load_checkpoint(m2, str(path), map_location="cuda")
    g = np.load(os.path.join(golden_dir, "model_fwd.npz"))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        out, out_s = m2(inp, inp_s, stats, maps, const_h)
    assert cases.compare_summary(out, g, "model.out", 1e-3) < 1e-3
    assert cases.compare_summary(out_s, g, "model.out_surface", 1e-3) < 1e-3


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_compact_bias_bit_exact(P, C, shifted):
    """Inference on the paper's compact bias table (reference layers.py:306-357, :384-391): the kernel's
    closed-form position index + gather gives the SAME BITS as the expanded (types, heads, 144, 144) table of the same
    values, and the table matches the oracle's attention through the expanded tensor."""
    from pangu_pytorch_amd import weights
    st = cases.STAGES[C]
    Z, H, W, heads, types = st["Z"], st["H"], 24, st["heads"], st["types"]
    N = Z * H * W
    compact = synth.uniform((3312, types, heads), 91, 0.5)
    esb = weights.expand_bias(compact)                                  # (1, types, heads, 144, 144)
    qkv = synth.uniform((1, N, 3 * C), 92, 1.5)
    b1 = synth.uniform((3 * C,), 93, 0.5)
    table = weights.compact_bias_table(esb)
    assert table is not None and torch.equal(table, compact.permute(1, 2, 0))
    o_e, l_e = P.ops.window_attention(qkv[0].cuda(), b1.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    o_c, l_c = P.ops.window_attention(qkv[0].cuda(), b1.cuda(), table.cuda(), Z, H, W, heads, shifted, want_lse=True, compact=True)
    assert torch.equal(o_e, o_c) and torch.equal(l_e, l_c)
    ref, _ = O.window_attention_core(qkv, b1, esb, Z, H, W, heads, shifted)
    assert ((o_c.cpu() - ref[0]).abs().max() / ref.abs().max()).item() < 1e-5


def test_model_compact_bias_mode(P):
    """PanguModel.use_compact_bias: refuses the reference's random expanded initialisation; with expansions of compact
    tables in every block the fp32 forward is bit-identical to the expanded path, survives an in-place weight edit (the
    tables are rebuilt from the parameter's version stamp), and training still runs on the expanded parameter."""
    from pangu_pytorch_amd import weights
    torch.manual_seed(3)
    m = P.PanguModel(device="cuda").cuda().eval()
    with pytest.raises(ValueError):
        m.use_compact_bias(True)
    m.use_compact_bias(False)
    g = torch.Generator(device="cuda").manual_seed(11)
    with torch.no_grad():
        for mod in m.modules():
            p = getattr(mod, "earth_specific_bias", None)
            if p is not None:
                c = torch.randn(3312, p.shape[1], p.shape[2], generator=g, device="cuda") * 0.02
                p.copy_(weights.expand_bias(c))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        ref, ref_s = m(inp, inp_s, stats, maps, const_h)
        m.use_compact_bias(True)
        out, out_s = m(inp, inp_s, stats, maps, const_h)
        assert torch.equal(out, ref) and torch.equal(out_s, ref_s)
        blk = m.layers[1].blocks[0].attention
        assert blk._esb_compact is not None and blk._esb_compact.shape == (64, 12, 3312)
        blk.earth_specific_bias.mul_(2.0)                             # still an expansion; the stamp changes
        out2, _ = m(inp, inp_s, stats, maps, const_h)
        m.use_compact_bias(False)
        ref2, _ = m(inp, inp_s, stats, maps, const_h)
        assert torch.equal(out2, ref2) and not torch.equal(out2, out)


@pytest.mark.parametrize("shape", [(1, 13, 721, 1440), (2, 3, 37, 24), (1, 2, 5, 7)])
def test_weighted_l1_loss_hip_vs_oracle(P, shape):
    """The two-pass HIP loss (csrc/loss.hip) against the oracle's restatement of reference pangu_sample.py:61-67 on the CPU
    (value, fp64 accumulation: 1e-6 relative) and against torch autograd of the same expression on the device (the gradient is
    sign(o - t) times a per-variable constant built in autograd's multiplication order: bit-identical).  Shapes: the model's
    fields, a per-GPU batch of 2, and planes whose length is not a multiple of 4 (scalar path)."""
    from pangu_pytorch_amd import train
    B, L, H, W = shape
    o = synth.uniform((B, 5, L, H, W), synth.name_seed("loss_o"), device="cuda").requires_grad_(True)
    t = synth.uniform((B, 5, L, H, W), synth.name_seed("loss_t"), device="cuda")
    os_ = synth.uniform((B, 4, H, W), synth.name_seed("loss_os"), device="cuda").requires_grad_(True)
    ts = synth.uniform((B, 4, H, W), synth.name_seed("loss_ts"), device="cuda")
    with torch.no_grad():
        t[0, 1, 0, 0, :3] = o[0, 1, 0, 0, :3]                 # exact ties: sign(0) = 0
    assert train._hip_loss_ok(o, os_, t, ts)
    loss = train.weighted_l1_loss(o, os_, t, ts)
    (loss * 1.7).backward()
    ref = O.train_loss(o.detach().cpu().double(), os_.detach().cpu().double(), t.cpu().double(), ts.cpu().double())
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    o2, os2 = o.detach().clone().requires_grad_(True), os_.detach().clone().requires_grad_(True)
    (train._weighted_l1_loss_torch(o2, os2, t, ts) * 1.7).backward()
    assert torch.equal(o.grad, o2.grad) and torch.equal(os_.grad, os2.grad)
    assert float(o.grad[0, 1, 0, 0, :3].abs().max()) == 0.0


def test_hip_adam_matches_torch_fused_adam(P):
    """train.HipAdam (csrc/adam.hip, one launch for every tensor) against torch.optim.Adam(fused=True) -- the optimiser of reference
    finetune_fully.py:121 -- over four steps with weight decay, tensors of ragged sizes (the 4-element vector path and its scalar
    tail), a tensor that joins late (different step counts: per-row bias corrections) and a non-default group.  The kernel follows
    ATen's adam_math operation by operation, so parameters and both moments must agree to the last bit."""
    from pangu_pytorch_amd import train
    torch.manual_seed(0)
    shapes = [(1037,), (64, 96), (3, 5, 7), (4096 * 3 + 1,), (2,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    kw = dict(lr=3e-3, weight_decay=3e-2, betas=(0.9, 0.98), eps=1e-8)
    oa = train.HipAdam([{"params": pa[:3]}, {"params": pa[3:], "lr": 1e-2, "weight_decay": 0.0}], **kw)
    ob_ = torch.optim.Adam([{"params": pb[:3]}, {"params": pb[3:], "lr": 1e-2, "weight_decay": 0.0}], fused=True, **kw)
    for step in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 1 and step == 0:
                a.grad = b.grad = None                    # joins at step 1: its step count stays one behind
                continue
            g = torch.randn_like(a) * (10.0 ** (i - 2))
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob_.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert torch.equal(a.detach(), b.detach()), (step, i, float((a - b).abs().max()))
            if a in oa.state and oa.state[a]:
                assert torch.equal(oa.state[a]["exp_avg"], ob_.state[b]["exp_avg"]), (step, i)
                assert torch.equal(oa.state[a]["exp_avg_sq"], ob_.state[b]["exp_avg_sq"]), (step, i)
    assert oa.state[pa[1]]["step"] == 3 and oa.state[pa[0]]["step"] == 4
    # a checkpoint written by torch's Adam (reference pangu_sample.py:95 saves optimizer.state_dict()) resumes under HipAdam
    oc = train.HipAdam([{"params": pa[:3]}, {"params": pa[3:], "lr": 1e-2, "weight_decay": 0.0}], **kw)
    import copy
    oc.load_state_dict(copy.deepcopy(ob_.state_dict()))      # as after torch.save / torch.load (state_dict() itself aliases the live tensors)
    for a, b in zip(pa, pb):
        g = torch.randn_like(a)
        a.grad, b.grad = g.clone(), g.clone()
    oc.step()
    ob_.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a.detach(), b.detach()), i


def test_hip_adam_missing_gradient_as_zero(P):
    """train_step's DropPath rule (a dropped branch's parameters get ZERO gradients, as in the reference where the branch is
    computed and multiplied by zero): HipAdam.step(missing_as_zero=True) with no gradient tensor at all == torch's fused Adam fed
    explicit zeros, bit for bit (moments decay, weight decay still applies)."""
    from pangu_pytorch_amd import train
    torch.manual_seed(1)
    pa = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (5000, 777)]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    kw = dict(lr=1e-2, weight_decay=1e-2)
    oa, ob_ = train.HipAdam(pa, **kw), torch.optim.Adam(pb, fused=True, **kw)
    for step in range(3):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 1 and step == 1:
                a.grad, b.grad = None, torch.zeros_like(b)
            else:
                g = torch.randn_like(a)
                a.grad, b.grad = g.clone(), g.clone()
        oa.step(missing_as_zero=True)
        ob_.step()
        for a, b in zip(pa, pb):
            assert torch.equal(a.detach(), b.detach())
            assert torch.equal(oa.state[a]["exp_avg_sq"], ob_.state[b]["exp_avg_sq"])

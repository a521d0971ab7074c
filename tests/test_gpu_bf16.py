"""bf16 path (BASELINE configs[2]/[4]): kernels vs fp32 references computed from the SAME bf16-rounded operands
(tolerance = bf16 output rounding, 2^-8), and whole-model drift vs the fp32 reference goldens (reported, bounded)."""
import os

import numpy as np
import pytest
import torch

import cases
import pangu_oracle as O
import synth

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ROUND = 1.0 / 128        # bf16 has 8 significant bits: one output rounding <= 2^-8 relative, x2 margin
# bf16 whole-model gradients vs the REFERENCE's fp32 autograd, worst of the 223 tensors, measured on MI355X with the goldens'
# O(1)-activation synthetic weights (not contractive: 16 blocks of bf16 forward drift feed the backward): gradient mass
# 2.2e-2 (median 2.1e-3), rel-L2 over a tensor's 256 samples 0.25-0.41 for the WORST tensor (a deep Earth-specific bias table;
# median 0.11 every time).  The worst tensor's figure is chaotic: swapping the forward resampling LayerNorm kernels for an
# equally accurate implementation (1.659e-3 vs 1.659e-3 rel-L2 against fp64 on random rows; PANGU_RESAMPLE_FAST=2 vs 3) moves it
# from 0.25 to 0.41 with the median unchanged -- a few flipped bf16 roundings, amplified by the non-contractive weights.  The
# bound below covers that spread; the meaningful bf16 bound is the reference-init one (test_..._refinit_vs_reference).
BF16_GRAD_SAMPLE_TOL = 0.5
BF16_GRAD_MASS_TOL = 3e-2


@pytest.fixture(scope="module")
def P():
    import pangu_pytorch_amd as P
    assert torch.cuda.is_available()
    P._lib.load()
    return P


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K,act,bias", [
    (1000, 192, 192, 0, True), (4099, 576, 192, 0, True), (2048, 768, 192, 1, True), (777, 192, 768, 0, True),
    (1531, 1152, 384, 0, True), (513, 1536, 384, 1, True), (300, 384, 1536, 0, True), (999, 384, 768, 0, False),
    (1234, 160, 384, 0, True), (321, 64, 384, 0, True), (650, 192, 128, 0, True), (128, 128, 64, 0, False),
    (700, 384, 160, 0, False), (500, 192, 576, 0, False),
])
@pytest.mark.parametrize("out_f32", [False, True])
def test_linear_bf16(P, M, N, K, act, bias, out_f32):
    from pangu_pytorch_amd import ops_bf16 as ob
    a = synth.uniform((M, K), 11).to(BF)
    w = synth.uniform((N, K), 12, 1.0 / K ** 0.5).to(BF)
    b = synth.uniform((N,), 13, 0.5) if bias else None
    ref = a.float() @ w.float().t()
    if bias:
        ref = ref + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    got = ob.linear(a.cuda(), w.cuda(), b.cuda() if bias else None, act=act,
                    out_dtype=torch.float32 if out_f32 else BF)
    assert got.dtype == (torch.float32 if out_f32 else BF)
    assert rel_err(got, ref) < (2e-4 if out_f32 else ROUND)


@pytest.mark.parametrize("M,N,K,bias", [(4099, 192, 768, False), (1531, 384, 1152, True), (5000, 192, 192, False)])
def test_linear_bf16_add_epilogue(P, M, N, K, bias):
    """act=ADD: out = a @ w^T + bias + aux (bf16 addend), the residual-gradient accumulation of the block backward."""
    from pangu_pytorch_amd import ops_bf16 as ob
    a = synth.uniform((M, K), 14).to(BF)
    w = synth.uniform((N, K), 15, 1.0 / K ** 0.5).to(BF)
    b = synth.uniform((N,), 16, 0.5) if bias else None
    add = synth.uniform((M, N), 17, 2.0).to(BF)
    ref = a.float() @ w.float().t() + add.float() + (b if bias else 0.0)
    got = ob.linear(a.cuda(), w.cuda(), b.cuda() if bias else None, act=ob.ACT_ADD, aux=add.cuda())
    assert got.dtype == BF and rel_err(got, ref) < ROUND


@pytest.mark.parametrize("M,N,K", [(1000, 192, 192), (4099, 192, 768), (1531, 384, 384), (777, 384, 1536), (128, 384, 64),
                                   (66001, 384, 384), (98400, 384, 96), (132070, 192, 192), (70000, 192, 128)])
@pytest.mark.parametrize("strided_out", [False, True])
def test_linear_ln_residual_bf16(P, M, N, K, strided_out):
    """Fused projection + post-norm residual == linear -> LayerNorm -> + shortcut (reference layers.py:250-251).  The large M: more
    128-row tiles than the persistent LDS-DMA kernel has workgroups (256 at N = 384, 512 at N = 192), so workgroups walk 2-4 tiles
    with the operand ring running across tile boundaries, ragged last tile included; K = 96 / 128: three / four K-steps (the ring's
    minimum); K = 64 takes the register-staged kernel."""
    from pangu_pytorch_amd import ops_bf16 as ob
    a = synth.uniform((M, K), 71).to(BF)
    w = synth.uniform((N, K), 72, 1.0 / K ** 0.5).to(BF)
    b = synth.uniform((N,), 73, 0.5)
    sc = synth.uniform((M, N), 74, 1.5).to(BF)
    g, be = synth.uniform((N,), 75, 0.5, 1.0), synth.uniform((N,), 76, 0.3)
    y = a.float() @ w.float().t() + b
    ref = sc.float() + torch.nn.functional.layer_norm(y, (N,), g, be, 1e-5)
    out = None
    if strided_out:
        full = torch.zeros((M, 2 * N), dtype=BF, device="cuda")
        out = full[:, N:]
    got = ob.linear_ln_residual(a.cuda(), w.cuda(), b.cuda(), sc.cuda(), g.cuda(), be.cuda(), out=out)
    assert got.dtype == BF and rel_err(got, ref) < ROUND
    if strided_out:
        assert float(full[:, :N].float().abs().max()) == 0.0


def _mlp_ref(x, w1, b1, w2, b2, g, be, scale):
    """fp64 reference of layers.py:251 + :264-270 on the bf16-rounded operands; the hidden activation is rounded to bf16
    like the kernel's second-product operand."""
    xd = x.double()
    h = torch.nn.functional.gelu(xd @ w1.double().t() + b1.double()).to(BF).double()
    y = h @ w2.double().t() + b2.double()
    return xd + scale * torch.nn.functional.layer_norm(y, (x.shape[1],), g.double(), be.double(), 1e-5)


@pytest.mark.parametrize("C,M", [(192, 256), (192, 1000), (192, 4099), (384, 128), (384, 777), (384, 2600)])
@pytest.mark.parametrize("strided_out,scale", [(False, 1.0), (True, 1.0), (False, 1.25)])
def test_mlp_ln_residual_fused_bf16(P, C, M, strided_out, scale):
    """One-launch MLP branch (csrc/mlp_fused_bf16.hip) == Linear -> GELU(erf) -> Linear -> LayerNorm -> residual
    (reference layers.py:251, :264-270), ragged M, strided output (the skip-concat buffer), DropPath branch scale."""
    from pangu_pytorch_amd import ops_bf16 as ob
    x = synth.uniform((M, C), 81, 1.5).to(BF)
    w1 = synth.uniform((4 * C, C), 82, 1.5 / C ** 0.5).to(BF)
    w2 = synth.uniform((C, 4 * C), 83, 1.0 / (2 * C ** 0.5)).to(BF)
    b1, b2 = synth.uniform((4 * C,), 84, 0.5), synth.uniform((C,), 85, 0.5)
    g, be = synth.uniform((C,), 86, 0.5, 1.0), synth.uniform((C,), 87, 0.3)
    ref = _mlp_ref(x, w1, b1, w2, b2, g, be, scale)
    img = ob.pack_mlp_weights(w1.cuda(), w2.cuda())
    out = None
    if strided_out:
        full = torch.zeros((M, 2 * C), dtype=BF, device="cuda")
        out = full[:, C:]
    got = ob.mlp_ln_residual(x.cuda(), img, b1.cuda(), b2.cuda(), g.cuda(), be.cuda(), out=out, branch_scale=scale)
    assert got.dtype == BF and got.shape == (M, C)
    err = (got.double().cpu() - ref).abs()
    # bf16 output rounding (2^-9 of |out| <= ~6) + the GELU fit (2.7e-4 per hidden unit, averaged down by the second product)
    assert err.max().item() < 2.5 * ROUND * ref.abs().max().item() / 2, err.max().item()
    assert (err.norm() / ref.norm()).item() < 3e-3
    if strided_out:
        assert float(full[:, :C].float().abs().max()) == 0.0
    # and the three-launch path it replaces agrees at the same level
    h = ob.linear(x.cuda(), w1.cuda(), b1.cuda(), act=ob.ACT_GELU)
    sep = ob.linear_ln_residual(h, w2.cuda(), b2.cuda(), x.cuda(), g.cuda(), be.cuda()) if scale == 1.0 else None
    if sep is not None:
        assert ((sep.double().cpu() - ref).norm() / ref.norm()).item() < 3e-3


@pytest.mark.parametrize("C,M", [(192, 256), (192, 4099), (384, 128), (384, 2600)])
@pytest.mark.parametrize("scale", [1.0, 1.25])
def test_mlp_ln_residual_train_fused_bf16(P, C, M, scale):
    """Training forward of the one-launch MLP branch: the result is the inference kernel's bit for bit; the side outputs are
    pre = x W1^T + b1 (before GELU) and m = GELU(pre) W2^T + b2 (before LayerNorm) -- reference layers.py:264-270 -- ragged M."""
    from pangu_pytorch_amd import ops_bf16 as ob
    x = synth.uniform((M, C), 81, 1.5).to(BF)
    w1 = synth.uniform((4 * C, C), 82, 1.5 / C ** 0.5).to(BF)
    w2 = synth.uniform((C, 4 * C), 83, 1.0 / (2 * C ** 0.5)).to(BF)
    b1, b2 = synth.uniform((4 * C,), 84, 0.5), synth.uniform((C,), 85, 0.5)
    g, be = synth.uniform((C,), 86, 0.5, 1.0), synth.uniform((C,), 87, 0.3)
    img = ob.pack_mlp_weights(w1.cuda(), w2.cuda())
    args = (x.cuda(), img, b1.cuda(), b2.cuda(), g.cuda(), be.cuda())
    want = ob.mlp_ln_residual(*args, branch_scale=scale)
    out, pre, m = ob.mlp_ln_residual_train(*args, branch_scale=scale)
    assert torch.equal(out, want)
    ref_pre = x.double() @ w1.double().t() + b1.double()
    ref_m = torch.nn.functional.gelu(ref_pre) @ w2.double().t() + b2.double()
    assert pre.shape == (M, 4 * C) and pre.dtype == BF
    e = (pre.double().cpu() - ref_pre).abs()
    assert e.max().item() < ROUND * ref_pre.abs().max().item() and rel_err(pre, ref_pre.float()) < ROUND
    assert m.shape == (M, C) and m.dtype == BF
    assert rel_err(m, ref_m.float()) < ROUND and (m.double().cpu() - ref_m).abs().max().item() < 2.5 * ROUND * ref_m.abs().max().item()


@pytest.mark.parametrize("M,C", [(1500, 192), (4099, 192), (5000, 384), (300, 384)])
def test_linear_gelu_bwd_with_hidden_bf16(P, M, C):
    """Backward through linear2 + GELU with h = GELU(pre) re-created in the same launch (weights-stationary kernel at
    K = 192, LDS-DMA ring at K = 384): dpre is the plain GELU_BWD epilogue's bit for bit, h == gelu(pre) to bf16 rounding."""
    from pangu_pytorch_amd import ops_bf16 as ob
    N = 4 * C
    pre = synth.uniform((M, N), 70, 2.5).to(BF).cuda()
    dm, w2t = synth.uniform((M, C), 68).to(BF).cuda(), synth.uniform((N, C), 69, 0.05).to(BF).cuda()
    want = ob.linear(dm, w2t, None, act=ob.ACT_GELU_BWD, aux=pre)
    dpre, h = ob.linear_gelu_bwd(dm, w2t, pre)
    assert torch.equal(dpre, want)
    ref_h = torch.nn.functional.gelu(pre.double().cpu())
    assert (h.double().cpu() - ref_h).abs().max().item() < ROUND * ref_h.abs().max().item()
    assert rel_err(h, ref_h.float()) < ROUND
    x = pre.float().cpu().requires_grad_(True)
    (torch.nn.functional.gelu(x) * (dm.float().cpu() @ w2t.float().cpu().t())).sum().backward()
    assert rel_err(dpre, x.grad) < ROUND
    d2, h2 = ob.linear_gelu_bwd(dm, w2t, pre, want_h=False)
    assert h2 is None and torch.equal(d2, want)


def test_linear_bf16_random_shapes(P):
    """Ragged M, K % 8 == 0, every kernel family (weights-stationary, LDS-DMA ring, register-staged), bias / GELU / add."""
    import random
    from pangu_pytorch_amd import ops_bf16 as ob
    rnd = random.Random(11)
    torch.manual_seed(11)
    for _ in range(40):
        N = rnd.choice([160, 192, 384, 576, 768, 1152, 1536, 136, 176, 64])
        K = 8 * rnd.randint(1, 200)
        M = rnd.randint(1, 6000)
        act = rnd.choice([0, 0, 1, 3])
        a = torch.randn(M, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF)
        b = torch.randn(N, device="cuda") if rnd.random() < 0.7 else None
        aux = torch.randn(M, N, device="cuda").to(BF) if act == 3 else None
        ref = a.double() @ w.double().t()
        if b is not None:
            ref = ref + b.double()
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        if act == 3:
            ref = ref + aux.double()
        got = ob.linear(a, w, b, act=act, aux=aux)
        err = ((got.double() - ref).norm() / ref.norm()).item()
        assert err < ROUND, (M, N, K, act, err)


def test_linear_bf16_gelu_aux_and_bwd(P):
    from pangu_pytorch_amd import ops_bf16 as ob
    M, N, K = 1500, 768, 192
    a, w, b = synth.uniform((M, K), 65).to(BF), synth.uniform((N, K), 66, 0.1).to(BF), synth.uniform((N,), 67, 0.3)
    pre = torch.empty((M, N), device="cuda", dtype=BF)
    h = ob.linear(a.cuda(), w.cuda(), b.cuda(), act=ob.ACT_GELU, aux=pre)
    ref_pre = a.float() @ w.float().t() + b
    assert rel_err(pre, ref_pre) < ROUND and rel_err(h, torch.nn.functional.gelu(ref_pre)) < ROUND
    dm, w2t = synth.uniform((M, 192), 68).to(BF), synth.uniform((N, 192), 69, 0.05).to(BF)
    got = ob.linear(dm.cuda(), w2t.cuda(), None, act=ob.ACT_GELU_BWD, aux=pre)
    x = pre.float().cpu().requires_grad_(True)
    (torch.nn.functional.gelu(x) * (dm.float() @ w2t.float().t())).sum().backward()
    assert rel_err(got, x.grad) < ROUND


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_bf16(P, C, shifted):
    from pangu_pytorch_amd import ops_bf16 as ob
    st = cases.STAGES[C]
    Z, H, W, heads = st["Z"], st["H"], 24, st["heads"]
    N = Z * H * W
    qkv = synth.uniform((1, N, 3 * C), 31, 1.5).to(BF)
    b1 = synth.uniform((3 * C,), 32, 0.5).to(BF)
    esb = synth.uniform((1, st["types"], heads, 144, 144), 33, 0.5).to(BF)
    ref, ref_lse = O.window_attention_core(qkv.float(), b1.float(), esb.float(), Z, H, W, heads, shifted)
    got, lse = ob.window_attention(qkv[0].cuda(), b1.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    # P is rounded to bf16 before P.V (as every bf16 flash kernel does): 2^-9 per term, averaged by the sum
    assert rel_err(got, ref[0]) < ROUND
    assert rel_err(lse, ref_lse[0]) < 1e-4


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_qkv_fused_bf16(P, C, shifted):
    """QKV projection fused into the attention kernel (csrc/attn_bf16.hip window_attn_qkv_bf16_kernel) == Linear ->
    window attention (reference layers.py:365-415) on the same bf16-rounded operands: q, k, v are rounded to bf16 once,
    exactly where the two-launch path rounds them; padded rows take part with q = k = v = bias (layers.py:192)."""
    from pangu_pytorch_amd import ops_bf16 as ob
    st = cases.STAGES[C]
    Z, H, W, heads = st["Z"], st["H"], 24, st["heads"]
    N = Z * H * W
    x = synth.uniform((N, C), 35, 1.5).to(BF)
    w = synth.uniform((3 * C, C), 36, 1.5 / C ** 0.5).to(BF)
    b = synth.uniform((3 * C,), 37, 0.5)
    esb = synth.uniform((1, st["types"], heads, 144, 144), 38, 0.5).to(BF)
    qkv = (x.double() @ w.double().t() + b.double()).to(BF)                      # the rounding point of the unfused path
    ref, ref_lse = O.window_attention_core(qkv.float()[None], b.to(BF).float(), esb.float(), Z, H, W, heads, shifted)
    # the oracle takes the pad rows' q/k/v from the (bf16-rounded) bias, the kernel computes them as 0 @ W + b -> same value
    got, lse = ob.window_attention_qkv(x.cuda(), w.cuda(), b.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    assert rel_err(got, ref[0]) < ROUND
    assert rel_err(lse, ref_lse[0]) < 2e-3                 # scores from bf16 q, k: same inputs, fp32 accumulation order differs
    # and the two-launch path on the GPU agrees
    two = ob.window_attention(ob.linear(x.cuda(), w.cuda(), b.cuda()), b.to(BF).cuda(), esb[0].cuda(), Z, H, W, heads, shifted)
    assert rel_err(got, two) < ROUND


@pytest.mark.parametrize("C", [192, 384])
def test_ln_residual_bf16(P, C):
    from pangu_pytorch_amd import ops_bf16 as ob
    N = 1003
    y, sc = synth.uniform((N, C), 41, 2.0, 0.3).to(BF), synth.uniform((N, C), 42).to(BF)
    g, b = synth.uniform((C,), 43, 0.1, 1.0), synth.uniform((C,), 44, 0.1)
    ref = sc.float() + torch.nn.functional.layer_norm(y.float(), (C,), g, b)
    got = ob.ln_residual(y.cuda(), sc.cuda(), g.cuda(), b.cuda())
    assert rel_err(got, ref) < ROUND


@pytest.mark.parametrize("arm,C,Co", [("fast", 192, 192), ("generic", 40, 36)])
def test_resample_ln_backward_bf16(P, arm, C, Co):
    """Backward of the down- / up-sampling LayerNorms (reference layers.py:441-454, :480-495) in bf16 against torch autograd of the
    fp32 expression on the same bf16-rounded inputs: the 16-B fast kernels at the model's width, and the generic
    one-row-per-wave kernels that any other width (not a multiple of 16 / 8 channels) dispatches to."""
    from pangu_pytorch_amd import ops_bf16 as ob
    Z, H, W = 8, 181, 24
    H2, W2 = 91, 12
    x = synth.uniform((Z * H * W, C), 61).to(BF)
    g = synth.uniform((4 * C,), 62, 0.1, 1.0)
    dout = synth.uniform((Z * H2 * W2, 4 * C), 63).to(BF)
    xf = x.float().requires_grad_(True)
    gf = g.clone().requires_grad_(True)
    bf_ = torch.zeros(4 * C, requires_grad=True)
    xr = torch.nn.functional.pad(xf.view(Z, H, W, C), (0, 0, 0, 0, 0, 1)).view(Z, H2, 2, W2, 2, C)
    out = torch.nn.functional.layer_norm(xr.permute(0, 1, 3, 2, 4, 5).reshape(-1, 4 * C), (4 * C,), gf, bf_)
    out.backward(dout.float())
    dx, dg, db = ob.downsample_ln_bwd(dout.cuda(), x.cuda(), g.cuda(), Z, H, W)
    assert rel_err(dx, xf.grad) < ROUND and rel_err(dg, gf.grad) < 1e-3 and rel_err(db, bf_.grad) < 1e-3
    y = synth.uniform((Z * H2 * W2, 4 * Co), 64).to(BF)
    g = synth.uniform((Co,), 65, 0.1, 1.0)
    dout = synth.uniform((Z * H * 2 * W2, Co), 66).to(BF)
    yf = y.float().requires_grad_(True)
    gf = g.clone().requires_grad_(True)
    bf_ = torch.zeros(Co, requires_grad=True)
    yr = yf.view(Z, H2, W2, 2, 2, Co).permute(0, 1, 3, 2, 4, 5).reshape(Z, 2 * H2, 2 * W2, Co)[:, :H]
    out = torch.nn.functional.layer_norm(yr.reshape(-1, Co), (Co,), gf, bf_)
    out.backward(dout.float())
    dy, dg, db = ob.upsample_ln_bwd(dout.cuda(), y.cuda(), g.cuda(), Z, H2, W2, H)
    assert rel_err(dy, yf.grad) < ROUND and rel_err(dg, gf.grad) < 1e-3 and rel_err(db, bf_.grad) < 1e-3
    # the cropped fine rows (h = 181 of 182) receive exactly zero
    dyv = dy.float().view(Z, H2, W2, 2, 2, Co)
    assert float(dyv[:, H2 - 1, :, 1].abs().max()) == 0.0


def test_downsample_upsample_embed_bf16(P):
    from pangu_pytorch_amd import ops_bf16 as ob
    Z, H, W, C = 8, 181, 24, 192
    x = synth.uniform((Z * H * W, C), 51).to(BF)
    g, b = synth.uniform((4 * C,), 52, 0.1, 1.0), synth.uniform((4 * C,), 53, 0.1)
    xr = torch.nn.functional.pad(x.float().view(Z, H, W, C), (0, 0, 0, 0, 0, 1)).view(Z, 91, 2, 12, 2, C)
    ref = torch.nn.functional.layer_norm(xr.permute(0, 1, 3, 2, 4, 5).reshape(-1, 4 * C), (4 * C,), g, b)
    assert rel_err(ob.downsample_ln(x.cuda(), g.cuda(), b.cuda(), Z, H, W), ref) < ROUND
    H2, W2, Co = 91, 12, 192
    y = synth.uniform((Z * H2 * W2, 4 * Co), 54).to(BF)
    g, b = synth.uniform((Co,), 55, 0.1, 1.0), synth.uniform((Co,), 56, 0.1)
    yr = y.float().view(Z, H2, W2, 2, 2, Co).permute(0, 1, 3, 2, 4, 5).reshape(Z, 2 * H2, 2 * W2, Co)[:, :H]
    ref = torch.nn.functional.layer_norm(yr.reshape(-1, Co), (Co,), g, b)
    assert rel_err(ob.upsample_ln(y.cuda(), g.cuda(), b.cuda(), Z, H2, W2, H), ref) < ROUND
    LAT, LON, H4, W4 = 41, 280, 11, 70
    gg = lambda n, s, sc=1.0, sh=0.0: synth.uniform(s, synth.name_seed(n), sc, sh)
    inp, inp_s = gg("i", (1, 5, 13, LAT, LON)), gg("is", (1, 4, LAT, LON))
    stats = (gg("sm", (4,), 0.3), gg("ss", (4,), 0.2, 1.2), gg("um", (13, 1, 1, 5), 0.3), gg("us", (13, 1, 1, 5), 0.2, 1.2))
    maps, const_h = gg("m", (1, 3, 4 * H4, LON)), gg("c", (1, 1, 1, 13, LAT, LON))
    ra_s, ra_u = O.patch_embed_matrices(inp, inp_s, stats, maps, const_h)
    a_s, a_u = ob.patch_embed_gather(inp[0].cuda(), inp_s[0].cuda(), stats[0].cuda(), stats[1].cuda(),
                                     stats[2].reshape(13, 5).cuda(), stats[3].reshape(13, 5).cuda(), maps[0].cuda(),
                                     const_h.reshape(13, LAT, LON).cuda())
    assert a_s.shape == (H4 * W4, 128) and float(a_s[:, 112:].float().abs().max()) == 0.0
    assert torch.equal(a_s[:, :112].cpu(), ra_s[0].to(BF)) and torch.equal(a_u.cpu(), ra_u[0].to(BF))


def test_full_model_bf16_drift(P, golden_dir):
    """bf16 forward vs the fp32 reference golden: drift is reported and bounded (the reference's own bf16 autocast
    drifts 3.8e-3 rel-L2 PER BLOCK; 16 blocks)."""
    g = np.load(os.path.join(golden_dir, "model_fwd.npz"))
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        ref, ref_s = m(inp, inp_s, stats, maps, const_h)                       # fp32 HIP path (parity-tested)
        m.set_compute_dtype(BF)
        out, out_s = m(inp, inp_s, stats, maps, const_h)
        m.set_compute_dtype(torch.float32)
        with torch.autocast("cuda", dtype=BF):
            out2, _ = m(inp, inp_s, stats, maps, const_h)
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    assert torch.equal(out, out2)                                             # autocast selects the same path
    l2 = ((out - ref).double().norm() / ref.double().norm()).item()
    l2s = ((out_s - ref_s).double().norm() / ref_s.double().norm()).item()
    print(f"bf16 vs fp32 rel-L2 drift: upper {l2:.3e} surface {l2s:.3e}")
    assert l2 < 3e-2 and l2s < 3e-2                         # measured 0.7-1.5e-2 (O(1)-activation synthetic weights)
    assert cases.compare_summary(out, g, "model.out", 1.0) < 0.1     # worst fingerprint element vs the REFERENCE forward


# ------------------------------------------------------------------------------------------------ bf16 backward
@pytest.mark.parametrize("M,N,K", [(4096, 192, 192), (5000, 576, 192), (3001, 768, 192), (2500, 192, 768),
                                   (4111, 1152, 384), (2222, 384, 1536), (3333, 160, 384), (1999, 64, 384),
                                   (2777, 192, 128), (100, 384, 768)])
def test_linear_wgrad_bf16(P, M, N, K):
    from pangu_pytorch_amd import ops_bf16 as ob
    dc, a = synth.uniform((M, N), 61).to(BF), synth.uniform((M, K), 62).to(BF)
    dw, db = ob.linear_wgrad(dc.cuda(), a.cuda())
    assert dw.dtype == torch.float32
    assert rel_err(dw, dc.double().t() @ a.double()) < 2e-4          # exact bf16 products, fp32 accumulation
    assert rel_err(db, dc.double().sum(0)) < 2e-4


@pytest.mark.parametrize("M,N,K,strided", [(131040, 1152, 384, False), (131040 + 37, 384, 1536, True),
                                           (400000, 192, 768, False)])
def test_linear_wgrad_bf16_model_size(P, M, N, K, strided):
    """The large products (>= 1e11 FLOP) run the LDS-DMA kernel (wgrad_bf16_dma.hip: source-side swizzled slabs, bias
    gradient as an MFMA column); N = 192 leaves the second 128-column tile half empty, M is ragged against the 32-token
    K-step, the operands may be row-strided views.  Reference: fp64 on the GPU over the same bf16 values."""
    from pangu_pytorch_amd import ops_bf16 as ob
    assert 2.0 * M * N * K >= 1.0e11
    dc = synth.uniform((M, N + (64 if strided else 0)), 63, device="cuda").to(BF)[:, :N]
    a = synth.uniform((M, K + (128 if strided else 0)), 64, device="cuda").to(BF)[:, :K]
    dw, db = ob.linear_wgrad(dc, a)
    ref_w = torch.zeros((N, K), dtype=torch.float64, device="cuda")
    for m0 in range(0, M, 32768):                       # chunked: keeps the fp64 copies small
        ref_w += dc[m0:m0 + 32768].double().t() @ a[m0:m0 + 32768].double()
    assert rel_err(dw, ref_w) < 2e-4
    assert rel_err(db, dc.double().sum(0)) < 2e-4


@pytest.mark.parametrize("M,N,K", [(131040, 1536, 384), (300000, 192, 1152), (260000, 576, 384)])
def test_linear_wgrad_bf16_without_workspace(P, M, N, K):
    """pangu_linear_wgrad_bf16 (no scratch buffer): the LDS-DMA kernels end in fp32 atomics on dW -- the 384 x 192 and 192 x 384
    12-wave tiles and the 128 x 192 4-wave tile (N = 576: a half-empty last tile) -- and ADD into what dW / db already hold."""
    from pangu_pytorch_amd import _lib
    assert 2.0 * M * N * K >= 1.0e11
    lib = _lib.load()
    dc = synth.uniform((M, N), 65, device="cuda").to(BF)
    a = synth.uniform((M, K), 66, device="cuda").to(BF)
    dw = torch.full((N, K), 0.5, device="cuda")
    db = torch.full((N,), -0.25, device="cuda")
    _lib.check(lib.pangu_linear_wgrad_bf16(torch.cuda.current_stream().cuda_stream, dc.data_ptr(), N, a.data_ptr(), K,
                                           dw.data_ptr(), db.data_ptr(), M, N, K), "linear_wgrad_bf16")
    ref_w = torch.zeros((N, K), dtype=torch.float64, device="cuda")
    for m0 in range(0, M, 32768):
        ref_w += dc[m0:m0 + 32768].double().t() @ a[m0:m0 + 32768].double()
    assert rel_err(dw - 0.5, ref_w) < 2e-4
    assert rel_err(db + 0.25, dc.double().sum(0)) < 2e-4


@pytest.mark.parametrize("C", [192, 384])
def test_ln_residual_bwd_bf16(P, C):
    from pangu_pytorch_amd import ops_bf16 as ob
    N = 2051
    y = synth.uniform((N, C), 71, 2.0, 0.3).to(BF)
    dout = synth.uniform((N, C), 74).to(BF)
    yr = y.float().requires_grad_(True)
    g = synth.uniform((C,), 72, 0.1, 1.0).requires_grad_(True)
    b = synth.uniform((C,), 73, 0.1).requires_grad_(True)
    (1.25 * torch.nn.functional.layer_norm(yr, (C,), g, b) * dout.float()).sum().backward()
    dy, dg, db = ob.ln_residual_bwd(dout.cuda(), y.cuda(), g.detach().cuda(), 1.25)
    assert rel_err(dy, yr.grad) < ROUND and rel_err(dg, g.grad) < 1e-3 and rel_err(db, b.grad) < 1e-3


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_bwd_bf16(P, C, shifted):
    from pangu_pytorch_amd import ops_bf16 as ob
    st = cases.STAGES[C]
    Z, H, W, heads = st["Z"], st["H"], 24, st["heads"]
    N = Z * H * W
    qkv = synth.uniform((1, N, 3 * C), 81, 1.5).to(BF)
    b1 = synth.uniform((3 * C,), 82, 0.5).to(BF)
    esb = synth.uniform((1, st["types"], heads, 144, 144), 83, 0.5).to(BF)
    do = synth.uniform((1, N, C), 84).to(BF)
    q32, b32, e32 = qkv.float().requires_grad_(True), b1.float().requires_grad_(True), esb.float().requires_grad_(True)
    ref, _ = O.window_attention_core(q32, b32, e32, Z, H, W, heads, shifted)
    (ref * do.float()).sum().backward()
    o, lse = ob.window_attention(qkv[0].cuda(), b1.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    dqkv, dqb, desb = ob.window_attention_bwd(qkv[0].cuda(), b1.cuda(), esb[0].cuda(), o, lse, do[0].cuda(), Z, H, W,
                                              heads, shifted)
    # P and dS are rounded to bf16 before the second products, dqkv is stored as bf16: a few 2^-9 steps
    assert rel_err(dqkv, q32.grad[0]) < 2 * ROUND
    assert rel_err(desb, e32.grad[0]) < 2 * ROUND
    assert rel_err(dqb, b32.grad) < 2 * ROUND

def test_block_backward_bf16_vs_fp32(P, golden_dir):
    """One block fwd+bwd in bf16 vs the reference's fp32 gradients: rel-L2 per tensor reported, bounded at 2e-2."""
    C, roll = 192, True
    tag = f"block_{C}_{int(roll)}"
    st = cases.STAGES[C]
    from pangu_pytorch_amd import autograd_bf16 as AB, fused_bf16
    blk = P.layers.EarthSpecificBlock(C, 0.0, st["heads"], device="cuda").cuda().eval()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, s, "cuda") for k, s in cases.block_param_shapes(C).items()})
    x = cases.block_input(C, 24, "cuda")
    # fp32 HIP path = reference-accurate (tests/test_gpu_backward.py)
    x32 = x.clone().requires_grad_(True)
    y32 = blk(x32, st["Z"], st["H"], 24, roll)
    cot = cases.cotangent(tag, y32.shape, "cuda")
    (y32 * cot).sum().backward()
    ref = {k: p.grad.clone() for k, p in blk.named_parameters()}
    ref_dx = x32.grad.clone()
    blk.zero_grad()
    att = blk.attention
    sh = fused_bf16.WeightShadow()
    xb = x[0].to(BF).requires_grad_(True)
    yb = AB.EarthBlockFnBF16.apply(xb, blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias,
                                   blk.linear.linear1.weight, blk.linear.linear1.bias, blk.linear.linear2.weight,
                                   blk.linear.linear2.bias, att.earth_specific_bias, att.linear1.weight, att.linear1.bias,
                                   att.linear2.weight, att.linear2.bias, (st["Z"], st["H"], 24, st["heads"], roll), 1.0,
                                   1.0, sh)
    (yb.float() * cot[0]).sum().backward()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    worst = max((l2(p.grad, ref[k]), k) for k, p in blk.named_parameters())
    dxe = l2(xb.grad.float(), ref_dx[0])
    print(f"bf16 block backward rel-L2: dx {dxe:.3e}, worst param {worst[1]} {worst[0]:.3e}")
    assert dxe < 2e-2 and worst[0] < 2e-2


def test_full_training_step_bf16(P):
    """Whole bf16 training step runs, loss close to the fp32 step's, gradients finite and close in rel-L2."""
    from pangu_pytorch_amd import train
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    tgt, tgt_s = cases.model_targets("cuda")
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    l32 = train.weighted_l1_loss(out, out_s, tgt, tgt_s)
    l32.backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    del out, out_s
    m.set_compute_dtype(BF)
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    assert out.dtype == torch.float32
    lb = train.weighted_l1_loss(out, out_s, tgt, tgt_s)
    lb.backward()
    assert abs(lb.item() - l32.item()) / l32.item() < 2e-2
    errs = []
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
        errs.append((((p.grad.double() - ref[k].double()).norm() / ref[k].double().norm().clamp_min(1e-30)).item(), k))
    errs.sort(reverse=True)
    print("bf16 training step: loss fp32 %.6f bf16 %.6f; worst grad rel-L2:" % (l32.item(), lb.item()), errs[:3])
    med = errs[len(errs) // 2][0]
    assert med < 0.1, med


@pytest.mark.parametrize("optimizer", ["torch-fused", "hip"])
def test_bf16_shadows_follow_fused_adam(P, optimizer):
    """The bf16 weight shadows (and packed / transposed images) must track the fp32 master weights across optimizer steps.
    torch's Adam(fused=True) -- train.make_optimizer's form -- does not bump `_version`, so the stamp carries an optimizer epoch
    (ops.param_stamp): after a training step with a LARGE learning rate the forward must be bit-identical to a forward on
    freshly rebuilt shadows, the stored shadow must equal the updated weight's bf16 image, and the output must have moved."""
    from pangu_pytorch_amd import train
    m = P.PanguModel(device="cuda").cuda().eval()            # eval: DropPath off, the three forwards are comparable
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    m.set_compute_dtype(BF)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    tgt, tgt_s = cases.model_targets("cuda")
    opt = (torch.optim.Adam([p for p in m.parameters()], lr=1e-3, fused=True) if optimizer == "torch-fused" else
           train.HipAdam([p for p in m.parameters()], lr=1e-3, shadow_of=m))
    w = m.layers[0].blocks[0].attention.linear1.weight
    w0 = w.detach().clone()
    with torch.no_grad():
        before = m(inp, inp_s, stats, maps, const_h)[0].clone()
    for _ in range(2):
        train.train_step(m, opt, (inp, inp_s, tgt, tgt_s), stats, maps, const_h)
    assert not torch.equal(w.detach(), w0)
    with torch.no_grad():
        after = m(inp, inp_s, stats, maps, const_h)[0].clone()
        sh = m._shadow
        assert sh.table is not None and len(sh.table[1]) + len(sh.table[3]) > 100         # the one-launch refresh ran (pangu_shadow_refresh_bf16); [3] = images the optimizer wrote itself
        blk = m.layers[1].blocks[1]
        assert torch.equal(sh.get(w), w.detach().to(BF))
        esb = blk.attention.earth_specific_bias
        assert torch.equal(sh.get(esb), esb.detach()[0].to(BF))
        for p in (blk.attention.linear1.weight, blk.linear.linear2.weight):
            assert torch.equal(sh.get_t(p), p.detach().t().to(BF).contiguous())
        w1, w2 = blk.linear.linear1.weight, blk.linear.linear2.weight
        from pangu_pytorch_amd import ops_bf16 as ob
        assert torch.equal(sh.get_mlp(w1, w2), ob.pack_mlp_weights(w1.detach(), w2.detach()))
        from_adam = len(sh.table[3])
        m.invalidate_shadows()
        fresh = m(inp, inp_s, stats, maps, const_h)[0]
    assert torch.equal(after, fresh)
    assert not torch.equal(after, before)
    if optimizer == "hip":            # the Earth-specific bias images (and the other plain casts) came from the Adam launch itself
        assert from_adam >= 16


def test_bf16_training_trajectory_tracks_fp32(P):
    """Twelve optimisation steps (train.train_step: forward, reference loss, backward, HipAdam; DropPath on with the same host RNG
    seed, lr 2e-4 so that the weights really move) on one fixed batch, once in fp32 and once in bf16: the two loss trajectories stay
    within 1e-2 of each other at every step and fall by > 25 %.  The end-to-end check that the bf16 step trains on CURRENT weights:
    with stale weight shadows (round-3 bug) the bf16 losses stop following the fp32 ones after the first step."""
    import bench
    from pangu_pytorch_amd import train
    traj = {}
    for dt in (torch.float32, BF):
        torch.manual_seed(0)
        m = P.PanguModel(device="cuda").cuda().train()
        m.set_compute_dtype(dt)
        inp, inp_s, stats, maps, const_h = bench.synthetic_inputs(torch.device("cuda"), 1000)
        tgt, tgt_s, *_ = bench.synthetic_inputs(torch.device("cuda"), 2000)
        opt = train.HipAdam([p for p in m.parameters()], lr=2e-4, weight_decay=3e-6, shadow_of=m)
        torch.manual_seed(7)
        traj[dt] = [float(train.train_step(m, opt, (inp, inp_s, tgt, tgt_s), stats, maps, const_h)) for _ in range(12)]
        del m, opt
        torch.cuda.empty_cache()
    a, b = traj[torch.float32], traj[BF]
    print("fp32", ["%.4f" % v for v in a], "bf16", ["%.4f" % v for v in b])
    assert max(abs(x - y) for x, y in zip(a, b)) < 1e-2
    assert a[-1] < 0.75 * a[0] and b[-1] < 0.75 * b[0]


def test_full_backward_smooth_bf16_vs_reference(P, golden_dir):
    """bf16 whole-model forward + backward under the smooth loss sum(out * cotangent) / numel against the REFERENCE's fp32
    autograd (tests/golden/model_bwd_smooth.npz): every one of the 223 gradient tensors, WORST tensor bounded -- gradient
    mass (sum |g|) and the rel-L2 error over the 256 stored samples of each tensor."""
    g = np.load(os.path.join(golden_dir, "model_bwd_smooth.npz"))
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    m.set_compute_dtype(BF)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    out, out_s = m(inp, inp_s, stats, maps, const_h)
    loss = ((out * cases.cotangent("model_out", out.shape, "cuda")).sum() +
            (out_s * cases.cotangent("model_out_s", out_s.shape, "cuda")).sum()) / out.numel()
    loss.backward()
    l2s, masses = [], []
    for k, p in m.named_parameters():
        flat = p.grad.detach().float().flatten()
        assert torch.isfinite(flat).all(), k
        pos = synth.sample_positions(flat.numel(), cases.NSAMP, synth.name_seed("pos_model.d_" + k), device=flat.device)[:256]
        gs = torch.as_tensor(g[f"model.d_{k}.samples"]).double()
        gabs = float(g[f"model.d_{k}.abs_sum"][0])
        l2s.append((((flat[pos].cpu().double() - gs).norm() / gs.norm().clamp_min(1e-30)).item(), k))
        masses.append((abs(flat.double().abs().sum().item() - gabs) / gabs, k))
    l2s.sort(reverse=True)
    masses.sort(reverse=True)
    print("BF16GRAD worst sample rel-L2:", l2s[:4], "median", l2s[len(l2s) // 2][0])
    print("BF16GRAD worst mass:", masses[:4], "median", masses[len(masses) // 2][0])
    assert l2s[0][0] < BF16_GRAD_SAMPLE_TOL, l2s[0]
    assert masses[0][0] < BF16_GRAD_MASS_TOL, masses[0]
    # the wide bound above is for the one chaotic worst tensor only: the bulk is pinned tightly -- a regression in
    # one bias table's gradient moves these (measured: median 0.11, 90th percentile 0.19, mass median 2.1e-3)
    srt = sorted(v for v, _ in l2s)
    print("BF16GRAD sample rel-L2 median %.3f p90 %.3f; mass median %.2e" % (srt[len(srt) // 2], srt[int(0.9 * len(srt))],
                                                                            sorted(v for v, _ in masses)[len(masses) // 2]))
    assert srt[len(srt) // 2] < 0.15, srt[len(srt) // 2]
    assert srt[int(0.9 * len(srt))] < 0.30, srt[int(0.9 * len(srt))]
    assert sorted(v for v, _ in masses)[len(masses) // 2] < 5e-3


@pytest.mark.parametrize("C,roll", [(192, False), (192, True), (384, False), (384, True)])
def test_block_bf16_drift_within_2x_of_reference_autocast(P, golden_dir, C, roll):
    """in the REFERENCE's initialisation regime (weights std 0.02, LayerNorm (1, 0), zero biases:
    models/pangu_model.py:41-48; synth.param_spec_refinit) the HIP bf16 block -- training forward and no-grad forward -- drifts
    from fp32 by no more than 2x what the reference's own CPU autocast(bfloat16) block does on the same input
    (tests/golden/refinit.npz: 4.2e-3 rel-L2, recorded by oracle/gen_golden.py refinit)."""
    from pangu_pytorch_amd import autograd_bf16 as AB, fused_bf16
    g = np.load(os.path.join(golden_dir, "refinit.npz"))
    tag = f"refinit_block_{C}_{int(roll)}"
    st = cases.STAGES[C]
    blk = P.layers.EarthSpecificBlock(C, 0.0, st["heads"], device="cuda").cuda().eval()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, sh_, "cuda", spec="refinit") for k, sh_ in cases.block_param_shapes(C).items()})
    x = cases.block_input(C, 24, "cuda")
    with torch.no_grad():
        y32 = blk(x, st["Z"], st["H"], 24, roll)                        # fp32 HIP path
    # ... which IS the reference's fp32 block (golden fingerprint of the reference's output)
    pos = synth.sample_positions(y32.numel(), cases.NSAMP, synth.name_seed("pos_" + tag + ".out"), device="cuda")
    gs = torch.as_tensor(g[tag + ".out.samples"])
    assert ((y32.flatten()[pos].cpu() - gs).abs().max() / gs.abs().max()).item() < 1e-4
    ref_drift = float(g[tag + ".autocast_drift"][0])
    att, shd = blk.attention, fused_bf16.WeightShadow()
    xb = x[0].to(BF)
    yb_train = AB.EarthBlockFnBF16.apply(xb.clone().requires_grad_(True), blk.norm1.weight, blk.norm1.bias, blk.norm2.weight,
                                         blk.norm2.bias, blk.linear.linear1.weight, blk.linear.linear1.bias,
                                         blk.linear.linear2.weight, blk.linear.linear2.bias, att.earth_specific_bias,
                                         att.linear1.weight, att.linear1.bias, att.linear2.weight, att.linear2.bias,
                                         (st["Z"], st["H"], 24, st["heads"], roll), 1.0, 1.0, shd).detach()
    with torch.no_grad():
        yb_inf = fused_bf16._block(blk, shd, xb, st["Z"], st["H"], 24, roll)
    l2 = lambda a: ((a.double() - y32[0].double()).norm() / y32[0].double().norm()).item()
    d_train, d_inf = l2(yb_train.float()), l2(yb_inf.float())
    print(f"{tag}: bf16 drift vs fp32: training forward {d_train:.2e}, inference forward {d_inf:.2e}; reference autocast {ref_drift:.2e}")
    # measured on MI355X: 1.00-1.04x the reference's own autocast drift (4.2-4.4e-3 vs 4.2-4.3e-3); the bound asked for is 2x
    assert d_train < 1.25 * ref_drift and d_inf < 1.25 * ref_drift


def test_full_backward_smooth_bf16_refinit_vs_reference(P, golden_dir):
    """whole-model bf16 backward against the REFERENCE's fp32 autograd in the reference's initialisation
    regime (tests/golden/refinit.npz): the loss, the output, and every one of the 223 gradient tensors -- each bounded by 1.5x the
    error the REFERENCE'S OWN CPU autocast(bfloat16) makes on that tensor (tests/golden/autocast.npz), not by a hand-picked constant."""
    g = np.load(os.path.join(golden_dir, "refinit.npz"))
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda", spec="refinit"))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")

    def run(dt):
        m.set_compute_dtype(dt)
        m.zero_grad(set_to_none=True)
        out, out_s = m(inp, inp_s, stats, maps, const_h)
        loss = ((out * cases.cotangent("model_out", out.shape, "cuda")).sum() +
                (out_s * cases.cotangent("model_out_s", out_s.shape, "cuda")).sum()) / out.numel()
        loss.backward()
        res = []
        for k, p in m.named_parameters():
            flat = p.grad.detach().float().flatten()
            assert torch.isfinite(flat).all(), k
            pos = synth.sample_positions(flat.numel(), cases.NSAMP, synth.name_seed("pos_model.d_" + k), device=flat.device)[:256]
            gs = torch.as_tensor(g[f"model.d_{k}.samples"]).double()
            res.append((((flat[pos].cpu().double() - gs).norm() / gs.norm().clamp_min(1e-30)).item(),
                        abs(flat.double().norm().item() - float(g[f"model.d_{k}.l2"][0])) / float(g[f"model.d_{k}.l2"][0]), k))
        pos = synth.sample_positions(out.numel(), cases.NSAMP, synth.name_seed("pos_model.out"), device="cuda")
        go = torch.as_tensor(g["model.out.samples"]).double()
        return loss.item(), ((out.detach().flatten()[pos].cpu().double() - go).norm() / go.norm()).item(), res

    # fp32: absolute bounds.  bf16: the yardstick is the REFERENCE'S OWN bf16 -- the same forward + backward run by
    # the reference under torch.autocast("cpu", dtype=torch.bfloat16) (models/pangu_sample.py:46-47, commented out there), its error
    # against its own fp32 autograd stored PER TENSOR with this test's metrics (tests/golden/autocast.npz, oracle/gen_golden.py
    # autocast_grads: sample error median 1.4e-2, worst 5.7e-1 on the LayerNorm parameters, which autocast reduces in bf16).
    # Every HIP bf16 gradient tensor must be within 1.5x of the reference-autocast error OF THAT TENSOR (absolute floor for the
    # tensors the reference's autocast happens to get nearly exactly), and so must the output.
    ac = np.load(os.path.join(golden_dir, "autocast.npz"))
    loss, oerr, res = run(torch.float32)
    worst_s, worst_n = max(res), max((r[1], r[2]) for r in res)
    print(f"REFINIT fp32: loss {loss:.8f} (reference {float(g['model.loss'][0]):.8f}), output rel-L2 {oerr:.2e}; gradient samples worst "
          f"{worst_s[0]:.2e} ({worst_s[2]}); gradient norm worst {worst_n[0]:.2e} ({worst_n[1]})")
    assert oerr < 1e-4 and worst_s[0] < 2e-3 and worst_n[0] < 1e-3 and abs(loss - float(g["model.loss"][0])) < 2e-6
    loss, oerr, res = run(BF)
    ref_s, ref_n, ref_o = ac["grads.sample_err"], ac["grads.norm_err"], float(ac["grads.out_err"][0])
    assert len(ref_s) == len(res)
    FLOOR_S, FLOOR_N = 2.5e-2, 1e-2
    ratios = sorted(((r[0] / max(rs, 1e-30), r[0], rs, r[2]) for r, rs in zip(res, ref_s)), reverse=True)
    med = sorted(r[0] for r in res)[len(res) // 2]
    print(f"REFINIT bf16: loss {loss:.8f} (reference fp32 {float(g['model.loss'][0]):.8f}, reference autocast {float(ac['grads.loss'][0]):.8f}), "
          f"output rel-L2 {oerr:.2e} (reference autocast {ref_o:.2e}); gradient samples worst {max(res)[0]:.2e} ({max(res)[2]}), median "
          f"{med:.2e} (reference autocast worst {ref_s.max():.2e}, median {np.median(ref_s):.2e})")
    print("largest HIP-bf16 / reference-autocast sample-error ratios:", [(f"{a:.2f}", f"{b:.1e}", f"{c:.1e}", k) for a, b, c, k in ratios[:6]])
    bad = [(k, e, rs) for (e, _, k), rs in zip(res, ref_s) if e > max(1.5 * rs, FLOOR_S)]
    bad_n = [(k, n, rn) for (_, n, k), rn in zip(res, ref_n) if n > max(1.5 * rn, FLOOR_N)]
    assert not bad, bad
    assert not bad_n, bad_n
    assert oerr <= max(1.5 * ref_o, 1e-2)
    assert abs(loss - float(g["model.loss"][0])) < 2e-5


@pytest.mark.parametrize("s1,s2", [(0.0, 1.25), (1.25, 0.0), (1.25, 1.25), (1.0, 1.0)])
def test_block_bf16_nograd_droppath(P, s1, s2):
    """bf16 inference block in train() mode under no_grad: DropPath factors per branch like the fp32 path (a dropped
    branch is skipped); (1,1) takes the fused projection+LN launches.  Compared with the fp32 oracle expression."""
    from pangu_pytorch_amd import fused_bf16
    C, roll, W = 192, True, 12
    st = cases.STAGES[C]
    blk = P.layers.EarthSpecificBlock(C, 0.2, st["heads"], device="cuda").cuda().train()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, s, "cuda") for k, s in cases.block_param_shapes(C).items()})
    seq = iter([s1, s2])
    blk.drop_path.sample_scale = lambda training: next(seq)
    x = cases.block_input(C, W)
    with torch.no_grad():
        y = fused_bf16._block(blk, fused_bf16.WeightShadow(), x[0].cuda().to(BF).contiguous(), st["Z"], st["H"], W, roll)
        p = cases.block_params(C, roll)
        g = lambda k: p[pre + k]
        xr = x.to(BF).float()
        a = O.window_attention(xr, g("attention.linear1.weight"), g("attention.linear1.bias"), g("attention.linear2.weight"),
                               g("attention.linear2.bias"), g("attention.earth_specific_bias"), st["Z"], st["H"], W,
                               st["heads"], roll)
        x1 = xr + s1 * torch.nn.functional.layer_norm(a, (C,), g("norm1.weight"), g("norm1.bias"))
        m = O.mlp(x1, g("linear.linear1.weight"), g("linear.linear1.bias"), g("linear.linear2.weight"), g("linear.linear2.bias"))
        ref = x1 + s2 * torch.nn.functional.layer_norm(m, (C,), g("norm2.weight"), g("norm2.bias"))
    assert y.dtype == BF and rel_err(y, ref[0]) < 3e-2


def test_bf16_batch_of_two(P):
    """bf16 path, batch of two: the forward equals the two samples run one by one (bit for bit), and the autograd path
    gives per-sample input-independent parameter gradients that sum (checked on the loss and on every gradient)."""
    from pangu_pytorch_amd import train
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    m.set_compute_dtype(BF)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    tgt, tgt_s = cases.model_targets("cuda")
    inp2 = torch.cat((inp, synth.uniform(inp.shape, 4242, device="cuda")), 0)
    inp_s2 = torch.cat((inp_s, synth.uniform(inp_s.shape, 4243, device="cuda")), 0)
    tgt2, tgt_s2 = torch.cat((tgt, tgt), 0), torch.cat((tgt_s, tgt_s), 0)
    with torch.no_grad():
        o2, os2 = m(inp2, inp_s2, stats, maps, const_h)
        oa, osa = m(inp2[:1], inp_s2[:1], stats, maps, const_h)
        ob_, osb = m(inp2[1:], inp_s2[1:], stats, maps, const_h)
    assert o2.shape == (2, 5, 13, 721, 1440)
    assert torch.equal(o2[0], oa[0]) and torch.equal(o2[1], ob_[0]) and torch.equal(os2[0], osa[0]) and torch.equal(os2[1], osb[0])
    del o2, os2, oa, osa, ob_, osb
    # autograd path: mean loss over the batch of two == mean of the two single-sample losses; gradients likewise
    out, out_s = m(inp2, inp_s2, stats, maps, const_h)
    l2 = train.weighted_l1_loss(out, out_s, tgt2, tgt_s2)
    l2.backward()
    g2 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    del out, out_s
    ls = []
    for i in range(2):
        out, out_s = m(inp2[i:i + 1], inp_s2[i:i + 1], stats, maps, const_h)
        li = train.weighted_l1_loss(out, out_s, tgt, tgt_s)
        (0.5 * li).backward()
        ls.append(li.item())
        del out, out_s
    assert abs(l2.item() - 0.5 * (ls[0] + ls[1])) < 1e-5 * abs(l2.item())
    worst = max((((p.grad.double() - g2[k].double()).norm() / g2[k].double().norm().clamp_min(1e-30)).item(), k)
                for k, p in m.named_parameters())
    assert worst[0] < 1e-3, worst
    m.set_compute_dtype(torch.float32)

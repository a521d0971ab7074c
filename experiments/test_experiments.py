"""Parity tests of the experiments library (kernels that lost their A/B; see README.md here).  Not collected by the driver's
`pytest tests/`:   make -C experiments && python -m pytest experiments -q -m experiments      (on a GPU box)"""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import cases   # noqa: E402
import pangu_oracle as O   # noqa: E402
import synth   # noqa: E402

pytestmark = [pytest.mark.experiments, pytest.mark.gpu]
BF = torch.bfloat16
TIGHT, ROUND = 2e-4, 1.0 / 128        # as tests/test_gpu_parity.py / tests/test_gpu_bf16.py


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def P():
    import pangu_pytorch_amd as P
    P._lib.load()
    return P


@pytest.fixture(scope="module")
def E():
    import exp_ops
    exp_ops.load()
    return exp_ops


@pytest.mark.parametrize("M,scale", [(1000, 1.0), (128 * 5 + 37, 1.25), (31, 0.5), (4099, 1.0)])
@pytest.mark.parametrize("strided", [False, True])
@pytest.mark.parametrize("C", [192])
def test_mlp_ln_residual_fused_f32(P, E, M, scale, strided, C):
    """The whole MLP branch in one fp32 launch (csrc/mlp_fused_f32.hip: hidden activation on chip) == the reference's op chain
    x + scale * norm2(linear2(GELU(linear1(x)))) (layers.py:251, :264-270; exact-erf GELU) evaluated in fp64 on the same inputs;
    ragged token counts (tiles of 128), row-strided input / output (the skip-concat halves), and == the two-launch path."""
    x = synth.uniform((M, C), 181, 1.5)
    w1 = synth.uniform((4 * C, C), 182, 1.0 / C ** 0.5)
    b1 = synth.uniform((4 * C,), 183, 0.5)
    w2 = synth.uniform((C, 4 * C), 184, 0.5 / C ** 0.5)
    b2 = synth.uniform((C,), 185, 0.5)
    g, be = synth.uniform((C,), 186, 0.5, 1.0), synth.uniform((C,), 187, 0.3)
    hd = torch.nn.functional.gelu(x.double() @ w1.double().t() + b1.double())
    m = hd @ w2.double().t() + b2.double()
    ref = x.double() + scale * torch.nn.functional.layer_norm(m, (C,), g.double(), be.double(), 1e-5)
    xd, out = x.cuda(), None
    if strided:
        xf = torch.zeros((M, 2 * C), device="cuda")
        xf[:, C:] = xd
        xd = xf[:, C:]
        full = torch.zeros((M, 2 * C), device="cuda")
        out = full[:, :C]
    args = (w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), g.cuda(), be.cuda())
    got = E.mlp_ln_residual_f32(xd, *args, out=out, branch_scale=scale)
    assert rel_err(got, ref) < TIGHT
    if strided:
        assert float(full[:, C:].abs().max()) == 0.0
    two = P.ops.linear_ln_residual(P.ops.linear(xd, args[0], args[1], act=P.ops.ACT_GELU), args[2], args[3], xd, args[4], args[5],
                                   branch_scale=scale)
    assert rel_err(got, two) < 2e-5          # the same fp32 MFMA arithmetic, another summation order


@pytest.mark.parametrize("variant", [40, 30, 20, 21, 11])
@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_qkv_walk_bf16(P, E, C, shifted, variant):
    """The longitude-walking form of the fused QKV attention (csrc/attn_walk_bf16.hip: one persistent workgroup per (window type,
    head), linear1's rows resident in LDS, `variant // 10` window pipelines of three waves, variant % 10 == 1: bias rows resident
    in registers -- reference layers.py:306-311,395: one bias per (type, head), broadcast over longitude) on SEVEN longitude
    windows (every pipeline walks more than one window and they take unequal shares) == the oracle on the same bf16-rounded
    operands, and == the (window, head) kernel bit for bit on the attention output (same tile code, same operand values)."""
    from pangu_pytorch_amd import ops_bf16 as ob
    walk = lambda *a, variant, **k: E.window_attention_qkv_walk(*a, variant=variant, **k)
    st = cases.STAGES[C]
    Z, H, W, heads = st["Z"], st["H"], 84, st["heads"]
    N = Z * H * W
    x = synth.uniform((N, C), 35, 1.5).to(BF)
    w = synth.uniform((3 * C, C), 36, 1.5 / C ** 0.5).to(BF)
    b = synth.uniform((3 * C,), 37, 0.5)
    esb = synth.uniform((1, st["types"], heads, 144, 144), 38, 0.5).to(BF)
    base, base_lse = ob.window_attention_qkv(x.cuda(), w.cuda(), b.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    got, lse = walk(x.cuda(), w.cuda(), b.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True, variant=variant)
    torch.cuda.synchronize()
    # q, k, v: the same MFMA chain over the same 32-channel steps, bias added after the chain instead of as its initial value
    # (one fp32 rounding apart before the bf16 rounding of q / k / v): bf16-rounding-level agreement with the other kernel
    assert rel_err(got, base) < ROUND and rel_err(lse, base_lse) < 2e-3
    if variant in (40, 21):      # and against the oracle (the pad rows' q/k/v from the bf16-rounded bias, layers.py:192)
        qkv = (x.double() @ w.double().t() + b.double()).to(BF)
        ref, ref_lse = O.window_attention_core(qkv.float()[None], b.to(BF).float(), esb.float(), Z, H, W, heads, shifted)
        assert rel_err(got, ref[0]) < ROUND
        assert rel_err(lse, ref_lse[0]) < 2e-3
    # every launch of the same inputs gives the same bits (the pipelines' rendezvous orders all LDS traffic)
    again, _ = walk(x.cuda(), w.cuda(), b.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True, variant=variant)
    assert torch.equal(again, got)



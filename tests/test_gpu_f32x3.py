"""Opt-in split-bf16 ("bf16x3") fp32 projection path: kernel accuracy vs fp64, whole-forward parity vs the reference."""
import os

import numpy as np
import pytest
import torch

import cases
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pangu_pytorch_amd as P
    P._lib.load()
    return P


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K,act,bias", [
    (1000, 192, 192, 0, True), (4099, 576, 192, 0, True), (2048, 768, 192, 1, True), (777, 192, 768, 0, True),
    (1531, 1152, 384, 0, True), (513, 1536, 384, 1, True), (300, 384, 1536, 0, False), (1234, 160, 384, 0, True),
    (321, 64, 384, 0, True), (650, 192, 112, 0, True), (128, 128, 16, 0, False),
])
def test_linear_f32x3(P, M, N, K, act, bias):
    a = synth.uniform((M, K), 11)
    w = synth.uniform((N, K), 12, 1.0 / K ** 0.5)
    b = synth.uniform((N,), 13, 0.5) if bias else None
    ref = a.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    with P.ops.f32_split(True):
        got = P.ops.linear(a.cuda(), w.cuda(), b.cuda() if bias else None, act=act)
    exact = P.ops.linear(a.cuda(), w.cuda(), b.cuda() if bias else None, act=act)
    e3, e1 = rel_err(got, ref), rel_err(exact, ref)
    assert e3 < 3e-5, (e3, e1)          # split products: ~2^-16 per term, averaged down by the sum


def test_linear_f32x3_gelu_aux_and_bwd(P):
    M, N, K = 1500, 768, 192
    a, w, b = synth.uniform((M, K), 65), synth.uniform((N, K), 66, 0.1), synth.uniform((N,), 67, 0.3)
    pre = torch.empty((M, N), device="cuda")
    with P.ops.f32_split(True):
        h = P.ops.linear(a.cuda(), w.cuda(), b.cuda(), act=P.ops.ACT_GELU, aux=pre)
        ref_pre = a @ w.t() + b
        assert rel_err(pre, ref_pre) < 3e-5 and rel_err(h, torch.nn.functional.gelu(ref_pre)) < 3e-5
        dm, w2 = synth.uniform((M, 192), 68), synth.uniform((192, N), 69, 0.05)
        got = P.ops.linear(dm.cuda(), w2.t().contiguous().cuda(), None, act=P.ops.ACT_GELU_BWD, aux=pre)
    x = ref_pre.clone().requires_grad_(True)
    (torch.nn.functional.gelu(x) * (dm @ w2)).sum().backward()
    assert rel_err(got, x.grad) < 5e-5


def test_full_model_f32x3_parity(P, golden_dir):
    """Whole forward with split-bf16 projections vs the REFERENCE's fp32 CPU forward: inside the 1e-3 contract."""
    g = np.load(os.path.join(golden_dir, "model_fwd.npz"))
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        exact, _ = m(inp, inp_s, stats, maps, const_h)
        m.set_compute_dtype(torch.float32, f32_split=True)
        out, out_s = m(inp, inp_s, stats, maps, const_h)
        m.set_compute_dtype(torch.float32)
    e_ref = cases.compare_summary(out, g, "model.out", 1e-3)
    e_ref_s = cases.compare_summary(out_s, g, "model.out_surface", 1e-3)
    l2 = ((out.double() - exact.double()).norm() / exact.double().norm()).item()
    print(f"f32x3 forward: fingerprint err vs reference {e_ref:.2e} / {e_ref_s:.2e}; rel-L2 vs exact-f32 path {l2:.2e}")
    assert e_ref < 1e-3 and e_ref_s < 1e-3

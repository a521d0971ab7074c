"""hipGraph-captured step and autoregressive rollout (BASELINE configs[4]) on the GPU."""
import pytest
import torch

import cases
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import pangu_pytorch_amd as P
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    return P, m, cases.model_inputs("cuda")


def _stats_last(stats, dev):
    s_mean, s_std, u_mean, u_std = stats
    return (s_mean.view(1, 4, 1, 1), s_std.view(1, 4, 1, 1),
            u_mean.reshape(13, 5).flip(0).t().reshape(1, 5, 13, 1, 1).contiguous(),
            u_std.reshape(13, 5).flip(0).t().reshape(1, 5, 13, 1, 1).contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_step_equals_eager(setup, dtype):
    P, m, (inp, inp_s, stats, maps, const_h) = setup
    m.set_compute_dtype(dtype)
    try:
        with torch.no_grad():
            ref, ref_s = m(inp, inp_s, stats, maps, const_h)
        g = P.rollout.GraphedStep(m, inp, inp_s, stats, maps, const_h)
        out, out_s = g.step()
        assert torch.equal(out, ref) and torch.equal(out_s, ref_s)          # same kernels, same order: bit-identical
        inp2 = synth.uniform(inp.shape, 777, device="cuda")
        g.load(inp2, inp_s)
        out2, _ = g.step()
        with torch.no_grad():
            ref2, _ = m(inp2, inp_s, stats, maps, const_h)
        assert torch.equal(out2, ref2)
    finally:
        m.set_compute_dtype(torch.float32)


def test_rollout_graph_equals_eager_and_bf16_drift(setup):
    """3 chained steps: graph == eager bit-for-bit (fp32); bf16 rollout drift vs fp32 reported per step."""
    P, m, (inp, inp_s, stats, maps, const_h) = setup
    sl = _stats_last(stats, "cuda")
    up_e, sf_e, hist_e = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=3, graph=False, keep=True)
    up_g, sf_g, hist_g = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=3, graph=True, keep=True)
    assert torch.equal(up_e, up_g) and torch.equal(sf_e, sf_g)
    m.set_compute_dtype(torch.bfloat16)
    try:
        _, _, hist_b = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=3, graph=True, keep=True)
    finally:
        m.set_compute_dtype(torch.float32)
    drifts = [((b[0].double() - e[0].double()).norm() / e[0].double().norm()).item() for b, e in zip(hist_b, hist_e)]
    print("bf16 rollout rel-L2 drift per step:", ["%.3e" % d for d in drifts])
    assert all(torch.isfinite(b[0]).all() for b in hist_b)
    assert drifts[0] < 5e-2 and drifts[-1] < 0.3

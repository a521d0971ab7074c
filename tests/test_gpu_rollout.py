"""hipGraph-captured step and autoregressive rollout (BASELINE configs[4]) on the GPU."""
import os

import numpy as np
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


def test_rollout_two_steps_vs_reference_golden(setup, golden_dir):
    """Two chained 24 h steps (the loop of reference inference/inference_singleOutput.py:97-105 with normBackData,
    era5_data/utils_data.py:324-330, between them) through the fp32 hipGraph rollout == the REFERENCE run the same way
    (tests/golden/rollout2.npz, oracle/gen_golden.py rollout2): normalised outputs of both steps and the final
    physical-unit fields within BASELINE's 1e-3."""
    P, m, (inp, inp_s, stats, maps, const_h) = setup
    g = np.load(os.path.join(golden_dir, "rollout2.npz"))
    sl = _stats_last(stats, "cuda")
    for graph in (True, False):
        up, sf, hist = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=2, graph=graph, keep=True)
        for k, (o, os_) in enumerate(hist):
            assert cases.compare_summary(o, g, f"rollout.step{k + 1}.out", 1e-3) < 1e-3, (graph, k)
            assert cases.compare_summary(os_, g, f"rollout.step{k + 1}.out_surface", 1e-3) < 1e-3, (graph, k)
        assert cases.compare_summary(up, g, "rollout.final_upper", 1e-3) < 1e-3
        assert cases.compare_summary(sf, g, "rollout.final_surface", 1e-3) < 1e-3


def test_rollout_seven_steps_vs_reference_golden(setup, golden_dir):
    """BASELINE configs[4] pinned to the REFERENCE for all seven 24 h steps: the loop of reference
    inference/inference_singleOutput.py:97-105 with normBackData (era5_data/utils_data.py:324-330) between the steps, run with the
    reference's torch model in the build container (oracle/gen_golden.py rollout7 -> tests/golden/rollout7.npz, fingerprints of
    every step's normalised outputs and of the final physical fields).  fp32 hipGraph rollout: every step within BASELINE's 1e-3;
    bf16 rollout: per-step fingerprint error against the reference itself, bounded."""
    P, m, (inp, inp_s, stats, maps, const_h) = setup
    g = np.load(os.path.join(golden_dir, "rollout7.npz"))
    g2 = np.load(os.path.join(golden_dir, "rollout2.npz"))
    for k in ("rollout.step1.out", "rollout.step2.out"):          # the 7-step run reproduces the committed 2-step run
        for part in (".samples", ".abs_sum"):
            assert np.array_equal(g[k + part], g2[k + part]), k
    sl = _stats_last(stats, "cuda")
    up, sf, hist = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
    errs = [max(cases.compare_summary(o, g, f"rollout.step{k + 1}.out", 1e-3),
                cases.compare_summary(os_, g, f"rollout.step{k + 1}.out_surface", 1e-3)) for k, (o, os_) in enumerate(hist)]
    print("fp32 7-step rollout, fingerprint error vs the reference per step:", ["%.2e" % e for e in errs])
    assert all(e < 1e-3 for e in errs), errs
    assert cases.compare_summary(up, g, "rollout.final_upper", 1e-3) < 1e-3
    assert cases.compare_summary(sf, g, "rollout.final_surface", 1e-3) < 1e-3
    m.set_compute_dtype(torch.bfloat16)
    try:
        _, _, hist_b = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
    finally:
        m.set_compute_dtype(torch.float32)
    errs_b = [cases.compare_summary(o, g, f"rollout.step{k + 1}.out", 1.0) for k, (o, _) in enumerate(hist_b)]
    print("bf16 7-step rollout, fingerprint error vs the reference per step:", ["%.2e" % e for e in errs_b])
    # The yardstick is the REFERENCE'S OWN bf16: the same seven steps run by the reference under
    # torch.autocast("cpu", dtype=torch.bfloat16) -- the switch its authors left commented out, models/pangu_sample.py:46-47 --
    # measured with the same fingerprint metric against the same fp32 goldens (tests/golden/autocast.npz, oracle/gen_golden.py
    # autocast_rollout: 2.0e-2 at step 1 growing to 2.8e-1 at step 7; these O(1)-activation weights are not contractive, one
    # step's rounding is carried into the next).  The HIP bf16 rollout must stay within 1.5x of that at EVERY step (floor 3e-2).
    ac = np.load(os.path.join(golden_dir, "autocast.npz"))
    ref_b = [float(ac[f"rollout.step{k + 1}.err"][0]) for k in range(7)]
    print("reference autocast-bf16 7-step rollout, same metric:              ", ["%.2e" % e for e in ref_b])
    assert all(torch.isfinite(o).all() for o, _ in hist_b)
    assert all(e <= max(1.5 * r, 3e-2) for e, r in zip(errs_b, ref_b)), (errs_b, ref_b)


def test_rollout_7x24h_bf16_drift_bounds(setup, golden_dir):
    """BASELINE configs[4]: 7 x 24 h in bf16, one hipGraph launch per step.  Step 1 and 2 against the REFERENCE's chained
    forwards (rollout2.npz), every step against the fp32 HIP rollout: drift bounded per step (measured 0.7-1.5e-2 rel-L2;
    the reference's own bf16 autocast drifts 3.8e-3 per block)."""
    P, m, (inp, inp_s, stats, maps, const_h) = setup
    g = np.load(os.path.join(golden_dir, "rollout2.npz"))
    sl = _stats_last(stats, "cuda")
    _, _, hist32 = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
    m.set_compute_dtype(torch.bfloat16)
    try:
        up, sf, hist = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
    finally:
        m.set_compute_dtype(torch.float32)
    assert torch.isfinite(up).all() and torch.isfinite(sf).all()
    drifts = [((b[0].double() - e[0].double()).norm() / e[0].double().norm()).item() for b, e in zip(hist, hist32)]
    print("bf16 7-step rollout rel-L2 drift vs fp32 per step:", ["%.3e" % d for d in drifts])
    # the goldens' O(1)-activation synthetic weights are not contractive: the bf16 error of one step (1.5e-2) is carried
    # into the next, so the bound grows linearly with the step; reference-initialised weights: next test
    assert all(d < 2.2e-2 * (k + 1) for k, d in enumerate(drifts)), drifts
    for k in range(2):      # fingerprint error vs the reference itself (max over samples / column sums / mass, relative)
        e = cases.compare_summary(hist[k][0], g, f"rollout.step{k + 1}.out", 1.0)
        assert e < 0.12, (k, e)


def test_rollout_7x24h_bf16_drift_reference_init():
    """The same 7-step bf16 rollout with the REFERENCE's initialisation (trunc-normal 0.02 weights, unit LayerNorms,
    pangu_model.py:41-48; the weights bench.py uses): per-step drift vs the fp32 rollout stays below 4e-2 at every step
    (measured 0.9e-2 after one step, 2.7e-2 after seven)."""
    import pangu_pytorch_amd as P
    torch.manual_seed(0)
    m = P.PanguModel(device="cuda").cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    u = lambda shape, scale=1.0, shift=0.0: (torch.rand(shape, generator=g, device="cuda") * 2 - 1) * scale + shift
    inp, inp_s = u((1, 5, 13, 721, 1440)), u((1, 4, 721, 1440))
    stats = (u((4,), 0.3), u((4,), 0.2, 1.2), u((13, 1, 1, 5), 0.3), u((13, 1, 1, 5), 0.2, 1.2))
    maps, const_h = u((1, 3, 724, 1440)), u((1, 1, 1, 13, 721, 1440))
    sl = _stats_last(stats, "cuda")
    _, _, h32 = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
    m.set_compute_dtype(torch.bfloat16)
    _, _, hbf = P.rollout.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
    drifts = [((b[0].double() - e[0].double()).norm() / e[0].double().norm()).item() for b, e in zip(hbf, h32)]
    print("bf16 7-step rollout (reference init) rel-L2 drift per step:", ["%.3e" % d for d in drifts])
    assert drifts[0] < 1.5e-2 and max(drifts) < 4e-2, drifts

"""The host-fed training step (reference models/pangu_sample.py:41-43,57,77 and era5_data/utils_data.py:16-51,117):
level reversal and target normalisation folded into the first / last kernels, the threaded page-locked pipeline, the step fed from
pageable host batches at full size -- and the gradients of the raw fields (reference models/layers.py:40-93 is plain autograd)."""
import os
import time

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


def _embed_inputs(LAT, LON, dev="cpu"):
    H4 = (LAT + 3) // 4
    g = lambda n, s, sc=1.0, sh=0.0: synth.uniform(s, synth.name_seed(n), sc, sh, device=dev)
    inp, inp_s = g("fi", (1, 5, 13, LAT, LON)), g("fis", (1, 4, LAT, LON))
    stats = (g("fsm", (4,), 0.3), g("fss", (4,), 0.2, 1.2), g("fum", (13, 1, 1, 5), 0.3), g("fus", (13, 1, 1, 5), 0.2, 1.2))
    return inp, inp_s, stats, g("fm", (1, 3, 4 * H4, LON)), g("fc", (1, 1, 1, 13, LAT, LON))


@pytest.mark.parametrize("LAT,LON", [(41, 280), (721, 1440)])
def test_patch_embed_gather_levels_reversed_is_an_address(P, LAT, LON):
    """`levels_reversed`: gathering a field stored in file order (ascending levels) == gathering its `flip(-3)` the plain way, bit
    for bit, fp32 and bf16 (the reader's `[::-1]`, reference era5_data/utils_data.py:117, costs no pass over the field)."""
    from pangu_pytorch_amd import ops_bf16 as ob
    inp, inp_s, stats, maps, const_h = _embed_inputs(LAT, LON, "cuda")
    args = (inp_s[0], stats[0], stats[1], stats[2].reshape(13, 5), stats[3].reshape(13, 5), maps[0], const_h.reshape(13, LAT, LON))
    stored = inp[0].flip(-3).contiguous()                 # what the file holds
    for mod in (P.ops, ob):
        a_s, a_u = mod.patch_embed_gather(inp[0], *args)
        b_s, b_u = mod.patch_embed_gather(stored, *args, levels_reversed=True)
        assert torch.equal(a_s, b_s) and torch.equal(a_u, b_u)
        c_s, c_u = mod.patch_embed_gather(stored, *args)                      # (and the flag is not a no-op)
        assert not torch.equal(a_u, c_u)


@pytest.mark.parametrize("LAT,LON", [(41, 280), (721, 1440)])
@pytest.mark.parametrize("rev", [False, True])
def test_patch_embed_input_gradients_vs_oracle_autograd(P, LAT, LON, rev):
    """d_input / d_input_surface of the patch embedding (reference models/layers.py:40-93, plain autograd there)
    against torch autograd over the oracle's restatement of the same gather on the CPU.  The scatter is pure data movement and the
    division by the std is one IEEE division per element, as autograd's: the kernel alone is BIT-EXACT against autograd given the same
    dA; through PatchEmbedFn (dA from the HIP GEMM) the fields' gradients agree to GEMM rounding.  Ragged sizes and the model's."""
    from pangu_pytorch_amd.autograd import PatchEmbedFn
    H4, W4 = (LAT + 3) // 4, LON // 4
    inp, inp_s, stats, maps, const_h = _embed_inputs(LAT, LON)
    # (a) the kernel against autograd of the gather for a given dA
    da_s, da_u = synth.uniform((H4 * W4, 112), synth.name_seed("da_s")), synth.uniform((7 * H4 * W4, 192), synth.name_seed("da_u"))
    xi, xs = inp.clone().requires_grad_(True), inp_s.clone().requires_grad_(True)
    ra_s, ra_u = O.patch_embed_matrices(xi, xs, stats, maps, const_h)
    ((ra_s[0] * da_s).sum() + (ra_u[0] * da_u).sum()).backward()
    d_in, d_in_s = P.ops.patch_embed_gather_bwd(da_s[:, :64].contiguous().cuda(), da_u[:, :160].contiguous().cuda(), stats[1].cuda(),
                                                stats[3].reshape(13, 5).cuda(), LAT, LON, levels_reversed=rev)
    want = xi.grad[0].flip(-3) if rev else xi.grad[0]
    assert torch.equal(d_in.cpu(), want) and torch.equal(d_in_s.cpu(), xs.grad[0])
    # (b) through the autograd Function: parameters frozen or not, the fields get their gradients
    cw, sw = synth.uniform((192, 192, 1), 11, 0.1), synth.uniform((192, 112, 1), 12, 0.1)
    cb, sb = synth.uniform((192,), 13, 0.1), synth.uniform((192,), 14, 0.1)
    cot = synth.uniform((8 * H4 * W4, 192), synth.name_seed("emb_cot"))
    xi.grad = xs.grad = None
    ref = O.patch_embed({"_input_layer.conv.weight": cw, "_input_layer.conv.bias": cb, "_input_layer.conv_surface.weight": sw,
                         "_input_layer.conv_surface.bias": sb}, xi, xs, stats, maps, const_h)
    (ref[0] * cot).sum().backward()
    gi = (inp[0].flip(-3) if rev else inp[0]).contiguous().cuda().requires_grad_(True)
    gs = inp_s[0].cuda().requires_grad_(True)
    x = PatchEmbedFn.apply(cw.cuda(), cb.cuda(), sw.cuda(), sb.cuda(), gi, gs, stats[0].cuda(), stats[1].cuda(),
                           stats[2].reshape(13, 5).cuda(), stats[3].reshape(13, 5).cuda(), maps[0].cuda(),
                           const_h.reshape(13, LAT, LON).cuda(), rev)
    (x * cot.cuda()).sum().backward()
    want = xi.grad[0].flip(-3) if rev else xi.grad[0]
    rel = lambda a, b: ((a.cpu() - b).abs().max() / b.abs().max()).item()
    assert rel(x.detach(), ref[0].detach()) < 1e-5
    assert rel(gi.grad, want) < 1e-5 and rel(gs.grad, xs.grad[0]) < 1e-5


def test_model_input_gradients_and_refusals(P, golden_dir):
    """`input.requires_grad_()` through the WHOLE model (all parameters frozen: the autograd path must still run), fp32 and bf16,
    eval-recompute and saving modes: input.grad against the REFERENCE's own autograd (tests/golden/model_bwd_input.npz, written by
    oracle/gen_golden.py model_bwd_input: fingerprints of d loss / d input for the smooth loss on the golden weights) when the
    fixture exists, and always: both modes agree, a field that did not ask gets None, and the constant operands
    (maps, const_h) asking for a gradient are refused instead of silently getting None."""
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    for p in m.parameters():
        p.requires_grad_(False)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    cot, cot_s = cases.cotangent("model_out", (1, 5, 13, 721, 1440), "cuda"), cases.cotangent("model_out_s", (1, 4, 721, 1440), "cuda")
    gpath = os.path.join(golden_dir, "model_bwd_input.npz")
    g = np.load(gpath) if os.path.exists(gpath) else None
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        m.set_compute_dtype(dt)
        for mode in ("recompute", "save"):
            m.eval_grad_mode = mode
            xi, xs = inp.clone().requires_grad_(True), inp_s.clone().requires_grad_(True)
            out, out_s = m(xi, xs, stats, maps, const_h)
            loss = ((out * cot).sum() + (out_s * cot_s).sum()) / out.numel()
            loss.backward()
            assert xi.grad is not None and xs.grad is not None and torch.isfinite(xi.grad).all()
            res[(dt, mode)] = (xi.grad, xs.grad)
        for a, b in zip(res[(dt, "recompute")], res[(dt, "save")]):      # the same autograd graph, run twice
            assert ((a.double() - b.double()).norm() / b.double().norm()).item() < 1e-5
        if g is not None:
            # bf16 on the goldens' O(1)-activation weights: 16 blocks of non-contractive bf16 drift feed the backward (the parameter
            # gradients of this regime need 0.35, tests/test_gpu_bf16.py); measured 0.28 -- the tight bf16 bound is the refinit one below
            tol = 1e-3 if dt == torch.float32 else 0.4
            e = max(cases.compare_summary(res[(dt, "save")][0], g, "model.d_input", tol),
                    cases.compare_summary(res[(dt, "save")][1], g, "model.d_input_surface", tol))
            print(f"input gradients vs the reference's autograd, {dt}: fingerprint error {e:.2e}")
            assert e < tol
    # bf16 against fp32 in the REFERENCE'S initialisation regime (weights std 0.02: contractive, like a trained model)
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda", spec="refinit"))
    m.eval_grad_mode = "save"
    gi = {}
    for dt in (torch.float32, torch.bfloat16):
        m.set_compute_dtype(dt)
        xi, xs = inp.clone().requires_grad_(True), inp_s.clone().requires_grad_(True)
        out, out_s = m(xi, xs, stats, maps, const_h)
        (((out * cot).sum() + (out_s * cot_s).sum()) / out.numel()).backward()
        gi[dt] = (xi.grad.double(), xs.grad.double())
    drift = max(((b - a).norm() / a.norm()).item() for a, b in zip(gi[torch.float32], gi[torch.bfloat16]))
    print(f"input gradients, refinit weights: bf16 vs fp32 rel-L2 {drift:.2e}")
    assert drift < 5e-2
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    m.eval_grad_mode = "recompute"
    m.set_compute_dtype(torch.float32)
    # only one field asks
    xi = inp.clone().requires_grad_(True)
    out, _ = m(xi, inp_s, stats, maps, const_h)
    out.sum().backward()
    assert xi.grad is not None and inp_s.grad is None
    # the constants are refused
    with pytest.raises(RuntimeError, match="maps.requires_grad"):
        m(inp, inp_s, stats, maps.clone().requires_grad_(True), const_h)
    with pytest.raises(RuntimeError, match="const_h.requires_grad"):
        m(inp, inp_s, stats, maps, const_h.clone().requires_grad_(True))


@pytest.mark.parametrize("shape", [(1, 13, 721, 1440), (2, 3, 37, 24), (1, 2, 5, 7)])
def test_loss_reversed_targets_and_fused_normdata(P, shape):
    """The loss kernel's target side: targets in physical units normalised on the fly (`normData`, reference
    era5_data/utils_data.py:315-321 at models/pangu_sample.py:57) and / or stored with ascending levels (utils_data.py:117) ==
    the torch expression on targets flipped and normalised beforehand: the value to fp64-accumulation accuracy, the gradient bit
    for bit; and == the same kernel fed the pre-processed targets, bit for bit (the fold changes no rounding)."""
    from pangu_pytorch_amd import train
    B, L, H, W = shape
    o = synth.uniform((B, 5, L, H, W), synth.name_seed("l2_o"), device="cuda")
    t_phys = synth.uniform((B, 5, L, H, W), synth.name_seed("l2_t"), 40.0, 250.0, device="cuda")      # K-like magnitudes
    os_ = synth.uniform((B, 4, H, W), synth.name_seed("l2_os"), device="cuda")
    ts_phys = synth.uniform((B, 4, H, W), synth.name_seed("l2_ts"), 500.0, 1e5, device="cuda")        # Pa-like
    sl = (synth.uniform((1, 4, 1, 1), 21, 300.0, 1e5, device="cuda"), synth.uniform((1, 4, 1, 1), 22, 100.0, 700.0, device="cuda"),
          synth.uniform((1, 5, L, 1, 1), 23, 20.0, 250.0, device="cuda"), synth.uniform((1, 5, L, 1, 1), 24, 5.0, 30.0, device="cuda"))
    t_norm, ts_norm = train.norm_data(t_phys, ts_phys, sl)
    for rev, st in ((False, None), (True, None), (False, sl), (True, sl)):
        tgt = (t_phys if st is not None else t_norm)
        tgt_s = ts_phys if st is not None else ts_norm
        stored = tgt.flip(-3).contiguous() if rev else tgt
        o1, os1 = o.clone().requires_grad_(True), os_.clone().requires_grad_(True)
        loss = train.weighted_l1_loss(o1, os1, stored, tgt_s, target_levels_reversed=rev, stats_last=st)
        (loss * 1.3).backward()
        o2, os2 = o.clone().requires_grad_(True), os_.clone().requires_grad_(True)
        ref = train._weighted_l1_loss_torch(o2, os2, t_norm, ts_norm)
        (ref * 1.3).backward()
        o3, os3 = o.clone().requires_grad_(True), os_.clone().requires_grad_(True)
        plain = train.weighted_l1_loss(o3, os3, t_norm, ts_norm)
        (plain * 1.3).backward()
        ref64 = O.train_loss(o.cpu().double(), os_.cpu().double(), t_norm.cpu().double(), ts_norm.cpu().double())
        assert abs(loss.item() - ref64.item()) <= 2e-6 * abs(ref64.item()), (rev, st is not None)
        assert torch.equal(loss, plain), (rev, st is not None)
        assert torch.equal(o1.grad, o2.grad) and torch.equal(os1.grad, os2.grad), (rev, st is not None)


class _Filler:
    """A loader that writes straight into the page-locked buffers it is handed (data.PinnedFiller protocol)."""

    def __init__(self, n):
        self.n, self.i = n, 0
        self.spec = [((1, 5, 13, 8, 16), torch.float32), ((1, 4, 8, 16), torch.float32)]

    def reset(self):
        self.i = 0

    def __len__(self):
        return self.n

    def fill_pinned(self, bufs):
        if self.i >= self.n:
            return False
        assert all(b.is_pinned() for b in bufs)
        bufs[0].numpy()[...] = self.i
        bufs[1].numpy()[...] = -self.i
        self.i += 1
        return True


@pytest.mark.timeout(300)
@pytest.mark.parametrize("threaded", [True, False])
def test_device_prefetcher_pipeline_semantics(P, threaded):
    """data.DevicePrefetcher: order and values over more batches than slots (buffer reuse), non-tensor items passed through, the
    fused flip leaves the data in file order and says so, an early `break` and a second epoch work, a loader's exception reaches
    the consumer, and a PinnedFiller is driven without a staging copy."""
    D = P.data
    mk = lambda i: (torch.full((1, 5, 13, 8, 16), float(i)) + torch.arange(13.0).view(1, 1, 13, 1, 1), torch.full((1, 4, 8, 16), float(i)),
                    torch.full((1, 5, 13, 8, 16), -float(i)), torch.zeros(1, 4, 8, 16), {"tag": i})
    batches = [mk(i) for i in range(9)]
    pf = D.DevicePrefetcher(batches, "cuda", flip_levels=True, fuse_flip=True, depth=2, threaded=threaded, copy_threads=3)
    assert pf.levels_reversed and len(pf) == 9
    for epoch in range(2):
        got = list(pf)
        assert len(got) == 9
        for i, (a, b, c, d, tag) in enumerate(got):
            assert a.is_cuda and tag == {"tag": i}
            assert torch.equal(a.cpu(), batches[i][0]) and torch.equal(c.cpu(), batches[i][2]) and torch.equal(b.cpu(), batches[i][1])
    dev_flip = D.DevicePrefetcher(batches, "cuda", flip_levels=True, threaded=threaded)
    assert not dev_flip.levels_reversed
    for i, (a, *_rest) in enumerate(dev_flip):
        assert torch.equal(a.cpu(), batches[i][0].flip(-3))
        if i == 3:
            break                                            # the consumer leaves early: the worker must not hang
    assert len(list(dev_flip)) == 9
    torch.cuda.synchronize()
    s = pf.summary()
    assert s["batches"] == 18 and s["bytes_per_batch"] == sum(t.numel() * 4 for t in batches[0][:4]) and s["threaded"] == threaded

    def bad():
        yield mk(0)
        raise ValueError("reader failed")
    with pytest.raises(ValueError, match="reader failed"):
        list(D.DevicePrefetcher(bad(), "cuda", threaded=threaded))
    # static device buffers: a batch is valid until the next one is requested -- checked INSIDE the loop, over more batches than
    # there are buffer sets (depth + 3), with a slow consumer kernel queued on each batch before the next is requested
    st = D.DevicePrefetcher(batches * 2, "cuda", depth=2, threaded=threaded, reuse_device_buffers=True)
    seen, ptrs = [], set()
    big = torch.zeros(1 << 24, device="cuda")
    for i, (a, b, c, d, tag) in enumerate(st):
        for _ in range(20):
            big.add_(1.0)                                    # keeps the consumer's stream busy while the worker refills
        seen.append((a.clone(), tag))
        ptrs.add(a.data_ptr())
    assert len(seen) == 18 and len(ptrs) <= 5
    for i, (a, tag) in enumerate(seen):
        assert tag == {"tag": i % 9} and torch.equal(a.cpu(), batches[i % 9][0])
    assert st.summary()["static_device_buffers"]
    fl = D.DevicePrefetcher(_Filler(7), "cuda", depth=2, threaded=threaded)
    for epoch in range(2):
        vals = [(float(a[0, 0, 0, 0, 0]), float(b[0, 0, 0, 0])) for a, b in fl]
        assert vals == [(float(i), -float(i)) for i in range(7)]
    assert fl.summary()["direct_fill"]


def _host_batches(n, ascending):
    """n distinct full-size PAGEABLE host samples (input, input_surface, target, target_surface); `ascending`: the level axis in
    file order, i.e. the logical fields flipped."""
    out = []
    for i in range(n):
        u = lambda name, shape: synth.uniform(shape, synth.name_seed(f"feed_{name}_{i}"))
        inp, tgt = u("in", (1, 5, 13, 721, 1440)), u("tg", (1, 5, 13, 721, 1440))
        if ascending:
            inp, tgt = inp.flip(-3).contiguous(), tgt.flip(-3).contiguous()
        out.append((inp, u("ins", (1, 4, 721, 1440)), tgt, u("tgs", (1, 4, 721, 1440))))
    return out


@pytest.mark.timeout(600)
def test_training_step_fed_from_host_full_size(P):
    """the bf16 training step fed from PAGEABLE full-size host batches (573 MB per step, three distinct samples
    in file level order) through data.DevicePrefetcher(fuse_flip=True) with the reference's per-step `loss.item()`
    (models/pangu_sample.py:77):
      * values: the first step's loss == the same step on a resident, host-flipped batch BIT FOR BIT (same kernels, same data: the
        fused flip changes no arithmetic); the later steps agree to 1e-5 (the backward's fp32 atomics make two runs of the very same
        loop differ in the last bits of the updated weights);
      * pipeline: fed ms/step <= 1.15 x resident ms/step (the staging copy and the host->device copy hide behind the step)."""
    from pangu_pytorch_amd import train
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = P.PanguModel(device=dev).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    _, _, stats, maps, const_h = cases.model_inputs("cuda")
    host = _host_batches(3, ascending=True)
    n_steps, n_warm = 9, 3

    def run(fed):
        m.load_state_dict(state0)
        m.train()
        opt = train.make_optimizer(m)
        torch.manual_seed(77)                      # DropPath draws
        losses, t0 = [], None
        if fed:
            loader = [host[i % 3] for i in range(n_warm + n_steps)]
            pf = P.data.DevicePrefetcher(loader, dev, flip_levels=True, fuse_flip=True, depth=2, reuse_device_buffers=True)
            for k, batch in enumerate(pf):
                if k == n_warm:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                losses.append(train.train_step(m, opt, batch, stats, maps, const_h, levels_reversed=pf.levels_reversed).item())
            summary = pf.summary()
        else:
            res = [tuple(t.flip(-3).contiguous().to(dev) if j in (0, 2) else t.to(dev) for j, t in enumerate(b)) for b in host]
            for k in range(n_warm + n_steps):
                if k == n_warm:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                losses.append(train.train_step(m, opt, res[k % 3], stats, maps, const_h).item())
            summary = None
        torch.cuda.synchronize()
        return losses, (time.perf_counter() - t0) / n_steps * 1e3, summary

    l_res, ms_res, _ = run(False)
    l_fed, ms_fed, summ = run(True)
    print(f"bf16 training step with loss.item() every step: resident {ms_res:.2f} ms, fed from pageable host batches {ms_fed:.2f} ms "
          f"({ms_fed / ms_res:.3f}x); pipeline {summ}")
    assert l_fed[0] == l_res[0], (l_fed, l_res)
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(l_fed, l_res)), (l_fed, l_res)
    assert ms_fed <= 1.15 * ms_res, (ms_fed, ms_res, summ)


def test_train_step_with_physical_targets_in_file_order(P):
    """The reference's loop body as it is written (models/pangu_sample.py:52-71): targets arrive in PHYSICAL units and are normalised
    inside the step (`normData`, :57), and the reader has reversed the level axis (era5_data/utils_data.py:117).  train_step with
    `stats_last` + `levels_reversed` on fields in file order == train_step on fields flipped and normalised beforehand: the loss bit
    for bit (the fold changes no rounding), every parameter gradient to the run-to-run spread of the backward's atomics, and the
    updated parameters within one Adam step of each other."""
    from pangu_pytorch_amd import train
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = P.PanguModel(device=dev).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    g = lambda n, s, sc, sh: synth.uniform(s, synth.name_seed(n), sc, sh, device="cuda")
    tgt_phys, tgt_s_phys = g("tp", (1, 5, 13, 721, 1440), 40.0, 250.0), g("tsp", (1, 4, 721, 1440), 500.0, 1e5)
    sl = (g("slm", (1, 4, 1, 1), 300.0, 1e5), g("sls", (1, 4, 1, 1), 100.0, 700.0), g("ulm", (1, 5, 13, 1, 1), 20.0, 250.0),
          g("uls", (1, 5, 13, 1, 1), 5.0, 30.0))
    tgt_n, tgt_s_n = train.norm_data(tgt_phys, tgt_s_phys, sl)

    def run(batch, **kw):
        m.load_state_dict(state0)
        m.train()
        opt = train.make_optimizer(m)
        torch.manual_seed(5)
        loss = train.train_step(m, opt, batch, stats, maps, const_h, **kw)
        return loss.item(), [p.detach().clone() for p in m.parameters()], [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()]

    l_ref, p_ref, g_ref = run((inp, inp_s, tgt_n, tgt_s_n))
    l_new, p_new, g_new = run((inp.flip(-3).contiguous(), inp_s, tgt_phys.flip(-3).contiguous(), tgt_s_phys), stats_last=sl,
                              levels_reversed=True)
    assert l_new == l_ref, (l_new, l_ref)
    assert [a is None for a in g_new] == [b is None for b in g_ref]                    # the same DropPath draws: the same dropped branches
    worst = max(((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item() for a, b in zip(g_new, g_ref) if b is not None)
    assert worst < 1e-4, worst
    # (an Adam step moves every element by at most ~lr = 5e-6: a gradient element whose sign sits inside the atomics' noise may move the
    # other way)
    assert max(float((a - b).abs().max()) for a, b in zip(p_new, p_ref)) <= 2.5e-5

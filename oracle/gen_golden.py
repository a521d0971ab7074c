"""Generate golden vectors by running the REFERENCE itself (build container only; needs /root/reference).

TEST INFRASTRUCTURE. Usage:  python oracle/gen_golden.py [index] [blocks] [layers] [model] [model_bwd] [model_bwd_smooth] [extras] [rollout2] [rollout7] [keys_table] [refinit] [attn_windows] [model_bwd_input] [autocast_grads] [autocast_rollout]
Outputs small fixtures (fingerprints: samples + sums, index tensors, packed masks) under tests/golden/.
Inputs and parameters are closed-form (oracle/synth.py), so tests regenerate them bit-identically.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases   # noqa: E402
import ref_import   # noqa: E402
import synth   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(os.cpu_count())


def save(name, d):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in d.items()})
    print("wrote", name, len(d), "arrays")


def load_params(module, prefix, spec="golden"):
    """Overwrite every parameter of a reference module with synthetic values keyed by prefix+local name."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(synth.synth_param(prefix + k, tuple(p.shape), spec=spec))


def gen_index(L, M):
    out = {}
    # position_index (layers.py:319-357)
    att = L.EarthAttention3D(192, 6, 0, (2, 6, 12), device="cpu")
    pi = att.position_index
    assert pi.shape == (20736,) and int(pi.min()) == 0 and int(pi.max()) == 3311
    out["position_index"] = pi.to(torch.int16)
    meta = {"position_index_sha": hashlib.sha256(pi.numpy().astype(np.int64).tobytes()).hexdigest()[:16]}
    for C in (192, 384):
        st = cases.STAGES[C]
        Z, H, W = st["Z"], st["H"], 24
        blk = L.EarthSpecificBlock(C, 0.0, st["heads"], device="cpu")
        # mask (layers.py:153-181) on the padded+rolled frame
        x = torch.zeros(1, Z, H + 5, W, C)
        mask = blk.gen_mask(x)                                # (nLon, types, 144, 144)
        assert all(torch.equal(mask[0], mask[i]) for i in range(mask.shape[0]))
        assert set(mask.unique().tolist()) <= {0.0, -100.0}
        out[f"mask_bits_{C}"] = np.packbits((mask[0] != 0).numpy())
        meta[f"mask_shape_{C}"] = list(mask[0].shape)
        meta[f"mask_frac_{C}"] = float((mask[0] != 0).float().mean())
        # window gather index: push token ids through the block with attention replaced by a recorder
        N = Z * H * W
        ids = (torch.arange(N, dtype=torch.float32) + 1).view(1, N, 1).expand(1, N, C).contiguous()
        for roll in (False, True):
            rec = {}

            class Recorder(torch.nn.Module):
                def forward(self, xw, mask):
                    rec["xw"] = xw[..., 0].clone()
                    return xw
            blk.attention = Recorder()
            blk.norm1, blk.norm2 = torch.nn.Identity(), torch.nn.Identity()
            blk.linear = torch.nn.Identity()
            y = blk(ids, Z, H, W, roll)
            # with attention = identity the partition->reverse->crop round trip must be exact: y = x + (x + x)
            assert torch.equal(y, ids * 4), "round trip not exact"
            out[f"win_index_{C}_{int(roll)}"] = (rec["xw"].to(torch.int64) - 1).to(torch.int32)
    save("index.npz", out)
    with open(os.path.join(OUT, "index_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_blocks(L, M):
    for C in (192, 384):
        st = cases.STAGES[C]
        Z, H, W = st["Z"], st["H"], 24
        for roll in (False, True):
            torch.manual_seed(0)
            blk = L.EarthSpecificBlock(C, 0.1, st["heads"], device="cpu").eval()   # DropPath present, eval => identity
            pre = cases.block_prefix(C, roll)
            load_params(blk, pre)
            x = cases.block_input(C, W).requires_grad_(True)
            t = time.time()
            y = blk(x, Z, H, W, roll)
            tag = f"block_{C}_{int(roll)}"
            d = cases.summarize(y, tag + ".out")
            cot = cases.cotangent(tag, y.shape)
            (y * cot).sum().backward()
            d.update(cases.summarize(x.grad, tag + ".dx"))
            for k, p in blk.named_parameters():
                d.update(cases.summarize(p.grad, tag + ".d_" + k))
            print(tag, "ref fwd+bwd %.1fs" % (time.time() - t))
            save(tag + ".npz", d)


def gen_attn_windows(L, M):
    """EarthAttention3D.forward(x_window, mask) on its own (layers.py:360-421), both stages, mask None and mask = gen_mask."""
    d = {}
    for C in (192, 384):
        st = cases.STAGES[C]
        Z, H, W = st["Z"], st["H"], 24
        blk = L.EarthSpecificBlock(C, 0.0, st["heads"], device="cpu").eval()
        for roll in (False, True):
            load_params(blk, cases.block_prefix(C, roll))
            xw = cases.attention_window_input(C, W // 12)
            mask = blk.gen_mask(torch.zeros(1, Z, H + 5, W, C)) if roll else None
            with torch.no_grad():
                y = blk.attention(xw, mask)
            assert y.shape == xw.shape
            d.update(cases.summarize(y, f"attn_windows_{C}_{int(roll)}.out"))
    save("attn_windows.npz", d)


def gen_layers(L, M):
    d = {}
    # ---- PatchEmbedding_pretrain (layers.py:12-93), full resolution only
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    pe = L.PatchEmbedding_pretrain((2, 4, 4), 192)
    load_params(pe, "_input_layer.")
    t = time.time()
    x0 = pe(inp, inp_s, stats, maps, const_h)
    print("embed ref %.1fs" % (time.time() - t), x0.shape)
    d.update(cases.summarize(x0, "embed.out"))
    cot = cases.cotangent("embed", x0.shape)
    (x0 * cot).sum().backward()
    for k, p in pe.named_parameters():
        d.update(cases.summarize(p.grad, "embed.d_" + k))
    del x0, cot
    # ---- DownSample (layers.py:423-459)
    ds = L.DownSample(192)
    load_params(ds, "downsample.")
    xin = synth.uniform((1, 8 * 181 * 360, 192), synth.name_seed("down_in")).requires_grad_(True)
    y = ds(xin, 8, 181, 360)
    d.update(cases.summarize(y, "down.out"))
    (y * cases.cotangent("down", y.shape)).sum().backward()
    d.update(cases.summarize(xin.grad, "down.dx"))
    for k, p in ds.named_parameters():
        d.update(cases.summarize(p.grad, "down.d_" + k))
    # ---- UpSample (layers.py:461-499)
    us = L.UpSample(384, 192)
    load_params(us, "upsample.")
    xin = synth.uniform((1, 8 * 91 * 180, 384), synth.name_seed("up_in")).requires_grad_(True)
    y = us(xin)
    d.update(cases.summarize(y, "up.out"))
    (y * cases.cotangent("up", y.shape)).sum().backward()
    d.update(cases.summarize(xin.grad, "up.dx"))
    for k, p in us.named_parameters():
        d.update(cases.summarize(p.grad, "up.d_" + k))
    # ---- PatchRecovery_pretrain (layers.py:501-545)
    pr = L.PatchRecovery_pretrain(384)
    load_params(pr, "_output_layer.")
    xin = synth.uniform((1, 8 * 181 * 360, 384), synth.name_seed("recover_in")).requires_grad_(True)
    o, os_ = pr(xin, 8, 181, 360)
    d.update(cases.summarize(o, "recover.out"))
    d.update(cases.summarize(os_, "recover.out_surface"))
    ((o * cases.cotangent("recover", o.shape)).sum() + (os_ * cases.cotangent("recover_s", os_.shape)).sum()).backward()
    d.update(cases.summarize(xin.grad, "recover.dx"))
    for k, p in pr.named_parameters():
        d.update(cases.summarize(p.grad, "recover.d_" + k))
    save("layers_fullres.npz", d)


def build_model(M):
    torch.manual_seed(0)
    model = M.PanguModel(device="cpu")
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "keys_shapes.json"), "w") as f:
        json.dump({"state_dict": shapes,
                   "named_parameters_order": [k for k, _ in model.named_parameters()],
                   "n_params": sum(p.numel() for p in model.parameters())}, f, indent=0)
    # the reference's own key table (keys_all.csv column 1) must be the same set
    import csv
    with open(os.path.join(ref_import.REF_ROOT, "keys_all.csv")) as f:
        csv_keys = [r["torch_name"] for r in csv.DictReader(f) if r["torch_name"]]
    assert set(csv_keys) == set(shapes), "keys_all.csv != state_dict keys"
    assert len(csv_keys) == 223
    load_params(model, "")
    return model


def gen_extras(L, M):
    """Goldens for the SURVEY 8(f) widenings: device scores (reference era5_data/score.py, imported standalone) and the
    compact -> expanded bias gather (reference layers.py:384-391 run with the reference's own position_index)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_score", os.path.join(ref_import.REF_ROOT, "era5_data", "score.py"))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    d = {}
    pred = synth.uniform((2, 5, 721, 1440), synth.name_seed("score_pred"))
    tgt = pred * 0.7 + 0.5 * synth.uniform((2, 5, 721, 1440), synth.name_seed("score_tgt"))
    d["rmse"] = sc.weighted_rmse_torch_channels(pred, tgt)
    d["acc"] = sc.weighted_acc_torch_channels(pred, tgt)
    d["rmse_mean"] = sc.weighted_rmse_torch(pred, tgt)
    d["acc_mean"] = sc.weighted_acc_torch(pred, tgt)
    att = L.EarthAttention3D(384, 12, 0, (2, 6, 12), device="cpu")
    compact = synth.uniform((3312, 64, 12), synth.name_seed("compact_bias"), 0.5)
    eb = compact[att.position_index]                                   # layers.py:384
    eb = eb.view(144, 144, 64, 12)                                     # :388
    eb = torch.permute(eb, (2, 3, 0, 1)).unsqueeze(0)                  # :390-391
    d.update(cases.summarize(eb, "expanded_bias"))
    save("extras.npz", d)


def gen_model_smooth(L, M):
    """Whole-model backward under a SMOOTH loss (sum(out*cot)): gradients without the L1 loss' sign discontinuity,
    so they can be compared tightly."""
    model = build_model(M).eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    t = time.time()
    out, out_s = model(inp, inp_s, stats, maps, const_h)
    loss = ((out * cases.cotangent("model_out", out.shape)).sum() +
            (out_s * cases.cotangent("model_out_s", out_s.shape)).sum()) / out.numel()
    loss.backward()
    print("model smooth fwd+bwd ref %.1fs loss %.8f" % (time.time() - t, loss.item()))
    d = {"model.loss": torch.tensor([loss.item()], dtype=torch.float64)}
    for k, p in model.named_parameters():
        s = cases.summarize(p.grad, "model.d_" + k)
        d[f"model.d_{k}.samples"] = s[f"model.d_{k}.samples"][:256]
        d[f"model.d_{k}.abs_sum"] = s[f"model.d_{k}.abs_sum"]
    save("model_bwd_smooth.npz", d)


def gen_refinit(L, M):
    """Goldens in the REFERENCE's initialisation regime (synth.param_spec_refinit: weights std 0.02, LayerNorm (1, 0), zero
    biases -- models/pangu_model.py:41-48), for tight bf16 bounds:
      * every block variant at W = 24: fp32 output + dx fingerprints, AND the same block under the reference's own CPU
        autocast(bfloat16) -- its drift against fp32 is recorded (`.autocast_drift`, SURVEY App. B measured 3.75e-3): the HIP
        bf16 block must stay within 2x of it;
      * the whole model's smooth backward (sum(out * cot)): 223 gradient fingerprints + the loss."""
    d = {}
    for C in (192, 384):
        st = cases.STAGES[C]
        Z, H, W = st["Z"], st["H"], 24
        for roll in (False, True):
            torch.manual_seed(0)
            blk = L.EarthSpecificBlock(C, 0.1, st["heads"], device="cpu").eval()
            load_params(blk, cases.block_prefix(C, roll), spec="refinit")
            x = cases.block_input(C, W).requires_grad_(True)
            y = blk(x, Z, H, W, roll)
            tag = f"refinit_block_{C}_{int(roll)}"
            d.update(cases.summarize(y, tag + ".out"))
            cot = cases.cotangent(tag, y.shape)
            (y * cot).sum().backward()
            d.update(cases.summarize(x.grad, tag + ".dx"))
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                yb = blk(x.detach(), Z, H, W, roll)
            yb = yb.float()
            d[tag + ".autocast_drift"] = torch.tensor([((yb - y.detach()).norm() / y.detach().norm()).item(),
                                                       ((yb - y.detach()).abs().max() / y.detach().abs().max()).item()])
            d[tag + ".out_full"] = y.detach().to(torch.float16) if y.numel() * 2 < (3 << 20) else torch.zeros(1)
            print(tag, "reference autocast-bf16 drift (rel-L2, max-abs/max):", d[tag + ".autocast_drift"].tolist())
    torch.manual_seed(0)
    model = M.PanguModel(device="cpu")
    load_params(model, "", spec="refinit")
    model.eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    t = time.time()
    out, out_s = model(inp, inp_s, stats, maps, const_h)
    loss = ((out * cases.cotangent("model_out", out.shape)).sum() +
            (out_s * cases.cotangent("model_out_s", out_s.shape)).sum()) / out.numel()
    loss.backward()
    print("refinit model smooth fwd+bwd ref %.1fs loss %.8f" % (time.time() - t, loss.item()))
    d["model.loss"] = torch.tensor([loss.item()], dtype=torch.float64)
    d.update(cases.summarize(out, "model.out"))
    for k, p in model.named_parameters():
        s = cases.summarize(p.grad, "model.d_" + k)
        d[f"model.d_{k}.samples"] = s[f"model.d_{k}.samples"][:256]
        d[f"model.d_{k}.abs_sum"] = s[f"model.d_{k}.abs_sum"]
        d[f"model.d_{k}.l2"] = p.grad.double().norm().to(torch.float32).reshape(1)
    save("refinit.npz", d)


def stats_last_of(stats):
    """The `weatherStatistics_output` view of the input statistics (reference era5_data/utils_data.py:214-236):
    surface (1,4,1,1); upper (13,1,1,5) -> levels reversed -> (1,5,13,1,1)."""
    s_mean, s_std, u_mean, u_std = stats
    rev = lambda u: torch.from_numpy(np.transpose(u.numpy()[::-1].copy(), (1, 3, 0, 2)))[..., None].contiguous()
    return s_mean.view(1, 4, 1, 1), s_std.view(1, 4, 1, 1), rev(u_mean), rev(u_std)


def gen_rollout(L, M, steps=2, name="rollout2.npz"):
    """`steps` chained REFERENCE forwards: the loop of inference/inference_singleOutput.py:97-105 (output of one 24 h
    step = input of the next) with the torch model's normalised outputs taken back to physical units by normBackData
    (era5_data/utils_data.py:324-330; era5_data cannot be imported here, its 2-line body is restated)."""
    model = build_model(M).eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    s_mean, s_std, u_mean, u_std = stats_last_of(stats)
    d = {}
    up, sf = inp, inp_s
    for k in range(steps):
        t = time.time()
        with torch.no_grad():
            out, out_s = model(up, sf, stats, maps, const_h)
        print("rollout step %d ref %.1fs" % (k + 1, time.time() - t))
        d.update(cases.summarize(out, f"rollout.step{k + 1}.out"))
        d.update(cases.summarize(out_s, f"rollout.step{k + 1}.out_surface"))
        up = out * u_std + u_mean                    # normBackData, utils_data.py:327
        sf = out_s * s_std + s_mean                  # :328
    d.update(cases.summarize(up, "rollout.final_upper"))
    d.update(cases.summarize(sf, "rollout.final_surface"))
    save(name, d)


def gen_autocast(L, M, what=("rollout", "grads")):
    """The REFERENCE's own bf16: the reference run under `torch.autocast("cpu", dtype=torch.bfloat16)` -- the
    autocast its authors left commented out in models/pangu_sample.py:46-47 -- measured against the reference's fp32 goldens with
    the SAME error metrics the GPU tests apply to the HIP bf16 path, so that those tests bound "HIP bf16 vs reference fp32" by a
    multiple of "reference bf16 vs reference fp32" instead of by hand-picked constants.  Stored in tests/golden/autocast.npz:
      rollout.step{k}.err          cases.compare_summary of step k's normalised upper-air output of the seven-step autocast rollout
                                   (golden-spec weights, loop of gen_rollout) against rollout7.npz's fp32 fingerprints; step 1 is
                                   the whole forward on the golden inputs
      rollout.step{k}.err_surface  the same for the surface output
      grads.sample_err / grads.norm_err   per parameter (named_parameters order): rel-L2 over the 256 stored samples and relative
                                   norm error of the autocast smooth-loss gradients (refinit weights) against refinit.npz
      grads.out_err, grads.loss    output-sample rel-L2 and the loss of that run
    Commands:  python oracle/gen_golden.py autocast_rollout ; python oracle/gen_golden.py autocast_grads  (each merges into the file)"""
    path = os.path.join(OUT, "autocast.npz")
    d = dict(np.load(path)) if os.path.exists(path) else {}
    if "rollout" in what:
        g = np.load(os.path.join(OUT, "rollout7.npz"))
        model = build_model(M).eval()
        inp, inp_s, stats, maps, const_h = cases.model_inputs()
        s_mean, s_std, u_mean, u_std = stats_last_of(stats)
        up, sf = inp, inp_s
        for k in range(7):
            t = time.time()
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
                out, out_s = model(up, sf, stats, maps, const_h)
            out, out_s = out.float(), out_s.float()
            e = cases.compare_summary(out, g, f"rollout.step{k + 1}.out", 1.0)
            es = cases.compare_summary(out_s, g, f"rollout.step{k + 1}.out_surface", 1.0)
            print("autocast rollout step %d ref %.1fs fingerprint err vs fp32 reference: %.3e (surface %.3e)" % (k + 1, time.time() - t, e, es),
                  flush=True)
            d[f"rollout.step{k + 1}.err"] = np.array([e], dtype=np.float64)
            d[f"rollout.step{k + 1}.err_surface"] = np.array([es], dtype=np.float64)
            up = out * u_std + u_mean
            sf = out_s * s_std + s_mean
        del model
        np.savez_compressed(path, **d)
    if "grads" in what:
        g = np.load(os.path.join(OUT, "refinit.npz"))
        torch.manual_seed(0)
        model = M.PanguModel(device="cpu")
        load_params(model, "", spec="refinit")
        model.eval()
        inp, inp_s, stats, maps, const_h = cases.model_inputs()
        t = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out, out_s = model(inp, inp_s, stats, maps, const_h)
        out, out_s = out.float(), out_s.float()
        loss = ((out * cases.cotangent("model_out", out.shape)).sum() +
                (out_s * cases.cotangent("model_out_s", out_s.shape)).sum()) / out.numel()
        loss.backward()
        print("autocast refinit smooth fwd+bwd ref %.1fs loss %.8f (fp32 %.8f)" % (time.time() - t, loss.item(), float(g["model.loss"][0])),
              flush=True)
        se, ne = [], []
        for k, p in model.named_parameters():
            flat = p.grad.detach().float().flatten()
            pos = synth.sample_positions(flat.numel(), cases.NSAMP, synth.name_seed("pos_model.d_" + k))[:256]
            gs = torch.as_tensor(g[f"model.d_{k}.samples"]).double()
            se.append(((flat[pos].double() - gs).norm() / gs.norm().clamp_min(1e-30)).item())
            ne.append(abs(flat.double().norm().item() - float(g[f"model.d_{k}.l2"][0])) / float(g[f"model.d_{k}.l2"][0]))
        pos = synth.sample_positions(out.numel(), cases.NSAMP, synth.name_seed("pos_model.out"))
        go = torch.as_tensor(g["model.out.samples"]).double()
        d["grads.sample_err"] = np.array(se)
        d["grads.norm_err"] = np.array(ne)
        d["grads.out_err"] = np.array([((out.detach().flatten()[pos].double() - go).norm() / go.norm()).item()])
        d["grads.loss"] = np.array([loss.item()])
        print("autocast grads: worst sample err %.3e median %.3e worst norm err %.3e out err %.3e" %
              (max(se), sorted(se)[len(se) // 2], max(ne), float(d["grads.out_err"][0])), flush=True)
        np.savez_compressed(path, **d)
    print("wrote autocast.npz", sorted(d))


def gen_model_input_grads(L, M):
    """Gradients of the RAW FIELDS through the whole reference model (the reference's patch embedding is plain
    autograd, models/layers.py:40-93, so `input.requires_grad_()` yields input.grad): smooth loss sum(out * cot) / numel on the
    golden weights and inputs, fingerprints of d loss / d input and d loss / d input_surface -> model_bwd_input.npz."""
    model = build_model(M).eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    inp, inp_s = inp.requires_grad_(True), inp_s.requires_grad_(True)
    t = time.time()
    out, out_s = model(inp, inp_s, stats, maps, const_h)
    loss = ((out * cases.cotangent("model_out", out.shape)).sum() +
            (out_s * cases.cotangent("model_out_s", out_s.shape)).sum()) / out.numel()
    loss.backward()
    print("model input-gradient fwd+bwd ref %.1fs loss %.8f" % (time.time() - t, loss.item()))
    d = {"model.loss": torch.tensor([loss.item()], dtype=torch.float64)}
    d.update(cases.summarize(inp.grad, "model.d_input"))
    d.update(cases.summarize(inp_s.grad, "model.d_input_surface"))
    save("model_bwd_input.npz", d)


def gen_keys_table(L, M):
    """The reference's torch_name -> onnx_name table (keys_all.csv, the lookup of models/onnx2torch.py:23-36) as a data
    fixture: 223 name pairs, no weights."""
    import csv
    with open(os.path.join(ref_import.REF_ROOT, "keys_all.csv")) as f:
        table = {r["torch_name"]: r["onnx_name"] for r in csv.DictReader(f) if r["torch_name"]}
    assert len(table) == 223 and all(isinstance(v, str) and v for v in table.values())
    with open(os.path.join(OUT, "keys_table.json"), "w") as f:
        json.dump(table, f, indent=0)
    print("wrote keys_table.json", len(table))


def gen_model(L, M, backward):
    model = build_model(M).eval()
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    d = {}
    t = time.time()
    if not backward:
        with torch.no_grad():
            out, out_s = model(inp, inp_s, stats, maps, const_h)
        print("model fwd ref %.1fs" % (time.time() - t))
        d.update(cases.summarize(out, "model.out"))
        d.update(cases.summarize(out_s, "model.out_surface"))
        save("model_fwd.npz", d)
        return
    # training-step body, models/pangu_sample.py:52-71 in eval mode (DropPath off), checkpointing as shipped
    out, out_s = model(inp, inp_s, stats, maps, const_h)
    tgt, tgt_s = cases.model_targets()
    crit = torch.nn.L1Loss(reduction="none")
    uw = torch.tensor([3.00, 0.60, 1.50, 0.77, 0.54]).view(1, 5, 1, 1, 1)      # config.py:45, utils_data.py:297-300
    sw = torch.tensor([1.50, 0.77, 0.66, 3.00]).view(1, 4, 1, 1)
    loss = torch.mean(crit(out, tgt) * uw) + torch.mean(crit(out_s, tgt_s) * sw) * 0.25
    loss.backward()
    print("model fwd+bwd ref %.1fs loss %.8f" % (time.time() - t, loss.item()))
    d["model.loss"] = torch.tensor([loss.item()], dtype=torch.float64)
    for k, p in model.named_parameters():
        s = cases.summarize(p.grad, "model.d_" + k)
        d[f"model.d_{k}.samples"] = s[f"model.d_{k}.samples"][:256]
        d[f"model.d_{k}.abs_sum"] = s[f"model.d_{k}.abs_sum"]
    save("model_bwd.npz", d)


if __name__ == "__main__":
    assert ref_import.available(), "reference not mounted"
    L, M = ref_import.load()
    what = sys.argv[1:] or ["index", "blocks", "layers", "model"]
    if "index" in what:
        gen_index(L, M)
    if "blocks" in what:
        gen_blocks(L, M)
    if "layers" in what:
        gen_layers(L, M)
    if "model" in what:
        gen_model(L, M, backward=False)
    if "model_bwd" in what:
        gen_model(L, M, backward=True)
    if "extras" in what:
        gen_extras(L, M)
    if "model_bwd_smooth" in what:
        gen_model_smooth(L, M)
    if "keys_table" in what:
        gen_keys_table(L, M)
    if "rollout2" in what:
        gen_rollout(L, M, steps=2)
    if "rollout7" in what:      # BASELINE configs[4]: all seven 24 h steps of the week-long rollout, fingerprints per step
        gen_rollout(L, M, steps=7, name="rollout7.npz")
    if "refinit" in what:
        gen_refinit(L, M)
    if "attn_windows" in what:
        gen_attn_windows(L, M)
    if "model_bwd_input" in what:
        gen_model_input_grads(L, M)
    if "autocast_rollout" in what:
        gen_autocast(L, M, what=("rollout",))
    if "autocast_grads" in what:
        gen_autocast(L, M, what=("grads",))

"""Pin the CPU oracle (oracle/pangu_oracle.py) to golden vectors produced by the reference itself
(oracle/gen_golden.py, run in the build container against /root/reference)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import cases
import pangu_oracle as O
import synth

FP_TOL = 2e-5      # fp32 CPU restatement vs fp32 CPU reference: summation-order noise only


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def test_position_index_bit_exact(golden_dir):
    g = _load(golden_dir, "index.npz")
    mine = O.position_index()
    assert mine.dtype == torch.int64 and mine.shape == (20736,)
    assert np.array_equal(mine.numpy(), g["position_index"].astype(np.int64))
    meta = json.load(open(os.path.join(golden_dir, "index_meta.json")))
    assert hashlib.sha256(mine.numpy().tobytes()).hexdigest()[:16] == meta["position_index_sha"]
    assert meta["position_index_sha"] == "514371e088c3a008"        # SURVEY.md §7 probe value


@pytest.mark.parametrize("C", [192, 384])
def test_shift_mask_bit_exact(golden_dir, C):
    g = _load(golden_dir, "index.npz")
    st = cases.STAGES[C]
    m = O.shift_mask(st["Z"], st["H"], 24)
    assert m.shape == (st["types"], 144, 144)
    assert set(m.unique().tolist()) == {0.0, -100.0}
    assert np.array_equal(np.packbits((m != 0).numpy()), g[f"mask_bits_{C}"])
    # closed form == region-id derivation, for every longitude window
    long_way = O.shift_mask_region_ids(st["Z"], st["H"], 24)
    for l in range(long_way.shape[0]):
        assert torch.equal(long_way[l], m)


@pytest.mark.parametrize("C", [192, 384])
def test_shift_mask_full_size_sha(C):
    """sha256-16 of the full (nLon,types,144,144) fp32 mask measured on the reference."""
    st = cases.STAGES[C]
    W = 360 if C == 192 else 180
    m = O.shift_mask(st["Z"], st["H"], W)
    h = hashlib.sha256()
    for _ in range(W // 12):
        h.update(m.numpy().tobytes())
    assert h.hexdigest()[:16] == {192: "10f6f498518d3c73", 384: "7fe79f2147d4b92c"}[C]


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("roll", [False, True])
def test_window_index_bit_exact(golden_dir, C, roll):
    g = _load(golden_dir, "index.npz")
    st = cases.STAGES[C]
    idx = O.window_source_index(st["Z"], st["H"], 24, roll)
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.numpy(), g[f"win_index_{C}_{int(roll)}"])
    # every real token appears exactly once
    flat = idx[idx >= 0].long()
    assert flat.numel() == st["Z"] * st["H"] * 24 and flat.unique().numel() == flat.numel()


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("roll", [False, True])
def test_block_forward_backward(golden_dir, C, roll):
    tag = f"block_{C}_{int(roll)}"
    g = _load(golden_dir, tag + ".npz")
    st = cases.STAGES[C]
    p = {k: v.requires_grad_(True) for k, v in cases.block_params(C, roll).items()}
    x = cases.block_input(C, 24).requires_grad_(True)
    pre = cases.block_prefix(C, roll)
    y = O.earth_block(p, pre, x, st["Z"], st["H"], 24, st["heads"], roll)
    assert cases.compare_summary(y, g, tag + ".out", FP_TOL) < FP_TOL
    (y * cases.cotangent(tag, y.shape)).sum().backward()
    assert cases.compare_summary(x.grad, g, tag + ".dx", FP_TOL) < 5 * FP_TOL
    for k in cases.block_param_shapes(C):
        err = cases.compare_summary(p[pre + k].grad, g, tag + ".d_" + k, FP_TOL)
        assert err < 10 * FP_TOL, (k, err)


def test_state_dict_contract(golden_dir):
    path = os.path.join(golden_dir, "keys_shapes.json")
    if not os.path.exists(path):
        pytest.skip("keys_shapes.json not generated")
    ks = json.load(open(path))
    mine = cases.model_param_shapes()
    assert len(mine) == 223 and ks["n_params"] == 276659936
    assert {k: list(v) for k, v in mine.items()} == ks["state_dict"]


def _params(prefixes):
    return {k: synth.synth_param(k, sh) for k, sh in cases.model_param_shapes().items() if k.startswith(prefixes)}


def test_fullres_layers_forward_backward(golden_dir):
    """The oracle's patch_embed / down_sample / up_sample / patch_recover (the functions the GPU tests and smoke() lean on)
    against the REFERENCE's PatchEmbedding_pretrain / DownSample / UpSample / PatchRecovery_pretrain at full resolution
    (reference layers.py:12-93, :423-499, :501-545; tests/golden/layers_fullres.npz), forward and every parameter /
    input gradient."""
    g = _load(golden_dir, "layers_fullres.npz")
    T = 5 * FP_TOL
    # ---- patch embedding
    p = {k: v.requires_grad_(True) for k, v in _params(("_input_layer.",)).items()}
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    x0 = O.patch_embed(p, inp, inp_s, stats, maps, const_h)
    assert cases.compare_summary(x0, g, "embed.out", T) < T
    (x0 * cases.cotangent("embed", x0.shape)).sum().backward()
    for k in p:
        assert cases.compare_summary(p[k].grad, g, "embed.d_" + k[len("_input_layer."):], T) < 4 * T, k
    del x0, inp, inp_s, maps, const_h
    # ---- down-sample
    p = {k: v.requires_grad_(True) for k, v in _params(("downsample.",)).items()}
    xin = synth.uniform((1, 8 * 181 * 360, 192), synth.name_seed("down_in")).requires_grad_(True)
    y = O.down_sample(p, xin, 8, 181, 360)
    assert cases.compare_summary(y, g, "down.out", T) < T
    (y * cases.cotangent("down", y.shape)).sum().backward()
    assert cases.compare_summary(xin.grad, g, "down.dx", T) < 4 * T
    for k in p:
        assert cases.compare_summary(p[k].grad, g, "down.d_" + k[len("downsample."):], T) < 4 * T, k
    # ---- up-sample
    p = {k: v.requires_grad_(True) for k, v in _params(("upsample.",)).items()}
    xin = synth.uniform((1, 8 * 91 * 180, 384), synth.name_seed("up_in")).requires_grad_(True)
    y = O.up_sample(p, xin, 8, 91, 180, 181)
    assert cases.compare_summary(y, g, "up.out", T) < T
    (y * cases.cotangent("up", y.shape)).sum().backward()
    assert cases.compare_summary(xin.grad, g, "up.dx", T) < 4 * T
    for k in p:
        assert cases.compare_summary(p[k].grad, g, "up.d_" + k[len("upsample."):], T) < 4 * T, k
    # ---- patch recovery
    p = {k: v.requires_grad_(True) for k, v in _params(("_output_layer.",)).items()}
    xin = synth.uniform((1, 8 * 181 * 360, 384), synth.name_seed("recover_in")).requires_grad_(True)
    o, os_ = O.patch_recover(p, xin, 8, 181, 360)
    assert cases.compare_summary(o, g, "recover.out", T) < T
    assert cases.compare_summary(os_, g, "recover.out_surface", T) < T
    ((o * cases.cotangent("recover", o.shape)).sum() + (os_ * cases.cotangent("recover_s", os_.shape)).sum()).backward()
    assert cases.compare_summary(xin.grad, g, "recover.dx", T) < 4 * T
    for k in p:
        assert cases.compare_summary(p[k].grad, g, "recover.d_" + k[len("_output_layer."):], T) < 4 * T, k


@pytest.mark.skipif(os.environ.get("PANGU_SKIP_SLOW") == "1", reason="whole-model CPU forward (~90 s on 8 cores)")
def test_oracle_forward_and_loss_vs_reference(golden_dir):
    """O.forward / O.train_loss against the reference's whole forward and its training loss on the same synthetic sample
    (tests/golden/model_fwd.npz, model_bwd.npz['model.loss'])."""
    g = _load(golden_dir, "model_fwd.npz")
    p = {k: synth.synth_param(k, sh) for k, sh in cases.model_param_shapes().items()}
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        out, out_s = O.forward(p, inp, inp_s, stats, maps, const_h)
    assert cases.compare_summary(out, g, "model.out", 1e-4) < 1e-4
    assert cases.compare_summary(out_s, g, "model.out_surface", 1e-4) < 1e-4
    gb = _load(golden_dir, "model_bwd.npz")
    tgt, tgt_s = cases.model_targets()
    loss = O.train_loss(out, out_s, tgt, tgt_s).item()
    assert abs(loss - float(gb["model.loss"][0])) < 1e-5 * float(gb["model.loss"][0])


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("roll", [False, True])
def test_attention_windows_golden(golden_dir, C, roll):
    """EarthAttention3D.forward(x_window, mask) on its own (reference layers.py:360-421), every slot non-zero."""
    g = _load(golden_dir, "attn_windows.npz")
    st = cases.STAGES[C]
    pre = cases.block_prefix(C, roll)
    p = cases.block_params(C, roll)
    xw = cases.attention_window_input(C, 2)
    mask = O.shift_mask(st["Z"], st["H"], 24).unsqueeze(0).expand(2, -1, -1, -1) if roll else None
    y = O.attention_windows(xw, p[pre + "attention.linear1.weight"], p[pre + "attention.linear1.bias"],
                            p[pre + "attention.linear2.weight"], p[pre + "attention.linear2.bias"],
                            p[pre + "attention.earth_specific_bias"], mask)
    assert cases.compare_summary(y, g, f"attn_windows_{C}_{int(roll)}.out", FP_TOL) < FP_TOL

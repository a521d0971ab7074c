"""Weight import + compact bias table and the oracle's score restatement (8(f)-3), CPU side."""
import io
import os

import numpy as np
import pytest
import torch

import cases
import pangu_oracle as O
import synth
import pangu_pytorch_amd as P
from pangu_pytorch_amd import weights as Wt


def test_position_index_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "index.npz"))
    mine = Wt.position_index()
    assert np.array_equal(mine.numpy(), g["position_index"].astype(np.int64))
    assert torch.equal(mine, O.position_index())
    assert int(mine.max()) == 3311 and mine.unique().numel() == 3312


def test_attention_module_position_index_attribute(golden_dir):
    """The reference's EarthAttention3D.position_index (layers.py:319-357): a plain attribute of every attention module,
    not part of the state_dict."""
    import pangu_pytorch_amd as P
    g = np.load(os.path.join(golden_dir, "index.npz"))
    for dim, heads in ((192, 6), (384, 12)):
        att = P.layers.EarthAttention3D(dim, heads, 0, (2, 6, 12))
        assert att.position_index.dtype == torch.int64 and att.position_index.shape == (20736,)
        assert np.array_equal(att.position_index.numpy(), g["position_index"].astype(np.int64))
        assert "position_index" not in att.state_dict()


def test_expand_bias_matches_reference_gather(golden_dir):
    g = np.load(os.path.join(golden_dir, "extras.npz"))
    compact = synth.uniform((3312, 64, 12), synth.name_seed("compact_bias"), 0.5)
    e = Wt.expand_bias(compact)
    assert e.shape == (1, 64, 12, 144, 144)
    assert cases.compare_summary(e, g, "expanded_bias", 0) == 0.0             # pure gather: bit-exact
    assert torch.equal(e, O.expand_bias(compact))
    back = Wt.compact_bias(e)
    assert torch.allclose(back, compact, rtol=0, atol=1e-6)                     # exact inverse up to the mean's rounding


def test_oracle_scores_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "extras.npz"))
    pred = synth.uniform((2, 5, 721, 1440), synth.name_seed("score_pred"))
    tgt = pred * 0.7 + 0.5 * synth.uniform((2, 5, 721, 1440), synth.name_seed("score_tgt"))
    assert np.allclose(O.weighted_rmse_channels(pred, tgt).numpy(), g["rmse"], rtol=1e-5)
    assert np.allclose(O.weighted_acc_channels(pred, tgt).numpy(), g["acc"], rtol=1e-5)


def test_load_checkpoint_formats(tmp_path):
    blk = P.layers.EarthSpecificBlock(192, 0.0, 6)
    sd = {k: torch.randn_like(v) for k, v in blk.state_dict().items()}
    Wt.load_checkpoint(blk, {"model": sd, "epoch": 3})                          # reference finetune_fully.py:115-116
    assert torch.equal(blk.norm1.weight, sd["norm1.weight"])
    path = tmp_path / "ckpt.pth"
    torch.save({"model": {k: v * 2 for k, v in sd.items()}}, path)
    Wt.load_checkpoint(blk, str(path))
    assert torch.equal(blk.norm1.weight, sd["norm1.weight"] * 2)
    with pytest.raises(RuntimeError):
        Wt.load_checkpoint(blk, {"model": {"norm1.weight": sd["norm1.weight"]}})   # strict


def test_load_onnx_initializers_semantics():
    """2-D MatMul initialisers are stored (in,out) and must be transposed; others copied (reference onnx2torch.py:36-52)."""
    m = P.layers.DownSample(192)
    table = {"linear.weight": "onnx::MatMul_1", "norm.weight": "b.norm.weight", "norm.bias": "b.norm.bias"}
    w_onnx = np.random.RandomState(0).randn(768, 384).astype(np.float32)             # (in, out)
    ow = {"onnx::MatMul_1": w_onnx, "b.norm.weight": np.full(768, 2.0, np.float32), "b.norm.bias": np.zeros(768, np.float32)}
    missing = Wt.load_onnx_initializers(m, ow, table, freeze=True)
    assert missing == []
    assert torch.equal(m.linear.weight, torch.from_numpy(w_onnx).t()) and not m.linear.weight.requires_grad
    assert float(m.norm.weight[0]) == 2.0
    with pytest.raises(ValueError):
        Wt.load_onnx_initializers(m, {**ow, "onnx::MatMul_1": w_onnx.T.copy()}, table)
    assert Wt.load_onnx_initializers(m, ow, {"linear.weight": "onnx::MatMul_1"}) == ["norm.weight", "norm.bias"]


def test_load_onnx_initializers_full_key_table(golden_dir):
    """All 223 rows of the reference's torch_name -> onnx_name table (keys_all.csv as tests/golden/keys_table.json):
    initialisers named and laid out the ONNX way (2-D MatMul weights (in, out), everything else as stored) land in the
    right parameters of the whole model (reference models/onnx2torch.py:23-52)."""
    import json
    table = json.load(open(os.path.join(golden_dir, "keys_table.json")))
    shapes = cases.model_param_shapes()
    assert set(table) == set(shapes) and len(set(table.values())) == 223
    m = P.PanguModel(depths=[2, 6, 6, 2])
    want = synth.fill_state_dict(shapes)
    onnx_weights = {table[k]: (v.t().contiguous() if v.dim() == 2 else v).numpy() for k, v in want.items()}
    assert sum(1 for k, v in want.items() if v.dim() == 2) == 67        # the transposed ones: every nn.Linear weight
    missing = Wt.load_onnx_initializers(m, onnx_weights, table, freeze=True)
    assert missing == []
    for k, p in m.named_parameters():
        assert torch.equal(p, want[k]), k
        assert not p.requires_grad


def test_compact_bias_table_roundtrip_and_refusal():
    """weights.compact_bias_table: an expansion of a compact table folds back to exactly that table (index axis last); an
    expanded tensor that is NOT an expansion (the reference's random init of the expanded parameter, layers.py:306-314) is
    refused -- the compact inference mode never approximates."""
    from pangu_pytorch_amd import weights
    g = torch.Generator().manual_seed(5)
    compact = torch.randn(3312, 7, 3, generator=g)
    expanded = weights.expand_bias(compact)
    table = weights.compact_bias_table(expanded)
    assert table.shape == (7, 3, 3312) and torch.equal(table, compact.permute(1, 2, 0))
    bad = expanded.clone()
    bad[0, 2, 1, 17, 5] += 1e-3                   # one of several entries sharing an index
    assert weights.compact_bias_table(bad) is None
    assert weights.compact_bias_table(torch.randn(1, 7, 3, 144, 144, generator=g)) is None


def test_param_stamp_changes_on_any_optimizer_step():
    """ops.param_stamp carries a process-wide optimizer epoch: torch's fused optimizers update parameters without bumping
    `_version`, so the derived copies (bf16 shadows, compact bias tables) are invalidated by the global post-step hook."""
    import torch
    from pangu_pytorch_amd import ops
    p = torch.nn.Parameter(torch.zeros(3))
    other = torch.nn.Parameter(torch.ones(2))
    s0 = ops.param_stamp(p)
    assert ops.param_stamp(p) == s0
    other.grad = torch.ones(2)
    torch.optim.SGD([other], lr=0.1).step()
    s1 = ops.param_stamp(p)
    assert s1 != s0 and s1[1:] == s0[1:]                 # only the epoch moved: p itself was not touched
    ops.bump_weights_epoch()
    assert ops.param_stamp(p) != s1

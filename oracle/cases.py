"""Shared definitions of the golden-vector cases (TEST INFRASTRUCTURE).

Used by oracle/gen_golden.py (reference side, build container only) and by tests/ (oracle side and
HIP side) so that both regenerate the same closed-form inputs.
"""
import math
import torch
import synth

NSAMP = 4096
STAGES = {192: dict(Z=8, H=181, heads=6, types=124, layer=0), 384: dict(Z=8, H=91, heads=12, types=64, layer=1)}


def block_prefix(C, roll):
    return f"layers.EarthSpecificLayer{STAGES[C]['layer']}.blocks.EarthSpecificBlock{1 if roll else 0}."


def block_param_shapes(C):
    t, h = STAGES[C]["types"], STAGES[C]["heads"]
    return {
        "norm1.weight": (C,), "norm1.bias": (C,), "norm2.weight": (C,), "norm2.bias": (C,),
        "linear.linear1.weight": (4 * C, C), "linear.linear1.bias": (4 * C,),
        "linear.linear2.weight": (C, 4 * C), "linear.linear2.bias": (C,),
        "attention.earth_specific_bias": (1, t, h, 144, 144),
        "attention.linear1.weight": (3 * C, C), "attention.linear1.bias": (3 * C,),
        "attention.linear2.weight": (C, C), "attention.linear2.bias": (C,),
    }


def block_params(C, roll, device="cpu"):
    pre = block_prefix(C, roll)
    return {pre + k: synth.synth_param(pre + k, s, device) for k, s in block_param_shapes(C).items()}


def block_input(C, W, device="cpu"):
    st = STAGES[C]
    N = st["Z"] * st["H"] * W
    return synth.uniform((1, N, C), synth.name_seed(f"block_input_{C}_{W}"), device=device)


def attention_window_input(C, nLon, device="cpu"):
    """A partitioned tensor (nLon, types, 144, C) for EarthAttention3D.forward taken on its own: EVERY slot non-zero (also the
    slots that hold the block's zero-pad rows when the module runs inside a block)."""
    return synth.uniform((nLon, STAGES[C]["types"], 144, C), synth.name_seed(f"attn_windows_{C}_{nLon}"), device=device)


def cotangent(name, shape, device="cpu"):
    """Fixed upstream gradient so that loss = sum(out * cotangent)."""
    return synth.uniform(shape, synth.name_seed("cot_" + name), device=device)


def model_param_shapes(depths=(2, 6, 6, 2), dims=(192, 384, 384, 192)):
    """223 (name -> shape) in the reference's named_parameters order (checked against golden keys_shapes.json)."""
    out = {
        "_input_layer.conv.weight": (192, 192, 1), "_input_layer.conv.bias": (192,),
        "_input_layer.conv_surface.weight": (192, 112, 1), "_input_layer.conv_surface.bias": (192,),
        "downsample.linear.weight": (384, 768), "downsample.norm.weight": (768,), "downsample.norm.bias": (768,),
    }
    for li, (d, C) in enumerate(zip(depths, dims)):
        for bi in range(d):
            pre = f"layers.EarthSpecificLayer{li}.blocks.EarthSpecificBlock{bi}."
            for k, s in block_param_shapes(C).items():
                out[pre + k] = s
    out.update({
        "upsample.linear1.weight": (768, 384), "upsample.linear2.weight": (192, 192),
        "upsample.norm.weight": (192,), "upsample.norm.bias": (192,),
        "_output_layer.conv.weight": (160, 384, 1), "_output_layer.conv.bias": (160,),
        "_output_layer.conv_surface.weight": (64, 384, 1), "_output_layer.conv_surface.bias": (64,),
    })
    return out


def model_inputs(device="cpu", B=1):
    """Synthetic full-resolution inputs + NON-trivial statistics (so normalisation and level reversal are exercised)."""
    g = lambda n, s, sc=1.0, sh=0.0: synth.uniform(s, synth.name_seed(n), sc, sh, device=device)
    inp = g("input", (B, 5, 13, 721, 1440))
    inp_s = g("input_surface", (B, 4, 721, 1440))
    stats = (g("surface_mean", (4,), 0.3), g("surface_std", (4,), 0.2, 1.2),
             g("upper_mean", (13, 1, 1, 5), 0.3), g("upper_std", (13, 1, 1, 5), 0.2, 1.2))
    maps = g("maps", (1, 3, 724, 1440))
    const_h = g("const_h", (1, 1, 1, 13, 721, 1440))
    return inp, inp_s, stats, maps, const_h


def model_targets(device="cpu", B=1):
    g = lambda n, s: synth.uniform(s, synth.name_seed(n), device=device)
    return g("target", (B, 5, 13, 721, 1440)), g("target_surface", (B, 4, 721, 1440))


def summarize(t, name):
    """Small fingerprint of a tensor: NSAMP pseudo-random samples, last-dim sums, total |.| sum."""
    t = t.detach().to(torch.float32).contiguous()
    pos = synth.sample_positions(t.numel(), NSAMP, synth.name_seed("pos_" + name), device=t.device)
    flat = t.flatten()
    return {
        name + ".samples": flat[pos].cpu(),
        name + ".lastdim_sum": t.reshape(-1, t.shape[-1]).to(torch.float64).sum(0).to(torch.float32).cpu(),
        name + ".abs_sum": flat.to(torch.float64).abs().sum().to(torch.float32).reshape(1).cpu(),
    }


def compare_summary(t, golden, name, rtol):
    """Max relative error (vs max |golden| of each fingerprint part) of tensor t against stored fingerprint."""
    mine = summarize(t, name)
    worst = 0.0
    for k, v in mine.items():
        g = torch.as_tensor(golden[k]).to(torch.float32)
        denom = g.abs().max().clamp_min(1e-30)
        if k.endswith(".lastdim_sum"):
            # column sums may cancel exactly (e.g. anything downstream of a LayerNorm backward):
            # measure the error against the column's absolute mass instead
            mass = torch.as_tensor(golden[name + ".abs_sum"]).to(torch.float32)[0] / g.numel()
            denom = torch.maximum(denom, mass)
        err = ((v - g).abs().max() / denom).item()
        worst = max(worst, err)
    return worst

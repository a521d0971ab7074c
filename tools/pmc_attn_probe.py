#!/usr/bin/env python3
"""Run the window-attention kernels a few times (for rocprofv3 --pmc runs): python tools/pmc_attn_probe.py [f32|bf16] [bwd]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pangu_pytorch_amd as P
from pangu_pytorch_amd import ops, ops_bf16 as ob
bf = len(sys.argv) > 1 and sys.argv[1] == "bf16"
bwd = "bwd" in sys.argv[1:]
dt = torch.bfloat16 if bf else torch.float32
for (Z, H, W, C, heads, types) in ((8, 181, 360, 192, 6, 124), (8, 91, 180, 384, 12, 64)):
    N = Z * H * W
    qkv = torch.randn(N, 3 * C, device="cuda").to(dt)
    qb = torch.randn(3 * C, device="cuda").to(dt)
    esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.02).to(dt)
    mod = ob if bf else ops
    for shifted in (False, True):
        if bwd:
            dout = torch.randn(N, C, device="cuda").to(dt)
            out, lse = mod.window_attention(qkv, qb, esb, Z, H, W, heads, shifted, want_lse=True)
        for _ in range(3):
            if bwd:
                mod.window_attention_bwd(qkv, qb, esb, out, lse, dout, Z, H, W, heads, shifted)
            else:
                mod.window_attention(qkv, qb, esb, Z, H, W, heads, shifted)
torch.cuda.synchronize()

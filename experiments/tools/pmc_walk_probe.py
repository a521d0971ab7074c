#!/usr/bin/env python3
"""Run the fused QKV attention in its two forms on the model's two shapes (for rocprofv3 --pmc passes):
python experiments/tools/pmc_walk_probe.py <variant,variant,..> [reps]   -- variant 0 = the (window, head) kernel, 40 / 21 / .. = attn_walk_bf16.hip"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "experiments"))
import exp_ops as E
import torch
from pangu_pytorch_amd import ops_bf16 as ob
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,40,21").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bf = torch.bfloat16
torch.manual_seed(0)
for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
    N = Z * H * W
    x = torch.randn(N, C, device="cuda").to(bf)
    w = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(bf)
    b = torch.randn(3 * C, device="cuda")
    esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(bf)
    for sh in (False, True):
        for v in variants:
            for _ in range(reps):
                if v:
                    E.window_attention_qkv_walk(x, w, b, esb, Z, H, W, heads, sh, variant=v)
                else:
                    ob.window_attention_qkv(x, w, b, esb, Z, H, W, heads, sh)
torch.cuda.synchronize()

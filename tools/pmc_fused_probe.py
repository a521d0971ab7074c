#!/usr/bin/env python3
"""Run the fused bf16 inference kernels of one block a few times at the model's two shapes (for rocprofv3 --pmc passes; the library
under test is chosen by PANGU_HIP_LIB):   python tools/pmc_fused_probe.py [attn_qkv] [gemm_ln] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pangu_pytorch_amd import ops_bf16 as ob
what = [a for a in sys.argv[1:] if not a.isdigit()] or ["attn_qkv", "gemm_ln"]
reps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
bf = torch.bfloat16
torch.manual_seed(0)
for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
    N = Z * H * W
    x = torch.randn(N, C, device="cuda").to(bf)
    if "attn_qkv" in what:
        w = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(bf)
        b = torch.randn(3 * C, device="cuda")
        esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(bf)
        for sh in (False, True):
            for _ in range(reps):
                ob.window_attention_qkv(x, w, b, esb, Z, H, W, heads, sh)
    if "gemm_ln" in what:
        w2 = (torch.randn(C, C, device="cuda") / C ** 0.5).to(bf)
        b2, g, be = torch.randn(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        sc = torch.randn(N, C, device="cuda").to(bf)
        for _ in range(reps):
            ob.linear_ln_residual(x, w2, b2, sc, g, be)
torch.cuda.synchronize()

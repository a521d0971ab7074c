#!/usr/bin/env python3
"""Run the fused MLP kernel on the two model shapes a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pangu_pytorch_amd import ops_bf16 as ob
bf = torch.bfloat16
for M, C in ((521280, 192), (131040, 384)):
    x = torch.randn(M, C, device="cuda").to(bf)
    w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(bf)
    w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(bf)
    b1, b2, g, be = (torch.randn(n, device="cuda") for n in (4 * C, C, C, C))
    img = ob.pack_mlp_weights(w1, w2)
    out = torch.empty_like(x)
    for _ in range(4):
        ob.mlp_ln_residual(x, img, b1, b2, g, be, out=out)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Run a few GEMM shapes back to back (for rocprofv3 --pmc runs):
   python tools/pmc_gemm_probe.py [f32|bf16] [M,N,K ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pangu_pytorch_amd as P
from pangu_pytorch_amd import ops, ops_bf16 as ob
bf = len(sys.argv) > 1 and sys.argv[1] == "bf16"
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or \
    [(521280, 576, 192), (521280, 192, 768), (131040, 1152, 384), (131040, 384, 1536)]
for M, N, K in shapes:
    dt = torch.bfloat16 if bf else torch.float32
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        (ob.linear if bf else ops.linear)(a, w, b)
torch.cuda.synchronize()

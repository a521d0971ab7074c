#!/usr/bin/env python3
"""The one-launch fp32 MLP branch (experiments/csrc/mlp_fused_f32.hip) vs the two launches of the product, interleaved:
python experiments/tools/bench_mlp_f32.py        (PANGU_EXP_LIB=<path> selects an ablation build of the experiments library)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "experiments"))
import torch
import exp_ops as E
from pangu_pytorch_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


M, C = 521280, 192
x = torch.randn(M, C, device="cuda")
w1, b1 = torch.randn(4 * C, C, device="cuda") / C ** 0.5, torch.randn(4 * C, device="cuda")
w2, b2 = torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5, torch.randn(C, device="cuda")
g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
out = torch.empty((M, C), device="cuda")
hbuf = torch.empty((M, 4 * C), device="cuda")
fl = 16.0 * M * C * C
for rnd in range(3):
    ms = timeit(lambda: E.mlp_ln_residual_f32(x, w1, b1, w2, b2, g, be, out=out))
    ms2 = timeit(lambda: ops.linear_ln_residual(ops.linear(x, w1, b1, act=ops.ACT_GELU, out=hbuf), w2, b2, x, g, be, out=out))
    print(f"mlp_f32 s0 M={M} C={C}: fused {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF/s ({fl / ms / 1e9 / 157.3:.3f} of peak)   | "
          f"two launches {ms2:7.3f} ms {fl / ms2 / 1e9:6.1f} TF/s")

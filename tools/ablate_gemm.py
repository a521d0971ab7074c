#!/usr/bin/env python3
"""Timing-only ablation of the bf16 projection GEMM (development tool): loads every scratch/libgemm_<tag>.so (the whole library
with csrc/gemm_bf16.hip built under -DGEMM_ABLATE=<n>: 1 no in-loop LDS-DMA requests, 2 and no barrier, 3 and no fragment reads,
4 everything but the epilogue, 5 = 1 without the epilogue) and times the model's shapes through the C-ABI, interleaved rounds."""
import ctypes
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int
libs = {}
for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "libgemm_*.so"))):
    tag = re.search(r"libgemm_(\w+)\.so", path).group(1)
    lib = ctypes.CDLL(path)
    lib.pangu_linear_fwd_bf16.argtypes = [P, P, I, P, P, P, I, I, I, I, I, P, I]
    libs[tag] = lib
bf = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream
# (name, M, N, K, act): act 0 = bias only, 3 = + aux (residual-gradient accumulation)
SHAPES = (("s1 qkv", 131040, 1152, 384, 0), ("s1 proj", 131040, 384, 384, 0), ("s1 dqkv+add", 131040, 384, 1152, 3),
          ("s1 dfc1", 131040, 384, 1536, 0), ("s0 qkv", 521280, 576, 192, 0), ("s0 dqkv+add", 521280, 192, 576, 3),
          ("up1", 131040, 768, 384, 0))
for name, M, N, K, act in SHAPES:
    a = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    aux = torch.randn(M, N, device="cuda").to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf)

    def run(lib):
        rc = lib.pangu_linear_fwd_bf16(stream, a.data_ptr(), K, w.data_ptr(), bias.data_ptr(), c.data_ptr(), N, M, N, K, act,
                                       aux.data_ptr() if act else None, 1)
        assert rc == 0, rc

    times = {t: [] for t in libs}
    for rnd in range(6):
        for tag, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(lib)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[tag].append(e0.elapsed_time(e1) / 5)
    line = f"{name:12s} M={M} N={N} K={K}: "
    for tag in libs:
        t = sorted(times[tag])
        line += f" {tag} {t[len(t) // 2] * 1e3:.0f} us"
    print(line + f"   (MFMA at peak {2.0 * M * N * K / 2.5e15 * 1e6:.0f} us, bytes at 5 TB/s {(M * K + M * N * (2 if act else 1)) * 2 / 5e12 * 1e6:.0f} us)")

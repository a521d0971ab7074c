#!/usr/bin/env python3
"""Timing-only ablation of the fused MLP kernel (development tool): loads every scratch/libmlp_<tag>.so (built from
csrc/mlp_fused_bf16.hip with -DPANGU_MLP_ABLATE=<mask>: 1 no in-loop weight requests, 2 no GELU, 4 no first product,
8 no second product, 16 no epilogue) and times the two model shapes, interleaved rounds in one process."""
import ctypes
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pangu_pytorch_amd import ops_bf16 as ob  # noqa: E402

P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
libs = {}
for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "libmlp_*.so"))):
    tag = re.search(r"libmlp_(\w+)\.so", path).group(1)
    lib = ctypes.CDLL(path)
    lib.pangu_mlp_ln_residual_fwd_bf16.argtypes = [P, P, I, P, P, P, P, P, P, I, I, I, F]
    libs[tag] = lib
bf = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream
for M, C in ((521280, 192), (131040, 384)):
    x = torch.randn(M, C, device="cuda").to(bf)
    w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(bf)
    w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(bf)
    b1, b2, g, be = (torch.randn(n, device="cuda") for n in (4 * C, C, C, C))
    img = ob.pack_mlp_weights(w1, w2)
    out = torch.empty_like(x)
    times = {t: [] for t in libs}
    for rnd in range(6):
        for tag, lib in libs.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                rc = lib.pangu_mlp_ln_residual_fwd_bf16(stream, x.data_ptr(), C, img.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                                                        g.data_ptr(), be.data_ptr(), out.data_ptr(), C, M, C, 1.0)
                assert rc == 0, rc
            b.record()
            torch.cuda.synchronize()
            if rnd:
                times[tag].append(a.elapsed_time(b) / 5)
    for tag in libs:
        t = sorted(times[tag])
        print(f"C={C} M={M} {tag:>8s}: median {t[len(t) // 2]:.3f} ms  min {t[0]:.3f} ms")
        if hasattr(libs[tag], "pangu_mlp_stamp_read"):
            buf = (ctypes.c_ulonglong * 8)()
            libs[tag].pangu_mlp_stamp_read(buf)          # clear what the timing rounds accumulated
            libs[tag].pangu_mlp_ln_residual_fwd_bf16(stream, x.data_ptr(), C, img.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                                                     g.data_ptr(), be.data_ptr(), out.data_ptr(), C, M, C, 1.0)
            libs[tag].pangu_mlp_stamp_read(buf)
            v = list(buf)
            n = max(v[3], 1)
            print(f"   stamps per steady iteration (cycles): sync {v[0] / n:.0f}  second product {v[1] / n:.0f}  first product {v[2] / n:.0f}"
                  f"  | whole kernel per wave {v[4] / max(v[5], 1):.0f} cycles ({v[5]} waves, {v[3] / max(v[5], 1):.0f} steady iterations each): "
                  f"prologue {v[6] / max(v[5], 1):.0f}, epilogue {v[7] / max(v[5], 1):.0f}")

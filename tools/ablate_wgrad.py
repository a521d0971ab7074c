#!/usr/bin/env python3
"""Timing + in-kernel stamps of the bf16 weight-gradient kernel (development tool): loads every scratch/libwg_<tag>.so (the
whole library with csrc/wgrad_bf16_dma.hip built under other -D switches; -DPANGU_WGRAD_STAMP adds per-wave s_memtime sums)
and times the model's shapes through the C-ABI, interleaved rounds in one process."""
import ctypes
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
libs = {}
for path in sorted(glob.glob(os.path.join(ROOT, "scratch", "libwg_*.so"))):
    tag = re.search(r"libwg_(\w+)\.so", path).group(1)
    lib = ctypes.CDLL(path)
    lib.pangu_linear_wgrad_bf16_ws.argtypes = [P, P, I, P, I, P, P, I, I, I, P, L]
    libs[tag] = lib
bf = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream
WS = 96 << 20
ws = torch.empty(WS // 4, device="cuda")
SHAPES = (("s0 proj", 521280, 192, 192), ("s1 proj", 131040, 384, 384), ("down", 131040, 384, 768), ("up1", 131040, 768, 384),
          ("embed", 456120, 192, 192), ("recover", 456120, 160, 384), ("s0 qkv", 521280, 576, 192), ("s0 mlp1", 521280, 768, 192), ("s0 mlp2", 521280, 192, 768),
          ("s1 qkv", 131040, 1152, 384), ("s1 mlp1", 131040, 1536, 384), ("s1 mlp2", 131040, 384, 1536))
for name, M, N, K in SHAPES:
    dc = torch.randn(M, N, device="cuda").to(bf)
    a = torch.randn(M, K, device="cuda").to(bf)
    dw = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")

    def run(lib):
        rc = lib.pangu_linear_wgrad_bf16_ws(stream, dc.data_ptr(), N, a.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K,
                                            ws.data_ptr(), WS)
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
    for tag, lib in libs.items():
        t = sorted(times[tag])
        print(f"{name:8s} M={M} N={N} K={K} {tag:>10s}: median {t[len(t) // 2]:.3f} ms  min {t[0]:.3f} ms   {2.0 * M * N * K / t[len(t) // 2] / 1e9:.0f} TF/s")
        if hasattr(lib, "pangu_wgrad_stamp_read"):
            buf = (ctypes.c_ulonglong * 8)()
            lib.pangu_wgrad_stamp_read(buf)
            run(lib)
            lib.pangu_wgrad_stamp_read(buf)
            v = list(buf)
            n = max(v[6], 1)
            print(f"   cycles per K-step and wave: own-DMA wait {v[0] / n:.0f}  barrier {v[1] / n:.0f}  DMA issue {v[2] / n:.0f}  first fragments "
                  f"{v[3] / n:.0f}  MFMA loop {v[4] / n:.0f}  | sum {sum(v[:5]) / n:.0f}; whole kernel per wave {v[5] / max(v[7], 1):.0f} cycles, "
                  f"{v[6] / max(v[7], 1):.0f} steps, {v[7]} waves")

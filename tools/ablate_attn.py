#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the fused QKV + window-attention kernel (development tool): loads scratch/libattn_stamp.so
(csrc/attn_bf16.hip built with -DPANGU_ATTN_STAMP) and prints where a workgroup's cycles go at the two model shapes."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int
lib = ctypes.CDLL(os.path.join(ROOT, "scratch", "libattn_stamp.so"))
lib.pangu_window_attn_qkv_fwd_bf16.argtypes = [P, P, I, P, P, P, P, P, I, I, I, I, I, I]
bf = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream
for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
    N = Z * H * W
    x = torch.randn(N, C, device="cuda").to(bf)
    w = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(bf)
    b = torch.randn(3 * C, device="cuda")
    esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(bf)
    out = torch.empty_like(x)
    for sh in (0, 1):
        buf = (ctypes.c_ulonglong * 8)()
        for _ in range(3):
            rc = lib.pangu_window_attn_qkv_fwd_bf16(stream, x.data_ptr(), C, w.data_ptr(), b.data_ptr(), esb.data_ptr(), out.data_ptr(),
                                                    None, Z, H, W, C, heads, sh)
            assert rc == 0, rc
        lib.pangu_attn_stamp_read(buf)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.pangu_window_attn_qkv_fwd_bf16(stream, x.data_ptr(), C, w.data_ptr(), b.data_ptr(), esb.data_ptr(), out.data_ptr(), None,
                                           Z, H, W, C, heads, sh)
        e.record()
        torch.cuda.synchronize()
        lib.pangu_attn_stamp_read(buf)
        v = list(buf)
        n = max(v[4], 1)
        print(f"C={C} shifted={sh}: {a.elapsed_time(e):.3f} ms; per wave (cycles): prologue {v[5] / n:.0f}  K-loop {v[0] / n:.0f}  "
              f"staging+barrier {v[1] / n:.0f}  3 attention tiles {v[2] / n:.0f}  whole {v[3] / n:.0f}  ({v[4]} waves)")

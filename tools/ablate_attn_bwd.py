#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the bf16 window-attention backward kernel (development tool): loads
scratch/libattn_bwd_stamp.so (csrc/attn_bwd_bf16.hip + attn_bf16.hip built with -DPANGU_ATTN_BWD_STAMP) and prints where a
wave's cycles go per longitude window at the two model shapes."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pangu_pytorch_amd  # noqa: E402,F401
from pangu_pytorch_amd import ops_bf16 as ob  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int
F32 = len(sys.argv) > 1 and sys.argv[1] == "f32"
if F32:
    from pangu_pytorch_amd import ops  # noqa: E402
    lib = ctypes.CDLL(os.path.join(ROOT, "scratch", os.environ.get("PANGU_BWD_LIB", "libattn_bwd_stamp.so")))
    lib.pangu_window_attn_bwd.argtypes = [P] * 10 + [I] * 6
    stream = torch.cuda.current_stream().cuda_stream
    for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
        N, nlon = Z * H * W, W // 12
        qkv = torch.randn(N, 3 * C, device="cuda")
        b1 = torch.randn(3 * C, device="cuda")
        esb = torch.randn(types, heads, 144, 144, device="cuda") * 0.1
        dout = torch.randn(N, C, device="cuda")
        dqkv, dqb, desb = torch.empty_like(qkv), torch.zeros(3 * C, device="cuda"), torch.empty(types, heads, 144, 144, device="cuda")
        for sh in (0, 1):
            out, lse = ops.window_attention(qkv, b1, esb, Z, H, W, heads, bool(sh), want_lse=True)
            args = (stream, qkv.data_ptr(), b1.data_ptr(), esb.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(),
                    dqkv.data_ptr(), dqb.data_ptr(), desb.data_ptr(), Z, H, W, C, heads, sh)
            buf = (ctypes.c_ulonglong * 8)()
            for _ in range(2):
                assert lib.pangu_window_attn_bwd(*args) == 0
            lib.pangu_attn_bwdf_stamp_read(buf)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            lib.pangu_window_attn_bwd(*args)
            e.record()
            torch.cuda.synchronize()
            lib.pangu_attn_bwdf_stamp_read(buf)
            v = list(buf)
            n = max(v[5], 1) * nlon
            print(f"f32 C={C} shifted={sh}: {a.elapsed_time(e):.3f} ms; owner wave per window (cycles): staging pass {v[0] / n:.0f} (wait at the top barrier {v[6] / n:.0f}, "
                  f"LDS writes + barrier {v[7] / n:.0f})  phase 1 {v[1] / n:.0f}  barriers A+B {v[2] / n:.0f}  dK/dV hand-over {v[3] / n:.0f}  phase 2 {v[4] / n:.0f}")
    sys.exit(0)
lib = ctypes.CDLL(os.path.join(ROOT, "scratch", os.environ.get("PANGU_BWD_LIB", "libattn_bwd_stamp.so")))
lib.pangu_window_attn_bwd_bf16.argtypes = [P] * 10 + [I] * 6
bf = torch.bfloat16
stream = torch.cuda.current_stream().cuda_stream
for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
    N = Z * H * W
    nlon = W // 12
    qkv = torch.randn(N, 3 * C, device="cuda").to(bf)
    b1 = torch.randn(3 * C, device="cuda").to(bf)
    esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(bf)
    dout = torch.randn(N, C, device="cuda").to(bf)
    dqkv = torch.empty_like(qkv)
    dqb = torch.zeros(3 * C, device="cuda")
    desb = torch.empty(types, heads, 144, 144, device="cuda")
    for sh in (0, 1):
        out, lse = ob.window_attention(qkv, b1, esb, Z, H, W, heads, bool(sh), want_lse=True)
        args = (stream, qkv.data_ptr(), b1.data_ptr(), esb.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(),
                dqkv.data_ptr(), dqb.data_ptr(), desb.data_ptr(), Z, H, W, C, heads, sh)
        buf = (ctypes.c_ulonglong * 8)()
        for _ in range(3):
            assert lib.pangu_window_attn_bwd_bf16(*args) == 0
        lib.pangu_attn_bwd_stamp_read(buf)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.pangu_window_attn_bwd_bf16(*args)
        e.record()
        torch.cuda.synchronize()
        lib.pangu_attn_bwd_stamp_read(buf)
        v = list(buf)
        n = max(v[5], 1) * nlon
        print(f"C={C} shifted={sh}: {a.elapsed_time(e):.3f} ms; per wave and window (cycles): staging {v[0] / n:.0f}  phase 1 {v[1] / n:.0f}  "
              f"dS barrier wait {v[2] / n:.0f}  phase 2 {v[3] / n:.0f}   | prologue {v[6] / max(v[5], 1):.0f}  whole kernel {v[4] / max(v[5], 1):.0f}  ({v[5]} waves)")

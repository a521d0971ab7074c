#!/usr/bin/env python3
"""Per-shape kernel micro-benchmarks on one MI355X (development tool, not the headline bench).

  python tools/bench_kernels.py gemm|attn|rows|mlp_fused|... [--lib-compare]
`--lib-compare` also times torch.matmul (hipBLASLt/rocBLAS) on the same shapes as an external yardstick."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pangu_pytorch_amd as P  # noqa: E402
from pangu_pytorch_amd import ops  # noqa: E402

GEMM_SHAPES = [  # (M, N, K, act, name)
    (521280, 576, 192, 0, "s0 qkv"), (521280, 192, 192, 0, "s0 proj"), (521280, 768, 192, 1, "s0 mlp1+gelu"),
    (521280, 192, 768, 0, "s0 mlp2"), (131040, 1152, 384, 0, "s1 qkv"), (131040, 384, 384, 0, "s1 proj"),
    (131040, 1536, 384, 1, "s1 mlp1+gelu"), (131040, 384, 1536, 0, "s1 mlp2"), (131040, 384, 768, 0, "down"),
    (131040, 768, 384, 0, "up1"), (456120, 192, 192, 0, "embed"), (456120, 160, 384, 0, "recover"),
]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def gemm(lib_compare):
    tot = tot_lib = 0.0
    for M, N, K, act, name in GEMM_SHAPES:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        ms = timeit(lambda: ops.linear(a, w, b, act=act, out=out))
        line = f"{name:14s} M={M:6d} N={N:4d} K={K:4d}  {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:6.1f} TF/s"
        if lib_compare:
            f = (lambda: torch.nn.functional.gelu(torch.addmm(b, a, w.t()))) if act else (lambda: torch.addmm(b, a, w.t()))
            ms2 = timeit(f)
            line += f"   | torch {ms2:7.3f} ms {2.0 * M * N * K / ms2 / 1e9:6.1f} TF/s"
        print(line)


def gemm_bf16(lib_compare):
    from pangu_pytorch_amd import ops_bf16 as ob
    bf = torch.bfloat16
    for M, N, K, act, name in GEMM_SHAPES:
        a = torch.randn(M, K, device="cuda").to(bf)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=bf)
        ms = timeit(lambda: ob.linear(a, w, b, act=act, out=out))
        by = (M * K + M * N + N * K) * 2.0
        line = f"{name:14s} M={M:6d} N={N:4d} K={K:4d}  {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:6.1f} TF/s {by / ms / 1e6:7.1f} GB/s"
        if lib_compare:
            bb = b.to(bf)
            f = (lambda: torch.nn.functional.gelu(torch.addmm(bb, a, w.t()))) if act else (lambda: torch.addmm(bb, a, w.t()))
            ms2 = timeit(f)
            line += f"   | torch {ms2:7.3f} ms {2.0 * M * N * K / ms2 / 1e9:6.1f} TF/s"
        print(line)


def mlp_fused():
    """One-launch MLP branch (csrc/mlp_fused_bf16.hip) vs the launches it replaces (MLP-up+GELU, MLP-down(+LN), LN)."""
    from pangu_pytorch_amd import ops_bf16 as ob
    bf = torch.bfloat16
    for M, C, name in ((521280, 192, "s0 mlp"), (131040, 384, "s1 mlp")):
        x = torch.randn(M, C, device="cuda").to(bf)
        w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(bf)
        w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(bf)
        b1, b2 = torch.randn(4 * C, device="cuda"), torch.randn(C, device="cuda")
        g, be = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        img = ob.pack_mlp_weights(w1, w2)
        out = torch.empty_like(x)
        ms = timeit(lambda: ob.mlp_ln_residual(x, img, b1, b2, g, be, out=out))

        def sep():
            h = ob.linear(x, w1, b1, act=ob.ACT_GELU)
            if C == 192:
                return ob.linear_ln_residual(h, w2, b2, x, g, be, out=out)
            return ob.ln_residual(ob.linear(h, w2, b2), x, g, be, out=out)
        ms2 = timeit(sep)
        fl = 16.0 * M * C * C
        print(f"{name:8s} M={M:6d} C={C:3d}  fused {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s ({fl / ms / 2.5e12 * 100:4.1f} % of 2.5 PF)"
              f"   | separate launches {ms2:7.3f} ms {fl / ms2 / 1e9:7.1f} TF/s")


def mlp_train():
    """Training forms of the MLP branch (bf16): the one-launch forward with its side outputs against the three launches it
    replaces, and the backward's data-gradient GEMM with / without re-creating h = GELU(pre)."""
    from pangu_pytorch_amd import ops_bf16 as ob
    bf = torch.bfloat16
    for M, C, name in ((521280, 192, "s0"), (131040, 384, "s1")):
        x = torch.randn(M, C, device="cuda").to(bf)
        w1 = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).to(bf)
        w2 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(bf)
        b1, b2 = torch.randn(4 * C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
        g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        img = ob.pack_mlp_weights(w1, w2)
        t_inf = timeit(lambda: ob.mlp_ln_residual(x, img, b1, b2, g, be))
        t_tr = timeit(lambda: ob.mlp_ln_residual_train(x, img, b1, b2, g, be))

        def sep():
            pre = torch.empty((M, 4 * C), dtype=bf, device="cuda")
            h = ob.linear(x, w1, b1, act=ob.ACT_GELU, aux=pre)
            return ob.ln_residual(ob.linear(h, w2, b2), x, g, be)
        t_sep = timeit(sep)
        dm = torch.randn(M, C, device="cuda").to(bf)
        pre = torch.randn(M, 4 * C, device="cuda").to(bf)
        w2t = w2.t().contiguous()
        t_b0 = timeit(lambda: ob.linear(dm, w2t, None, act=ob.ACT_GELU_BWD, aux=pre))
        t_b1 = timeit(lambda: ob.linear_gelu_bwd(dm, w2t, pre))
        print(f"{name} MLP forward  M={M:6d} C={C:3d}: inference launch {t_inf:6.3f} ms | training launch (pre + m out) {t_tr:6.3f} | "
              f"three launches writing pre AND h {t_sep:6.3f}")
        print(f"{name} MLP backward M={M:6d} C={C:3d}: dpre = (dm W2) * gelu'(pre) {t_b0:6.3f} ms | the same + h = GELU(pre) written {t_b1:6.3f}")


def gemm_ln():
    for M, N, K, name in ((521280, 192, 192, "s0 proj+LN"), (521280, 192, 768, "s0 mlp2+LN"), (131040, 384, 384, "s1 proj+LN"),
                          (131040, 384, 1536, "s1 mlp2+LN")):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b, g, be = torch.randn(N, device="cuda"), torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        sc = torch.randn(M, N, device="cuda")
        out, y = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
        ms = timeit(lambda: ops.linear_ln_residual(a, w, b, sc, g, be, out=out))
        ms2 = timeit(lambda: ops.ln_residual(ops.linear(a, w, b, out=y), sc, g, be, out=out))
        print(f"{name:12s} M={M:6d} N={N:4d} K={K:4d}  fused {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:6.1f} TF/s   | separate {ms2:7.3f} ms")


def gemm_ln_bf16():
    from pangu_pytorch_amd import ops_bf16 as ob
    bf = torch.bfloat16
    for M, N, K, name in ((521280, 192, 192, "s0 proj+LN"), (521280, 192, 768, "s0 mlp2+LN"), (131040, 384, 384, "s1 proj+LN"),
                          (131040, 384, 1536, "s1 mlp2+LN")):
        a = torch.randn(M, K, device="cuda").to(bf)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        b, g, be = torch.randn(N, device="cuda"), torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        sc = torch.randn(M, N, device="cuda").to(bf)
        out = torch.empty((M, N), device="cuda", dtype=bf)
        y = torch.empty((M, N), device="cuda", dtype=bf)
        ms = timeit(lambda: ob.linear_ln_residual(a, w, b, sc, g, be, out=out))
        ms2 = timeit(lambda: ob.ln_residual(ob.linear(a, w, b, out=y), sc, g, be, out=out))
        print(f"{name:12s} M={M:6d} N={N:4d} K={K:4d}  fused {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:6.1f} TF/s   | separate {ms2:7.3f} ms")


def wgrad(bf16):
    from pangu_pytorch_amd import ops_bf16 as ob
    dt = torch.bfloat16 if bf16 else torch.float32
    for M, N, K, act, name in GEMM_SHAPES:
        if bf16 and K % 8:
            continue
        dc = torch.randn(M, N, device="cuda").to(dt)
        a = torch.randn(M, K, device="cuda").to(dt)
        f = (lambda: ob.linear_wgrad(dc, a)) if bf16 else (lambda: ops.linear_wgrad(dc, a))
        ms = timeit(f)
        line = f"wgrad {'bf16' if bf16 else 'f32 '} {name:14s} M={M:6d} N={N:4d} K={K:4d}  {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:6.1f} TF/s"
        if "--lib-compare" in sys.argv:          # external yardstick only: dW = dC^T A (+ column sums) through torch / hipBLASLt
            ms2 = timeit(lambda: (torch.mm(dc.t(), a), dc.sum(0)))
            line += f"   | torch {ms2:7.3f} ms {2.0 * M * N * K / ms2 / 1e9:6.1f} TF/s"
        print(line)


def attn_bf16():
    from pangu_pytorch_amd import ops_bf16 as ob
    bf = torch.bfloat16
    for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
        N = Z * H * W
        qkv = torch.randn(N, 3 * C, device="cuda").to(bf)
        b1 = torch.randn(3 * C, device="cuda").to(bf)
        esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(bf)
        for sh in (False, True):
            ms = timeit(lambda: ob.window_attention(qkv, b1, esb, Z, H, W, heads, sh))
            Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
            fl = 4.0 * Np * 144 * C
            by = (N * 4 * C + esb.numel()) * 2.0
            print(f"attn_bf16 C={C} shifted={int(sh)}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF/s  {by / ms / 1e6:7.1f} GB/s (algorithmic)")


def attn_qkv_bf16():
    """QKV projection fused into the attention kernel vs the two launches it replaces."""
    from pangu_pytorch_amd import ops_bf16 as ob
    bf = torch.bfloat16
    for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
        N = Z * H * W
        x = torch.randn(N, C, device="cuda").to(bf)
        w = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(bf)
        b = torch.randn(3 * C, device="cuda")
        bb = b.to(bf)
        esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(bf)
        Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
        fl = 4.0 * Np * 144 * C + 6.0 * Np * C * C
        for sh in (False, True):
            ms = timeit(lambda: ob.window_attention_qkv(x, w, b, esb, Z, H, W, heads, sh))
            ms2 = timeit(lambda: ob.window_attention(ob.linear(x, w, b), bb, esb, Z, H, W, heads, sh))
            print(f"attn_qkv_bf16 C={C} shifted={int(sh)}: fused {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF/s   | qkv GEMM + attention {ms2:7.3f} ms")


def attn_bwd():
    """backward of the window attention, both dtypes: 5 useful GEMMs (dP, dV, dQ, dK + recomputed S) = 10*Np*144*C flop"""
    from pangu_pytorch_amd import ops_bf16 as ob
    for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
        N = Z * H * W
        Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
        for mod, dt, tag in ((ops, torch.float32, "f32 "), (ob, torch.bfloat16, "bf16")):
            qkv = torch.randn(N, 3 * C, device="cuda").to(dt)
            b1 = torch.randn(3 * C, device="cuda").to(dt)
            esb = (torch.randn(types, heads, 144, 144, device="cuda") * 0.1).to(dt)
            dout = torch.randn(N, C, device="cuda").to(dt)
            for sh in (False, True):
                out, lse = mod.window_attention(qkv, b1, esb, Z, H, W, heads, sh, want_lse=True)
                ms = timeit(lambda: mod.window_attention_bwd(qkv, b1, esb, out, lse, dout, Z, H, W, heads, sh))
                print(f"attn_bwd {tag} C={C} shifted={int(sh)}: {ms:7.3f} ms  {10.0 * Np * 144 * C / ms / 1e9:6.1f} TF/s")
            del qkv, esb, dout, out, lse


def attn():
    for C, Z, H, W, heads, types in ((192, 8, 181, 360, 6, 124), (384, 8, 91, 180, 12, 64)):
        N = Z * H * W
        qkv = torch.randn(N, 3 * C, device="cuda")
        b1 = torch.randn(3 * C, device="cuda")
        esb = torch.randn(types, heads, 144, 144, device="cuda") * 0.1
        table = torch.randn(types, heads, 3312, device="cuda") * 0.1      # the paper's compact bias table, kernel layout
        for sh in (False, True):
            ms = timeit(lambda: ops.window_attention(qkv, b1, esb, Z, H, W, heads, sh))
            msc = timeit(lambda: ops.window_attention(qkv, b1, table, Z, H, W, heads, sh, compact=True))
            Np = (Z // 2) * ((H + 5) // 6) * (W // 12) * 144
            fl = 4.0 * Np * 144 * C
            by = (N * 4 * C + esb.numel()) * 4.0
            print(f"attn C={C} shifted={int(sh)}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF/s  {by / ms / 1e6:7.1f} GB/s (algorithmic)"
                  f"   | compact bias table {msc:7.3f} ms  {fl / msc / 1e9:6.1f} TF/s")


def rows():
    for N, C in ((521280, 192), (131040, 384)):
        y, s = torch.randn(N, C, device="cuda"), torch.randn(N, C, device="cuda")
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        out = torch.empty_like(y)
        ms = timeit(lambda: ops.ln_residual(y, s, g, b, out=out))
        print(f"ln_residual N={N} C={C}: {ms:7.3f} ms  {3.0 * N * C * 4 / ms / 1e6:7.1f} GB/s")
        from pangu_pytorch_amd import ops_bf16 as ob
        yb, sb = y.bfloat16(), s.bfloat16()
        ob_out = torch.empty_like(yb)
        ms = timeit(lambda: ob.ln_residual(yb, sb, g, b, out=ob_out))
        print(f"ln_residual bf16 N={N} C={C}: {ms:7.3f} ms  {3.0 * N * C * 2 / ms / 1e6:7.1f} GB/s")
        ms = timeit(lambda: ob.ln_residual_bwd(sb, yb, g, 1.0))
        print(f"ln_residual_bwd bf16 N={N} C={C}: {ms:7.3f} ms  {3.0 * N * C * 2 / ms / 1e6:7.1f} GB/s")
        ms = timeit(lambda: ops.ln_residual_bwd(s, y, g, 1.0))
        print(f"ln_residual_bwd f32  N={N} C={C}: {ms:7.3f} ms  {3.0 * N * C * 4 / ms / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    {"gemm": lambda: gemm("--lib-compare" in sys.argv), "attn": attn, "attn_bwd": attn_bwd, "rows": rows,
     "gemm_bf16": lambda: gemm_bf16("--lib-compare" in sys.argv), "attn_bf16": attn_bf16, "gemm_ln_bf16": gemm_ln_bf16, "mlp_fused": mlp_fused, "mlp_train": mlp_train, "attn_qkv_bf16": attn_qkv_bf16, "gemm_ln": gemm_ln,
     "wgrad_bf16": lambda: wgrad(True), "wgrad": lambda: wgrad(False)}[what]()

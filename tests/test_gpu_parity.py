"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference-generated golden
fixtures.  Needs an MI355X: run with `-m gpu`.

Tolerances: integer/index/mask tensors bit-exact; fp32 kernels <= 1e-3 relative (north_star), in practice
1e-5..1e-4 (both sides accumulate in fp32; only summation order differs)."""
import os

import numpy as np
import pytest
import torch

import cases
import pangu_oracle as O
import synth

pytestmark = pytest.mark.gpu
REL = 1e-3          # the contract (BASELINE.json north_star)
TIGHT = 2e-4        # what fp32-vs-fp32 should achieve; a regression canary


@pytest.fixture(scope="module")
def P():
    import pangu_pytorch_amd as P
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    P._lib.load()
    return P


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


# ---------------------------------------------------------------- integer contract
@pytest.mark.parametrize("Z,H,W", [(8, 181, 24), (8, 91, 24), (8, 181, 360), (8, 91, 180)])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_index_bit_exact(P, Z, H, W, shifted):
    got = P.ops.window_index(Z, H, W, shifted, "cuda").cpu()
    assert torch.equal(got, O.window_source_index(Z, H, W, shifted))


@pytest.mark.parametrize("Z,H,W", [(8, 181, 24), (8, 91, 24), (8, 181, 360)])
def test_window_mask_bit_exact(P, Z, H, W):
    got = P.ops.window_mask(Z, H, W, "cuda").cpu()
    assert torch.equal(got, O.shift_mask(Z, H, W))


def test_block_gen_mask_dropin(P, golden_dir):
    """EarthSpecificBlock.gen_mask(x) of the reference (layers.py:153-181): (nLon, types, 144, 144) in {0, -100}."""
    g = np.load(os.path.join(golden_dir, "index.npz"))
    for C, heads in ((192, 6), (384, 12)):
        st = cases.STAGES[C]
        blk = P.layers.EarthSpecificBlock(C, 0.0, heads, device="cuda")
        x = torch.zeros(1, st["Z"], st["H"] + 5, 24, 1, device="cuda")          # the padded activation the reference passes
        m = blk.gen_mask(x)
        assert m.shape == (2, blk.type_of_windows, 144, 144) and m.dtype == torch.float32
        assert torch.equal(m[0].cpu(), O.shift_mask(st["Z"], st["H"], 24)) and torch.equal(m[0], m[1])
        assert np.array_equal(np.packbits(m[0].cpu().numpy() != 0), g[f"mask_bits_{C}"])


def test_window_index_golden(P, golden_dir):
    g = np.load(os.path.join(golden_dir, "index.npz"))
    for C in (192, 384):
        st = cases.STAGES[C]
        for roll in (0, 1):
            got = P.ops.window_index(st["Z"], st["H"], 24, roll, "cuda").cpu().numpy()
            assert np.array_equal(got, g[f"win_index_{C}_{roll}"])
        m = P.ops.window_mask(st["Z"], st["H"], 24, "cuda").cpu().numpy()
        assert np.array_equal(np.packbits(m != 0), g[f"mask_bits_{C}"])


# ---------------------------------------------------------------- projections
@pytest.mark.parametrize("M,N,K,act,bias", [
    (1000, 192, 192, 0, True), (4099, 576, 192, 0, True), (2048, 768, 192, 1, True), (777, 192, 768, 0, True),
    (1531, 1152, 384, 0, True), (513, 1536, 384, 1, True), (300, 384, 1536, 0, True), (999, 384, 768, 0, False),
    (1234, 160, 384, 0, True), (321, 64, 384, 0, True), (650, 192, 112, 0, True), (128, 128, 16, 0, False),
])
def test_linear(P, M, N, K, act, bias):
    a = synth.uniform((M, K), 11)
    w = synth.uniform((N, K), 12, 1.0 / K ** 0.5)
    b = synth.uniform((N,), 13, 0.5) if bias else None
    ref = a @ w.t()
    if bias:
        ref = ref + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    got = P.ops.linear(a.cuda(), w.cuda(), b.cuda() if bias else None, act=act)
    assert rel_err(got, ref) < TIGHT


@pytest.mark.parametrize("M,N,K,bias", [(4099, 192, 768, False), (1531, 384, 1152, True), (650, 160, 384, False)])
def test_linear_add_epilogue(P, M, N, K, bias):
    """act=ADD: out = a @ w^T + bias + aux (the residual-gradient accumulation of the block backward)."""
    a = synth.uniform((M, K), 14)
    w = synth.uniform((N, K), 15, 1.0 / K ** 0.5)
    b = synth.uniform((N,), 16, 0.5) if bias else None
    add = synth.uniform((M, N), 17, 2.0)
    ref = a @ w.t() + add + (b if bias else 0.0)
    got = P.ops.linear(a.cuda(), w.cuda(), b.cuda() if bias else None, act=P.ops.ACT_ADD, aux=add.cuda())
    assert rel_err(got, ref) < TIGHT


@pytest.mark.parametrize("M,K,bias,scale", [(1000, 192, True, 1.0), (4099, 768, True, 1.25), (777, 192, False, 1.0),
                                              (128, 16, True, 0.5)])
@pytest.mark.parametrize("strided", [False, True])
@pytest.mark.parametrize("N", [192, 384])
def test_linear_ln_residual(P, M, K, bias, scale, strided, N):
    """Fused projection + post-norm residual == linear -> LayerNorm -> shortcut + scale * (.) (reference layers.py:250-251)."""
    a = synth.uniform((M, K), 81)
    w = synth.uniform((N, K), 82, 1.0 / K ** 0.5)
    b = synth.uniform((N,), 83, 0.5) if bias else None
    sc = synth.uniform((M, N), 84, 1.5)
    g, be = synth.uniform((N,), 85, 0.5, 1.0), synth.uniform((N,), 86, 0.3)
    y = a @ w.t() + (b if bias else 0.0)
    ref = sc + scale * torch.nn.functional.layer_norm(y, (N,), g, be, 1e-5)
    out, scd = None, sc.cuda()
    if strided:
        full = torch.zeros((M, 2 * N), device="cuda")
        out = full[:, N:]
        scf = torch.zeros((M, 2 * N), device="cuda")
        scf[:, :N] = scd
        scd = scf[:, :N]
    got = P.ops.linear_ln_residual(a.cuda(), w.cuda(), b.cuda() if bias else None, scd, g.cuda(), be.cuda(), out=out,
                                   branch_scale=scale)
    assert rel_err(got, ref) < TIGHT
    if strided:
        assert float(full[:, :N].abs().max()) == 0.0


def test_linear_random_shapes(P):
    """Ragged M, every K % 16 == 0 up to 1600, all tile families (192-wide / 128-wide LDS-DMA tiles, register-staged
    TN = 1), bias / GELU / residual-add epilogues: the projection GEMM against an fp64 product on the same inputs."""
    import random
    rnd = random.Random(7)
    torch.manual_seed(7)
    for _ in range(40):
        N = rnd.choice([160, 192, 384, 576, 768, 1152, 1536, 132, 176, 64])
        K = 16 * rnd.randint(1, 100)
        M = rnd.randint(1, 3000)
        act = rnd.choice([0, 0, 1, 3])
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda") if rnd.random() < 0.7 else None
        aux = torch.randn(M, N, device="cuda") if act == 3 else None
        ref = a.double() @ w.double().t()
        if b is not None:
            ref = ref + b.double()
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        if act == 3:
            ref = ref + aux.double()
        got = P.ops.linear(a, w, b, act=act, aux=aux)
        err = ((got.double() - ref).norm() / ref.norm()).item()
        assert err < 2e-6, (M, N, K, act, err)


def test_linear_strided_rows(P):
    a_full = synth.uniform((700, 384), 21).cuda()
    w = synth.uniform((192, 192), 22, 0.07).cuda()
    out_full = torch.zeros((700, 384), device="cuda")
    P.ops.linear(a_full[:, 192:], w, None, out=out_full[:, :192])
    ref = a_full[:, 192:].cpu() @ w.cpu().t()
    assert rel_err(out_full[:, :192], ref) < TIGHT
    assert float(out_full[:, 192:].abs().max()) == 0.0


# ---------------------------------------------------------------- attention core
@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_core(P, C, shifted):
    st = cases.STAGES[C]
    Z, H, W, heads = st["Z"], st["H"], 24, st["heads"]
    N = Z * H * W
    qkv = synth.uniform((1, N, 3 * C), 31, 1.5)
    b1 = synth.uniform((3 * C,), 32, 0.5)
    esb = synth.uniform((1, st["types"], heads, 144, 144), 33, 0.5)
    ref, ref_lse = O.window_attention_core(qkv, b1, esb, Z, H, W, heads, shifted)
    got, lse = P.ops.window_attention(qkv[0].cuda(), b1.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    assert rel_err(got, ref[0]) < TIGHT
    assert rel_err(lse, ref_lse[0]) < TIGHT


# ---------------------------------------------------------------- row kernels
@pytest.mark.parametrize("C", [192, 384, 768])
def test_ln_residual(P, C):
    N = 1003
    y, sc = synth.uniform((N, C), 41, 2.0, 0.3), synth.uniform((N, C), 42)
    g, b = synth.uniform((C,), 43, 0.1, 1.0), synth.uniform((C,), 44, 0.1)
    ref = sc + 1.25 * torch.nn.functional.layer_norm(y, (C,), g, b)
    got, stats = P.ops.ln_residual(y.cuda(), sc.cuda(), g.cuda(), b.cuda(), branch_scale=1.25, want_stats=True)
    assert rel_err(got, ref) < TIGHT
    assert rel_err(stats[:, 0], y.mean(1)) < TIGHT
    assert rel_err(stats[:, 1], (y.var(1, unbiased=False) + 1e-5).rsqrt()) < TIGHT


def test_downsample_upsample(P):
    Z, H, W, C = 8, 181, 24, 192
    p = {k: synth.synth_param(k, s) for k, s in cases.model_param_shapes().items()
         if k.startswith(("downsample.", "upsample."))}
    x = synth.uniform((1, Z * H * W, C), 51)
    ref = O.down_sample(p, x, Z, H, W)
    g = P.ops.downsample_ln(x[0].cuda(), p["downsample.norm.weight"].cuda(), p["downsample.norm.bias"].cuda(), Z, H, W)
    got = P.ops.linear(g, p["downsample.linear.weight"].cuda())
    assert rel_err(got, ref[0]) < TIGHT
    H2, W2 = 91, 12
    x2 = synth.uniform((1, Z * H2 * W2, 384), 52)
    ref = O.up_sample(p, x2, Z, H2, W2, H)
    y = P.ops.linear(x2[0].cuda(), p["upsample.linear1.weight"].cuda())
    g = P.ops.upsample_ln(y, p["upsample.norm.weight"].cuda(), p["upsample.norm.bias"].cuda(), Z, H2, W2, H)
    got = P.ops.linear(g, p["upsample.linear2.weight"].cuda())
    assert rel_err(got, ref[0]) < TIGHT


def test_patch_embed_gather_and_recover_small(P):
    """Ragged sizes: LAT not a multiple of 4, W4 not a multiple of the 64-token chunk."""
    LAT, LON = 41, 4 * 70
    H4, W4 = 11, 70
    g = lambda n, s, sc=1.0, sh=0.0: synth.uniform(s, synth.name_seed(n), sc, sh)
    inp, inp_s = g("i", (1, 5, 13, LAT, LON)), g("is", (1, 4, LAT, LON))
    stats = (g("sm", (4,), 0.3), g("ss", (4,), 0.2, 1.2), g("um", (13, 1, 1, 5), 0.3), g("us", (13, 1, 1, 5), 0.2, 1.2))
    maps, const_h = g("m", (1, 3, 4 * H4, LON)), g("c", (1, 1, 1, 13, LAT, LON))
    ra_s, ra_u = O.patch_embed_matrices(inp, inp_s, stats, maps, const_h)
    a_s, a_u = P.ops.patch_embed_gather(inp[0].cuda(), inp_s[0].cuda(), stats[0].cuda(), stats[1].cuda(),
                                        stats[2].reshape(13, 5).cuda(), stats[3].reshape(13, 5).cuda(),
                                        maps[0].cuda(), const_h.reshape(13, LAT, LON).cuda())
    assert rel_err(a_s, ra_s[0]) < 1e-6 and rel_err(a_u, ra_u[0]) < 1e-6
    # recover: scatter of random GEMM outputs == oracle's un-patchify with identity convs
    yu, ys = g("yu", (7 * H4 * W4, 160)), g("ys", (H4 * W4, 64))
    o, os_ = P.ops.patch_recover_scatter(yu.cuda(), ys.cuda(), LAT, LON)
    ro = yu.view(7, H4, W4, 5, 2, 4, 4).permute(3, 0, 4, 1, 5, 2, 6).reshape(5, 14, 4 * H4, LON)[:, :13, :LAT]
    rs = ys.view(H4, W4, 4, 4, 4).permute(2, 0, 3, 1, 4).reshape(4, 4 * H4, LON)[:, :LAT]
    assert torch.equal(o.cpu(), ro) and torch.equal(os_.cpu(), rs)


@pytest.mark.parametrize("LAT,LON", [(41, 280), (721, 1440)])
def test_patch_recover_gather_bwd_is_the_scatter_adjoint(P, LAT, LON):
    """Backward of the patch-recovery un-patchify (reference layers.py:522-543): pure data movement, so the gathered gradient
    must equal the inverse permutation of the field gradients bit for bit (fp32) / after one bf16 rounding (bf16), with zeros in
    the cropped positions (level 14, latitudes >= LAT); ragged sizes and the model's."""
    from pangu_pytorch_amd import ops_bf16 as ob
    H4, W4 = (LAT + 3) // 4, LON // 4
    d_o = synth.uniform((5, 13, LAT, LON), synth.name_seed("gb_o"))
    d_os = synth.uniform((4, LAT, LON), synth.name_seed("gb_os"))
    full = torch.zeros(5, 14, 4 * H4, LON)
    full[:, :13, :LAT] = d_o
    ref_u = full.view(5, 7, 2, H4, 4, W4, 4).permute(1, 3, 5, 0, 2, 4, 6).reshape(7 * H4 * W4, 160)
    fs = torch.zeros(4, 4 * H4, LON)
    fs[:, :LAT] = d_os
    ref_s = fs.view(4, H4, 4, W4, 4).permute(1, 3, 0, 2, 4).reshape(H4 * W4, 64)
    dy_u, dy_s = P.ops.patch_recover_gather_bwd(d_o.cuda(), d_os.cuda())
    assert torch.equal(dy_u.cpu(), ref_u) and torch.equal(dy_s.cpu(), ref_s)
    by_u, by_s = ob.patch_recover_gather_bwd(d_o.cuda(), d_os.cuda())
    assert torch.equal(by_u.cpu(), ref_u.to(torch.bfloat16)) and torch.equal(by_s.cpu(), ref_s.to(torch.bfloat16))


# ---------------------------------------------------------------- block level vs reference golden + oracle
def _load_block(P, C, roll):
    st = cases.STAGES[C]
    blk = P.layers.EarthSpecificBlock(C, 0.1, st["heads"], device="cuda").cuda().eval()
    pre = cases.block_prefix(C, roll)
    blk.load_state_dict({k: synth.synth_param(pre + k, s, "cuda") for k, s in cases.block_param_shapes(C).items()})
    return blk, st


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("roll", [False, True])
def test_block_forward_golden(P, golden_dir, C, roll):
    tag = f"block_{C}_{int(roll)}"
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    blk, st = _load_block(P, C, roll)
    x = cases.block_input(C, 24, "cuda")
    with torch.no_grad():
        y = blk(x, st["Z"], st["H"], 24, roll)
    assert cases.compare_summary(y, g, tag + ".out", REL) < TIGHT


@pytest.mark.parametrize("C,W", [(192, 12), (384, 36)])
def test_block_forward_oracle_other_widths(P, C, W):
    blk, st = _load_block(P, C, True)
    x = cases.block_input(C, W, "cuda")
    with torch.no_grad():
        y = blk(x, st["Z"], st["H"], W, True)
    p = cases.block_params(C, True)
    ref = O.earth_block(p, cases.block_prefix(C, True), x.cpu(), st["Z"], st["H"], W, st["heads"], True)
    assert rel_err(y, ref) < TIGHT


# ---------------------------------------------------------------- full-resolution layers and model
def _model(P):
    m = P.PanguModel(device="cuda").cuda().eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    return m


def test_fullres_layers_golden(P, golden_dir):
    g = np.load(os.path.join(golden_dir, "layers_fullres.npz"))
    m = _model(P)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        x0 = m._input_layer(inp, inp_s, stats, maps, const_h)
        assert x0.shape == (1, 521280, 192)
        assert cases.compare_summary(x0, g, "embed.out", REL) < TIGHT
        xin = synth.uniform((1, 8 * 181 * 360, 192), synth.name_seed("down_in"), device="cuda")
        assert cases.compare_summary(m.downsample(xin, 8, 181, 360), g, "down.out", REL) < TIGHT
        xin = synth.uniform((1, 8 * 91 * 180, 384), synth.name_seed("up_in"), device="cuda")
        assert cases.compare_summary(m.upsample(xin), g, "up.out", REL) < TIGHT
        xin = synth.uniform((1, 8 * 181 * 360, 384), synth.name_seed("recover_in"), device="cuda")
        o, os_ = m._output_layer(xin, 8, 181, 360)
        assert o.shape == (1, 5, 13, 721, 1440) and os_.shape == (1, 4, 721, 1440)
        assert cases.compare_summary(o, g, "recover.out", REL) < TIGHT
        assert cases.compare_summary(os_, g, "recover.out_surface", REL) < TIGHT


def test_full_model_forward_golden(P, golden_dir):
    """BASELINE config 2: single-GPU fp32 forward, parity vs the reference's CPU forward <= 1e-3."""
    path = os.path.join(golden_dir, "model_fwd.npz")
    if not os.path.exists(path):
        pytest.skip("model_fwd.npz not generated")
    g = np.load(path)
    m = _model(P)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    with torch.no_grad():
        out, out_s = m(inp, inp_s, stats, maps, const_h)
    assert out.shape == (1, 5, 13, 721, 1440) and out_s.shape == (1, 4, 721, 1440)
    assert torch.isfinite(out).all() and torch.isfinite(out_s).all()
    assert cases.compare_summary(out, g, "model.out", REL) < REL
    assert cases.compare_summary(out_s, g, "model.out_surface", REL) < REL
    # two-argument call with registered constants gives the same result
    m.set_constants(stats, maps, const_h)
    with torch.no_grad():
        out2, _ = m(inp, inp_s)
    assert torch.equal(out, out2)


def test_batch_of_two_equals_two_singles(P):
    """The reference is B=1 only (layers.py:219,227); the drop-in is batch-generic: B=2 == two B=1 calls, bit for bit."""
    m = _model(P)
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    inp2 = torch.cat((inp, synth.uniform(inp.shape, 4242, device="cuda")), 0)
    inp_s2 = torch.cat((inp_s, synth.uniform(inp_s.shape, 4243, device="cuda")), 0)
    with torch.no_grad():
        o2, os2 = m(inp2, inp_s2, stats, maps, const_h)
        oa, osa = m(inp2[:1], inp_s2[:1], stats, maps, const_h)
        ob, osb = m(inp2[1:], inp_s2[1:], stats, maps, const_h)
    assert o2.shape == (2, 5, 13, 721, 1440)
    assert torch.equal(o2[0], oa[0]) and torch.equal(o2[1], ob[0]) and torch.equal(os2[1], osb[0])


@pytest.mark.parametrize("Z,H,W,heads", [(2, 1, 12, 1), (4, 7, 24, 2), (2, 13, 36, 5), (6, 19, 12, 3)])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_other_geometries(P, Z, H, W, heads, shifted):
    """Geometries other than the model's two (odd head counts, a single latitude row, three z-windows): the closed-form
    window addressing / mask against the oracle's explicit gather index, fp32 and bf16 kernels."""
    from pangu_pytorch_amd import ops_bf16 as ob
    C = 32 * heads
    N = Z * H * W
    types = (Z // 2) * ((H + 5) // 6)
    qkv = synth.uniform((1, N, 3 * C), 91, 1.5)
    b1 = synth.uniform((3 * C,), 92, 0.5)
    esb = synth.uniform((1, types, heads, 144, 144), 93, 0.5)
    ref, ref_lse = O.window_attention_core(qkv, b1, esb, Z, H, W, heads, shifted)
    got, lse = P.ops.window_attention(qkv[0].cuda(), b1.cuda(), esb[0].cuda(), Z, H, W, heads, shifted, want_lse=True)
    assert rel_err(got, ref[0]) < TIGHT
    bf = torch.bfloat16
    q16, b16, e16 = qkv.to(bf), b1.to(bf), esb.to(bf)
    ref16, _ = O.window_attention_core(q16.float(), b16.float(), e16.float(), Z, H, W, heads, shifted)
    got16 = ob.window_attention(q16[0].cuda(), b16.cuda(), e16[0].cuda(), Z, H, W, heads, shifted)
    assert rel_err(got16, ref16[0]) < 1.0 / 64


def test_linear_rows_beyond_4gb(P):
    """The GEMM kernels use 32-bit byte offsets (range-checked buffer addressing): the C entry refuses a matrix whose
    rows reach 4 GB (PANGU_E_RANGE, nothing launched) and the Python layer splits such calls by rows -- results equal."""
    ops = P.ops
    ld, M, K, N = 1 << 20, 1100, 64, 128                 # 1100 rows x 4 MB row stride = 4.6 GB
    big = torch.empty((M, ld), dtype=torch.float32, device="cuda")
    a = big[:, :K]
    a.copy_(synth.uniform((M, K), 31, device="cuda"))
    w = synth.uniform((N, K), 32, device="cuda")
    b = synth.uniform((N,), 33, device="cuda")
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    lib = P._lib.load()
    rc = lib.pangu_linear_fwd(torch.cuda.current_stream().cuda_stream, a.data_ptr(), ld, w.data_ptr(), b.data_ptr(),
                              out.data_ptr(), N, M, N, K, 0, None)
    assert rc == -5                                      # PANGU_E_RANGE
    got = ops.linear(a, w, b)
    ref = a.double() @ w.double().t() + b.double()
    assert ((got.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    dw, db = ops.linear_wgrad(a, a)                      # both operands beyond 4 GB: chunks accumulate into one dW
    refw = a.double().t() @ a.double()
    assert ((dw.double() - refw).abs().max() / refw.abs().max()).item() < 1e-5
    assert ((db.double() - a.double().sum(0)).abs().max() / a.double().sum(0).abs().max()).item() < 1e-5
    del big

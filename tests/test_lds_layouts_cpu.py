"""LDS images of the bf16 attention kernels (csrc/attn_bf16.hip) under the gfx950 bank rules (tools/lds_banks.py =
MI355X_MICROARCH.md's LDS table): every fragment read conflict-free, the writes as documented.  The address functions below
restate the kernel's (kswz, vt_off<true>, key_of); VERDICT r3 item 2 (bank-conflict share 0.20-0.27 of the fused QKV + attention kernel)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from lds_banks import conflict_free, cycles  # noqa: E402

F = [0, 2, 3, 1]


def kswz(row, chunk):
    return row * 64 + ((chunk ^ ((F[(row >> 3) & 3] ^ ((row >> 1) & (3 if row < 128 else 1))) & 3)) << 4)


def vt_off(d, key):
    c = key >> 3
    return d * 384 + ((((c ^ (d >> 1) ^ (d >> 4)) & 7) | (c & ~7)) << 4) + (key & 7) * 2


def key_of(j, i):
    return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3) if j < 8 else 128 + i


def test_images_are_bijective():
    assert len({kswz(r, c) for r in range(144) for c in range(4)}) == 576 and max(kswz(r, c) for r in range(144) for c in range(4)) < 9216
    offs = {vt_off(d, k) for d in range(32) for k in range(144)}
    assert len(offs) == 32 * 144 and max(offs) < 32 * 384 and all(o % 2 == 0 for o in offs)
    for d in range(32):                      # 8 consecutive keys = one 16-B chunk, 4 consecutive keys = one aligned 8-B piece
        for k in range(0, 144, 8):
            assert vt_off(d, k) % 16 == 0 and [vt_off(d, k + e) - vt_off(d, k) for e in range(8)] == [2 * e for e in range(8)]


@pytest.mark.parametrize("j", range(9))
def test_k_fragment_reads_conflict_free(j):
    assert conflict_free("read_b128", lambda l: kswz(key_of(j, l & 15), l >> 4))


@pytest.mark.parametrize("tile", range(9))
def test_k_image_writes_of_the_fused_kernel_conflict_free(tile):
    # lane (lq, lg) writes the 16-B piece (row tile * 16 + lq, logical chunk lg); the tail tile's rows: 2-way
    c, n = cycles("write_b128", lambda l: kswz(tile * 16 + (l & 15), l >> 4))
    assert c == n if tile < 8 else c <= 2 * n
    # the unfused kernel: thread f -> (row f >> 2, chunk f & 3)
    assert conflict_free("write_b128", lambda l: kswz(tile * 16 + (l >> 2), l & 3))


@pytest.mark.parametrize("dt", range(2))
def test_vt_fragment_reads_conflict_free(dt):
    for u in range(4):
        assert conflict_free("read_b128", lambda l: vt_off(dt * 16 + (l & 15), 32 * u + 8 * (l >> 4)))
    assert conflict_free("read_b64", lambda l: vt_off(dt * 16 + (l & 15), 128 + 4 * (l >> 4)))


def test_vt_writes():
    # (the kernel that reads a qkv tensor keeps the padded 336-byte rows: its 2-byte scatter addresses are cheaper that way)
    # fused kernel: 8-B pieces (row 16 dt + lq, keys 16 tile + 4 lg ..): 2-way (six instructions per wave and launch)
    for t in range(9):
        c, n = cycles("write_b64", lambda l: vt_off(l & 15, 16 * t + 4 * (l >> 4)))
        assert c <= 2 * n


def test_attn_bwd_bf16_key_query_images():
    """csrc/attn_bwd_bf16.hip: the [key][query] bf16 images (dS of the current window, bias^T), 288-byte rows, 8-byte pieces
    swizzled inside a row (dS: ^ ((key >> 2) & 3), bias^T: ^ 2 ((key >> 3) & 1)).  lane = (lq = l & 15, lg = l >> 4)."""
    LD = 288
    for wave in range(9):
        kn = lambda l: wave * 16 + (l & 15)
        for i in range(9):
            # phase 1: dS piece write of key kn, queries 16 i + 4 lg ..; bias^T piece read
            assert conflict_free("write_b64", lambda l: kn(l) * LD + 8 * ((l >> 4) ^ (((l & 15) >> 2) & 3)) + 32 * i)
            assert conflict_free("read_b64", lambda l: kn(l) * LD + 8 * ((l >> 4) ^ (2 * ((l & 15) >> 3))) + 32 * i)
        for m in range(9):
            # phase 2: transposed read of 4 key rows x 16 queries per 16-lane group: rows 16 m + 4 lg + tq, piece 4 wave + (tp ^ lg)
            assert conflict_free("read_tr16", lambda l: (16 * m + 4 * (l >> 4) + ((l & 15) >> 2)) * LD + wave * 32 + 8 * ((l & 3) ^ ((l >> 4) & 3)))
    # the swizzles are bijections of a row's 36 pieces and the reader finds what the writer stored
    for key in range(144):
        assert sorted((p ^ ((key >> 2) & 3)) for p in range(36)) == list(range(36))
        assert sorted((p ^ (2 * ((key >> 3) & 1))) for p in range(36)) == list(range(36))


def _wgrad_swz(rowb, row, chunk):
    mask = 2 * (row & 7) if rowb % 256 == 0 else 2 * ((row >> 1) & 3)
    return row * rowb + ((chunk ^ mask) << 4)


@pytest.mark.parametrize("rowb", [256, 384, 768])
def test_wgrad_bf16_dma_slab_images(rowb):
    """csrc/wgrad_bf16_dma.hip: token-major slabs of 32 rows x rowb bytes (dC at 2 / 3 / 6 wave rows, A at 2 / 4 wave columns),
    16-B chunks XOR-swizzled on the DMA's source side.  Every transposing fragment read (lane 4q + p of a 16-lane group: token row
    row0 + 4 lg + q, 8 bytes inside chunk col0 / 8 + (p >> 1)) is conflict-free, and the swizzle permutes the chunks of a row."""
    nch = rowb // 16
    for row in range(32):
        assert sorted(_wgrad_swz(rowb, row, c) for c in range(nch)) == [row * rowb + 16 * c for c in range(nch)]
    for row0 in (0, 16):
        for col0 in range(0, rowb // 2, 16):
            def addr(l):
                lc, lg = l & 15, l >> 4
                return _wgrad_swz(rowb, row0 + 4 * lg + (lc >> 2), (col0 >> 3) + ((lc & 3) >> 1)) + 8 * (lc & 1)
            assert conflict_free("read_tr16", addr), (rowb, row0, col0)


def test_closed_form_fragment_addresses_of_attn_tile():
    """Round 6: `attn_tile` (csrc/attn_bf16_tile.h) no longer evaluates kswz / vt_off per fragment read but uses, per lane, six bases
    + immediates -- K: (j odd ? ke ^ 32 : ke) + 2048 (j >> 1) + 256 (j & 1), tail kt; V^T (swizzled image): (u odd ? vb ^ 64 : vb) +
    (u >= 2 ? 128 : 0), tail vt; padded image: vb + 64 u.  They must be the swizzle functions, for every lane, tile and k-step."""
    for lane in range(64):
        lq, lg = lane & 15, lane >> 4
        ke, kt = kswz(key_of(0, lq), lg), kswz(128 + lq, lg)
        for j in range(9):
            got = kt if j == 8 else ((ke ^ 32) if j & 1 else ke) + 2048 * (j >> 1) + 256 * (j & 1)
            assert got == kswz(key_of(j, lq), lg), (lane, j)
        for dt in range(2):
            vb = vt_off(dt * 16 + lq, 8 * lg)
            for u in range(4):
                assert ((vb ^ 64) if u & 1 else vb) + (128 if u >= 2 else 0) == vt_off(dt * 16 + lq, 32 * u + 8 * lg), (lane, dt, u)
            pad = lambda d, key: d * 336 + key * 2                  # vt_off<false>
            for u in range(4):
                assert pad(dt * 16 + lq, 8 * lg) + 64 * u == pad(dt * 16 + lq, 32 * u + 8 * lg)


@pytest.mark.parametrize("base", [0, 64, 128, 128 + 96, 128 + 288])
def test_gemm_ln_bf16_ring_images_conflict_free(base):
    """Round 6, csrc/gemm_ln_bf16.hip: the operand ring of the persistent projection + LayerNorm kernel.  K-steps of 32 channels:
    64-byte rows, chunk ^ F[(row >> 2) & 3] (`kswz64`); K-steps of 64: 128-byte rows, chunk ^ ((row >> 1) & 7) (`swz`).  A fragment
    read = 16 consecutive rows (lane & 15) x one 16-B chunk per 16-lane group: conflict-free under the real ds_read_b128 lane groups
    for every wave tile's base row; the DMA fills whole 1-KB runs linearly (the swizzle is on the source side)."""
    kswz64 = lambda row, chunk: row * 64 + ((chunk ^ F[(row >> 2) & 3]) << 4)
    swz = lambda row, chunk: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
    assert conflict_free("read_b128", lambda l: kswz64(base + (l & 15), l >> 4))
    for kk in range(2):
        assert conflict_free("read_b128", lambda l: swz(base + (l & 15), kk * 4 + (l >> 4)))
    # the images are bijective over a stage's rows
    assert len({kswz64(r, c) for r in range(512) for c in range(4)}) == 2048
    assert len({swz(r, c) for r in range(512) for c in range(8)}) == 4096

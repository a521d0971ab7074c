"""Host-side packing of the fused-kernel weight images (CPU): the image is read back exactly the way the HIP kernel's
lanes address it (csrc/mlp_fused_bf16.hip) and must reproduce the plain MLP, and every fragment read must be
conflict-free under the gfx950 ds_read_b128 bank rules."""
import numpy as np
import pytest
import torch

import pangu_pytorch_amd as P
from pangu_pytorch_amd import ops_bf16 as ob


def _b128_groups():
    """Lane groups of one ds_read_b128 wave-instruction (MI355X_MICROARCH.md, LDS table)."""
    g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    return [g0, g1, [l + 32 for l in g0], [l + 32 for l in g1]]


def _conflict_free(byte_addr_of_lane):
    for grp in _b128_groups():
        slots = {(byte_addr_of_lane(l) // 16) % 16 for l in grp}
        if len(slots) != 16:
            return False
    return True


@pytest.mark.parametrize("C", [192, 384])
def test_mlp_image_matches_kernel_addressing(C):
    HID, nch, KS, RT = 4 * C, 4 * C // 32, C // 16, C // 32
    fsw = (lambda r: r & 15) if C == 384 else (lambda r: (r >> 1) & 7)
    g = torch.Generator().manual_seed(C)
    w1 = torch.randn(HID, C, generator=g)
    w2 = torch.randn(C, HID, generator=g)
    img = ob.pack_mlp_weights(w1, w2)
    assert img.shape == (2, nch, 32 * C) and img.dtype == torch.bfloat16
    im = img.float().numpy()
    w1b, w2b = w1.to(torch.bfloat16).float().numpy(), w2.to(torch.bfloat16).float().numpy()
    for ch in (0, 1, nch - 1):
        for lane in range(64):
            lr, lh = lane & 31, lane >> 5
            for ks in range(KS):       # first product: A fragment = W1[32ch + lr][16ks + 8lh .. +7]
                off = (lr * 2 * C + (((2 * ks + lh) ^ fsw(lr)) << 4)) // 2
                np.testing.assert_array_equal(im[0, ch, off:off + 8], w1b[32 * ch + lr, 16 * ks + 8 * lh:16 * ks + 8 * lh + 8])
            for rt in range(RT):       # second product: k-step s, element j <-> hidden 16s + 8(j>>2) + 4lh + (j&3)
                for sk in range(2):
                    off = (((sk * 2 + lh) * C + 32 * rt + lr) * 16) // 2
                    hid = [32 * ch + 16 * sk + 8 * (j >> 2) + 4 * lh + (j & 3) for j in range(8)]
                    np.testing.assert_array_equal(im[1, ch, off:off + 8], w2b[32 * rt + lr, hid])
    # bank conflicts of the fragment reads
    for ks in range(KS):
        assert _conflict_free(lambda l: (l & 31) * 2 * C + (((2 * ks + (l >> 5)) ^ fsw(l & 31)) << 4))
    for rt in range(RT):
        for sk in range(2):
            assert _conflict_free(lambda l: ((sk * 2 + (l >> 5)) * C + 32 * rt + (l & 31)) * 16)


def test_shadow_cache_revalidates_on_data_swap_and_is_not_copied():
    """`param.data = w` (reference models/onnx2torch.py:37-52) must not serve a stale bf16 shadow, and the
    shadow cache must not travel with deepcopy / pickle."""
    import copy
    import pickle
    from pangu_pytorch_amd import fused_bf16
    sh = fused_bf16.WeightShadow()
    p = torch.nn.Parameter(torch.ones(8, 8))
    a = sh.get(p)
    assert sh.get(p) is a
    p.data = torch.full((8, 8), 2.0)
    b = sh.get(p)
    assert b is not a and float(b[0, 0]) == 2.0
    with torch.no_grad():
        p.mul_(2.0)
    assert float(sh.get(p)[0, 0]) == 4.0
    m = P.PanguModel(depths=[1, 1, 1, 1])
    m._shadow = fused_bf16.WeightShadow()
    m._shadow.get(m.downsample.linear.weight)
    assert len(m._shadow.cache) == 1
    assert copy.deepcopy(m)._shadow is None
    assert pickle.loads(pickle.dumps(m))._shadow is None
    assert len(m._shadow.cache) == 1                      # the original keeps its own
    m.load_state_dict(m.state_dict())
    assert len(m._shadow.cache) == 0

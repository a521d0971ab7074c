"""CPU-side checks of the drop-in boundary: the C ABI library exports what include/pangu_hip.h declares, the
Python binding table matches the header, and the nn.Module surface matches the reference's contract."""
import copy
import ctypes
import json
import os
import pickle

import pytest
import torch

import pangu_pytorch_amd as P
from pangu_pytorch_amd import _lib


def test_header_vs_binding_table():
    assert _lib.header_functions() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpangu_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.header_functions():
        assert hasattr(lib, name), name
    lib.pangu_abi_version.restype = ctypes.c_int
    assert lib.pangu_abi_version() == 1


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (no GPU needed)."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libpangu_hip.so not built")
    lib = _lib.load()
    assert lib.pangu_linear_fwd(None, None, 0, None, None, None, 0, 1, 1, 16, 0, None) == -2    # NULL
    assert lib.pangu_linear_fwd(None, 8, 16, 8, None, 8, 8, 4, 8, 8, 0, None) == -1              # K % 16
    assert lib.pangu_linear_fwd_bf16(None, 8, 16, 8, None, 8, 8, 4, 8, 12, 0, None, 1) == -1     # K % 8
    assert lib.pangu_linear_fwd_bf16(None, 8, 16, 8, None, 8, 8, 4, 8, 16, 0, None, 7) == -3     # dtype tag
    assert lib.pangu_window_attn_fwd(None, 8, 8, 8, 8, None, 8, 181, 25, 192, 6, 0) == -1         # W % 12
    assert lib.pangu_window_attn_fwd(None, 8, 8, 8, 8, None, 8, 181, 24, 192, 5, 0) == -1         # C != 32*heads
    assert lib.pangu_error_string(-1) == b"unsupported shape"
    # round 3 entries: NULL / shape / range checks happen before any launch
    P8 = 8          # any non-NULL address: the calls below return before touching memory
    assert lib.pangu_mlp_ln_residual_train_fwd_bf16(None, P8, 192, P8, P8, P8, P8, P8, P8, 192, P8, 768, None, 192, 64, 192, 1.0) == -2   # m is required
    assert lib.pangu_mlp_ln_residual_train_fwd_bf16(None, P8, 192, P8, P8, P8, P8, P8, P8, 192, P8, 760, P8, 192, 64, 192, 1.0) == -1    # ldp < 4C
    assert lib.pangu_mlp_ln_residual_train_fwd_bf16(None, P8, 256, P8, P8, P8, P8, P8, P8, 256, P8, 1024, P8, 256, 64, 256, 1.0) == -1   # C not 192 / 384
    assert lib.pangu_mlp_ln_residual_train_fwd_bf16(None, P8, 192, P8, P8, P8, P8, P8, P8, 192, None, 768, P8, 192, 64, 192, 1.0) == -2   # pre is required
    assert lib.pangu_linear_gelu_bwd_bf16(None, P8, 192, P8, P8, 768, 64, 768, 192, None, None) == -2                                   # pre is required
    assert lib.pangu_linear_gelu_bwd_bf16(None, P8, 192, P8, P8, 760, 64, 768, 192, P8, None) == -1                                    # ldc < N
    assert lib.pangu_attn_windows_fwd(None, P8, P8, None, 0, P8, 2, 124, 5, 192) == -1                                                 # C != 32 * heads
    assert lib.pangu_attn_windows_fwd(None, P8, P8, P8, 7, P8, 2, 124, 6, 192) == -4                                                   # mask stride
    assert lib.pangu_patch_recover_scatter_denorm(None, P8, P8, P8, P8, P8, P8, P8, P8, P8, None, 721, 1440) == -2
    assert lib.pangu_patch_recover_scatter_denorm(None, P8, P8, P8, P8, P8, P8, P8, P8, P8, P8, 721, 1442) == -1                       # LON % 4
    # 32-bit buffer offsets of the attention backward kernels: a grid whose (n_tok x 3C) tensor reaches 2 GiB is refused
    big_w = 12 * 60            # 8 x 181 x 720 tokens x 576 channels x 4 B = 2.4 GB
    assert lib.pangu_window_attn_bwd(None, P8, P8, P8, P8, P8, P8, P8, P8, P8, 8, 181, big_w, 192, 6, 0) == -5
    assert lib.pangu_window_attn_bwd_bf16(None, P8, P8, P8, P8, P8, P8, P8, P8, P8, 8, 181, 2 * big_w, 192, 6, 0) == -5
    assert lib.pangu_error_string(-5) is not None
    assert lib.pangu_adam_step_multi(None, None, 3, 10, 1e-3, 0.9, 0.999, 0.0, 1e-8, 0.1, 0.03) == -2
    assert lib.pangu_adam_step_multi(None, P8, 3, 10, 1e-3, 1.0, 0.999, 0.0, 1e-8, 0.1, 0.03) == -4          # beta1 = 1
    # blocks never straddle a (sample, variable, level) plane: (5 x 13 + 4) planes of 721 x 1440 -> 127 blocks each
    assert lib.pangu_weighted_l1_loss_blocks(1, 5, 13 * 721 * 1440, 4, 721 * 1440, 13) == (5 * 13 + 4) * 127
    assert lib.pangu_weighted_l1_loss_blocks(1, 5, 13 * 721 * 1440, 4, 721 * 1440, 11) == -1         # plane_u % levels != 0
    assert lib.pangu_weighted_l1_loss_fwd(None, P8, P8, P8, P8, P8, P8, None, P8, 1, 5, 100, 4, 100, 1, 0, None, None, None, None) == -2
    assert lib.pangu_weighted_l1_loss_bwd(None, P8, P8, P8, P8, P8, P8, P8, P8, P8, 0, 5, 100, 4, 100, 1, 0, None, None, None, None) == -1
    assert lib.pangu_weighted_l1_loss_fwd(None, P8, P8, P8, P8, P8, P8, P8, P8, 1, 5, 100, 4, 100, 1, 0, P8, None, None, None) == -2   # statistics: all four or none
    assert lib.pangu_patch_embed_gather_bwd(None, P8, P8, P8, P8, P8, None, 721, 1440, 0) == -2
    assert lib.pangu_patch_embed_gather_bwd(None, P8, P8, P8, P8, P8, P8, 721, 1442, 0) == -1
    # the host side of the input pipeline is plain host code: callable (and checked) without a GPU
    import numpy as np
    src = np.arange(3 * (1 << 20) + 13, dtype=np.float32)              # 12 MB + a ragged tail: three spans on three threads
    dst = np.zeros_like(src)
    assert lib.pangu_host_copy(dst.ctypes.data, src.ctypes.data, src.nbytes, 3) == 0 and np.array_equal(dst, src)
    dst[:] = 0
    assert lib.pangu_host_copy(dst.ctypes.data, src.ctypes.data, 1000, 64) == 0 and np.array_equal(dst[:250], src[:250]) and dst[250] == 0
    assert lib.pangu_host_copy(dst.ctypes.data, None, 16, 1) == -2 and lib.pangu_host_copy(None, None, 0, 1) == 0
    assert lib.pangu_host_copy(dst.ctypes.data, src.ctypes.data, -1, 1) == -1
    assert lib.pangu_shadow_refresh_bf16(None, None, 3, 10) == -2
    assert lib.pangu_shadow_refresh_bf16(None, P8, 0, 10) == -1
    assert lib.pangu_shadow_refresh_bf16(None, P8, 3, 1 << 31) == -1           # more blocks than a grid dimension holds


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    return P.PanguModel(device="cpu")


def test_state_dict_keys_and_shapes(model, golden_dir):
    ks = json.load(open(os.path.join(golden_dir, "keys_shapes.json")))
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert sd == ks["state_dict"] and len(sd) == 223
    assert sum(p.numel() for p in model.parameters()) == ks["n_params"] == 276659936
    assert [k for k, _ in model.named_parameters()] == ks["named_parameters_order"]


def test_strict_load_state_dict_roundtrip(model):
    sd = {k: torch.zeros_like(v) for k, v in model.state_dict().items()}
    m2 = P.PanguModel(device="cpu")
    missing = m2.load_state_dict({"model": sd}["model"], strict=True)       # reference test_main.py:64-65
    assert not missing.missing_keys and not missing.unexpected_keys


def test_module_is_deepcopy_and_pickle_safe(model):
    blk = model.layers[0].blocks[0]
    assert pickle.loads(pickle.dumps(blk)).attention.head_number == 6      # reference pangu_sample.py:162-164
    m2 = copy.deepcopy(model.layers[1])
    assert m2.blocks[1].attention.earth_specific_bias.shape == (1, 64, 12, 144, 144)


def test_init_matches_reference_policy(model):
    blk = model.layers[0].blocks[1]
    assert float(blk.norm1.weight.min()) == 1.0 and float(blk.norm1.bias.abs().max()) == 0.0
    assert float(blk.linear.linear1.bias.abs().max()) == 0.0
    assert 0.015 < float(blk.linear.linear1.weight.std()) < 0.025
    assert 0.015 < float(blk.attention.earth_specific_bias.std()) < 0.025
    dpr = [b.drop_path.drop_prob if hasattr(b.drop_path, "drop_prob") else 0.0
           for l in model.layers for b in l.blocks]
    assert dpr[0] == 0.0 and abs(dpr[-1] - 0.2) < 1e-6 and len(dpr) == 16


def test_cpu_tensors_fail_loudly(model):
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        model(torch.zeros(1, 5, 13, 721, 1440), torch.zeros(1, 4, 721, 1440),
              (torch.zeros(4), torch.ones(4), torch.zeros(13, 1, 1, 5), torch.ones(13, 1, 1, 5)),
              torch.zeros(1, 3, 724, 1440), torch.zeros(1, 1, 1, 13, 721, 1440))

"""Closed-form synthetic data generator (TEST INFRASTRUCTURE, shared by oracle, tests and bench).

Every tensor used for golden vectors is a pure function of (seed, flat element index) through a
splitmix64 hash, so the reference-side script (oracle/gen_golden.py, runs only where /root/reference
exists) and the GPU-box tests regenerate bit-identical inputs without shipping them and without
depending on any library RNG stream.  Integer hashing is done in wrapping int64 torch arithmetic and the
float mapping in IEEE float64, so CPU and GPU produce the same bits.

Values are uniform in [-sqrt(3), sqrt(3)) (unit variance, zero mean) times `scale` plus `shift`.
"""
import math
import torch

_GOLD = -7046029254386353131   # 0x9E3779B97F4A7C15 as int64
_M1 = -4658895280553007687     # 0xBF58476D1CE4E5B9
_M2 = -7723592293110705685     # 0x94D049BB133111EB
_SQRT3 = 1.7320508075688772


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _splitmix64(x):
    x = x + _GOLD
    x = (x ^ _lsr(x, 30)) * _M1
    x = (x ^ _lsr(x, 27)) * _M2
    return x ^ _lsr(x, 31)


def _wrap(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def name_seed(name):
    """Stable 63-bit seed from a string (FNV-1a)."""
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFFFFFFFFFF


def _base(seed, device):
    return _splitmix64(torch.tensor([_wrap(int(seed))], dtype=torch.int64, device=device))


def uniform(shape, seed, scale=1.0, shift=0.0, device="cpu", dtype=torch.float32, chunk=1 << 24):
    """Tensor of `shape`; element i = f(seed, i). Unit variance before scale."""
    shape = tuple(int(s) for s in shape)
    n = 1
    for s in shape:
        n *= s
    out = torch.empty(n, dtype=dtype, device=device)
    base = _base(seed, device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i = torch.arange(s, e, dtype=torch.int64, device=device)
        h = _splitmix64(i * _GOLD + base)
        u = _lsr(h, 40).to(torch.float64) * (1.0 / (1 << 24))  # [0,1), 24 bits
        out[s:e] = ((u * 2.0 - 1.0) * (_SQRT3 * scale) + shift).to(torch.float32).to(dtype)
    return out.reshape(shape)


def sample_positions(numel, count, seed, device="cpu"):
    """`count` pseudo-random flat positions in [0, numel) (int64, may repeat)."""
    i = torch.arange(count, dtype=torch.int64, device=device)
    h = _splitmix64(i * _GOLD + _base(int(seed) ^ 0x5A5A5A5A, device))
    return _lsr(h, 1) % int(numel)


def param_spec(name, shape):
    """(scale, shift) of the synthetic value for a model parameter, keyed by its state_dict name.

    Chosen so that every term matters numerically (non-zero biases, non-unit LayerNorm gains):
      LayerNorm weight : 1 + 0.1 u        LayerNorm bias : 0.1 u
      earth_specific_bias : 0.5 u  (strong enough to shape the softmax)
      linear/conv weights : u / sqrt(fan_in)   (variance preserving)
      linear/conv biases  : 0.1 u
    """
    if name.endswith("earth_specific_bias"):
        return 0.5, 0.0
    is_norm = ".norm" in name
    if name.endswith(".weight"):
        if is_norm:
            return 0.1, 1.0
        return 1.0 / math.sqrt(shape[1]), 0.0
    if name.endswith(".bias"):
        return 0.1, 0.0
    raise ValueError(name)


def param_spec_refinit(name, shape):
    """(scale, shift) in the REFERENCE's initialisation scales (models/pangu_model.py:41-48, layers.py:314): Linear / Conv1d
    weights and the Earth-specific bias tables std 0.02 (trunc-normal there; the closed-form uniform of the same std here, so
    both sides of a test regenerate them), LayerNorm (1, 0), biases 0 -- the contractive regime real checkpoints live in, where
    bf16 errors are not amplified from block to block as under `param_spec`'s O(1) weights."""
    if name.endswith("earth_specific_bias"):
        return 0.02, 0.0
    if name.endswith(".weight"):
        return (0.0, 1.0) if ".norm" in name else (0.02, 0.0)
    if name.endswith(".bias"):
        return 0.0, 0.0
    raise ValueError(name)


def synth_param(name, shape, device="cpu", dtype=torch.float32, spec="golden"):
    scale, shift = (param_spec_refinit if spec == "refinit" else param_spec)(name, shape)
    return uniform(shape, name_seed(name), scale, shift, device=device, dtype=dtype)


def fill_state_dict(shapes, device="cpu", dtype=torch.float32, spec="golden"):
    """shapes: {name: shape}. Returns {name: tensor} with synthetic values (spec: "golden" = param_spec, "refinit")."""
    return {k: synth_param(k, tuple(v), device, dtype, spec) for k, v in shapes.items()}

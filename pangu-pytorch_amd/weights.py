"""Weight import: checkpoints in the reference's formats -> the drop-in model.

* `load_checkpoint`: the `{'model': state_dict, ...}` .pth of reference finetune_fully.py:115-116 / test_main.py:64-65
  (or a bare state_dict), strict.
* `load_onnx_initializers`: the semantics of the reference's models/onnx2torch.py:23-52 — look every parameter up in a
  torch_name -> onnx_name table (the two columns of keys_all.csv), copy 1-/3-/5-D arrays as they are and 2-D MatMul
  initialisers TRANSPOSED (ONNX stores (in,out), nn.Linear (out,in)), with the same shape assertions.
* `expand_bias` / `compact_bias`: the paper's compact Earth-specific bias table (3312, types, heads) <-> the expanded
  (1, types, heads, 144, 144) parameter the ONNX export (and this model) uses, through `position_index`
  (reference layers.py:319-357 builds the index, :384-391 shows the gather, commented out)."""
import torch

WTOK = 144


def load_checkpoint(model, ckpt, strict=True, map_location=None):
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "read"):
        ckpt = torch.load(ckpt, map_location=map_location, weights_only=True)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict) else ckpt
    return model.load_state_dict(sd, strict=strict)


def load_onnx_initializers(model, onnx_weights, key_table, freeze=False):
    """onnx_weights: {onnx_name: array-like}; key_table: {torch_name: onnx_name} (keys_all.csv).  Returns the list of
    parameters that had no table entry.  freeze=True mirrors onnx2torch.py's requires_grad=False."""
    missing = []
    with torch.no_grad():
        for name, p in model.named_parameters():
            onnx_name = key_table.get(name)
            if not isinstance(onnx_name, str) or onnx_name not in onnx_weights:
                missing.append(name)
                continue
            w = torch.as_tensor(onnx_weights[onnx_name])
            if p.dim() == 2:
                if tuple(p.shape) != tuple(w.t().shape):
                    raise ValueError(f"{name}: {tuple(p.shape)} vs transposed ONNX {tuple(w.t().shape)}")
                w = w.t()
            elif tuple(p.shape) != tuple(w.shape):
                raise ValueError(f"{name}: {tuple(p.shape)} vs ONNX {tuple(w.shape)}")
            p.copy_(w.to(p.dtype))
            if freeze:
                p.requires_grad = False
    return missing


def position_index(device="cpu"):
    """int64 (20736,) in [0, 3312): reference layers.py:319-357 in closed form (bit-exact, tests/test_weights.py)."""
    n = torch.arange(WTOK, device=device)
    zi, hi, wi = n // 72, (n // 12) % 6, n % 12
    dz = zi.view(-1, 1) + zi.view(1, -1) * 2
    dh = hi.view(-1, 1) + hi.view(1, -1) * 6
    dw = wi.view(-1, 1) - wi.view(1, -1) + 11
    return (dz * 23 * 36 + dh * 23 + dw).flatten()


def expand_bias(compact):
    """(3312, types, heads) -> (1, types, heads, 144, 144): bias[position_index] reshaped/permuted as layers.py:384-391."""
    idx = position_index(compact.device)
    types, heads = compact.shape[1], compact.shape[2]
    return compact[idx].view(WTOK, WTOK, types, heads).permute(2, 3, 0, 1).unsqueeze(0).contiguous()


def compact_bias_table(expanded):
    """(1, types, heads, 144, 144) -> the kernel-side compact table (types, heads, 3312) fp32, or None when the expanded
    tensor is NOT an expansion of a compact table (e.g. the reference's trunc-normal initialisation of the expanded
    parameter, layers.py:306-314): every entry must equal the entries it shares a position index with, bit for bit.  The
    published weights are expansions (the paper trains the compact table; the ONNX export expanded it)."""
    idx = position_index(expanded.device)
    types, heads = expanded.shape[1], expanded.shape[2]
    e = expanded[0].reshape(types, heads, WTOK * WTOK)
    first = torch.zeros(3312, dtype=torch.long, device=e.device)
    first.scatter_(0, idx.flip(0), torch.arange(WTOK * WTOK - 1, -1, -1, device=e.device))     # first position of every index
    table = e[:, :, first].contiguous()
    if not torch.equal(table[:, :, idx], e):
        return None
    return table.float().contiguous()


def compact_bias(expanded):
    """(1, types, heads, 144, 144) -> (3312, types, heads): mean over the entries that share a position index (exact
    inverse of expand_bias; for a trained expanded table it is the least-squares compact fit).  6.3x fewer bytes."""
    idx = position_index(expanded.device)
    e = expanded[0].permute(2, 3, 0, 1).reshape(WTOK * WTOK, expanded.shape[1], expanded.shape[2])
    out = torch.zeros((3312,) + tuple(e.shape[1:]), dtype=e.dtype, device=e.device)
    cnt = torch.zeros(3312, dtype=e.dtype, device=e.device)
    out.index_add_(0, idx, e)
    cnt.index_add_(0, idx, torch.ones_like(idx, dtype=e.dtype))
    return out / cnt.view(-1, 1, 1)

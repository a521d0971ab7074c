"""Data-parallel training step of the REAL model at world size 2 on one GPU (gloo backend, both ranks on cuda:0, 2 x 66 GB
fp32): the bucketed flat-buffer all-reduce issued from the backward hooks, with DropPath on and a different sample and
different DropPath draws per rank, must equal the single-process mean of the two per-sample gradients -- the semantics of
the reference's gather_grad (era5_data/utils_dist.py:125-134: SUM then / world).  Also checks that the Earth-specific bias
gradients (94 % of the 1.107 GB) reach the flat buffer without the copy pass."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dtype,world,mode", [("f32", 2, "all_reduce"), ("bf16", 2, "all_reduce"), ("bf16", 4, "all_reduce"),
                                              ("bf16", 2, "reduce_scatter"), ("bf16", 4, "reduce_scatter")])
def test_n_rank_train_step_matches_single_process_mean(tmp_path, dtype, world, mode):
    """world 4: four ranks of the bf16 model on one GPU (4 x 37 GB), DropPath on.  mode:
    FlatGradSync's bucketed all_reduce, or reduce_scatter + all_gather per bucket in place."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "helpers", "dp2_worker.py"), str(tmp_path), dtype, mode]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(os.path.join(tmp_path, "dp2.pt"))
    flat, info = got["flat"], got["info"]
    # the bias tables were written in place: what the copy fallback moved is the small accumulated tensors only
    assert info["copied_bytes"] < 0.08 * info["flat_bytes"], info
    # every bucket's all-reduce was issued from inside backward, in order -- also with DropPath-dropped branches (rank 0
    # drops six here): autograd hands the hooks materialised zero gradients for a Function's None outputs, so no bucket stalls
    assert info["launched_in_backward"] == info["buckets"] == 20, info
    assert info["order"][0].startswith("_output_layer") and "EarthSpecificLayer3.blocks.EarthSpecificBlock1" in info["order"][1]
    # single-process reference: the two per-sample backward passes, same seeds, no sync, then the mean
    import dp2_worker as W
    import cases
    import synth
    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import dist as D
    model = P.PanguModel(device="cuda").cuda().train()
    model.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    model.set_compute_dtype({"f32": torch.float32, "bf16": torch.bfloat16}[dtype])
    order = [p for b in D.default_buckets(model) for p in b if p.requires_grad]
    # strictly equal collective order on every rank, all of it issued from inside backward
    per_rank = [torch.load(os.path.join(tmp_path, f"dp_info_r{r}.pt")) for r in range(world)]
    assert all(i["launches"] == list(range(20)) and i["launched_in_backward"] == 20 for i in per_rank), \
        [(i["launches"], i["launched_in_backward"]) for i in per_rank]
    assert len({tuple(map(tuple, i["pattern"])) for i in per_rank}) > 1      # the ranks did draw different DropPath patterns
    assert info["mode"] == mode
    names = {id(p): n for n, p in model.named_parameters()}
    offs = info["offsets"]                      # parameter name -> offset in the flat buffer (bucket ends are zero-padded)
    mean = torch.zeros_like(flat)
    for rank in range(world):
        model.zero_grad(set_to_none=True)
        W.one_backward(model, rank)
        for p in order:
            off = offs[names[id(p)]]
            g = p.grad.float().flatten().cpu() if p.grad is not None else torch.zeros(p.numel())
            mean[off:off + p.numel()] += g / world
    # identical kernels on both sides; the weight-gradient kernels accumulate with fp32 atomics (order-dependent last bits)
    worst = []
    for p in order:
        off = offs[names[id(p)]]
        d = (flat[off:off + p.numel()] - mean[off:off + p.numel()]).abs().max().item()
        worst.append((d, names[id(p)], mean[off:off + p.numel()].abs().max().item(), flat[off:off + p.numel()].abs().max().item()))
    worst.sort(reverse=True)
    print("dp2 worst parameters (abs err, name, max |mean|, max |flat|):", worst[:4])
    err = (flat - mean).abs().max().item() / mean.abs().max().item()
    l2 = ((flat - mean).norm() / mean.norm()).item()
    print(f"dp{world} {dtype}: max err rel to max |g| {err:.2e}, rel-L2 {l2:.2e}, copied {info['copied_bytes'] / 2**20:.1f} MiB of "
          f"{info['flat_bytes'] / 2**20:.0f} MiB; rank 0 dropped {info['dropped_branches']} branches, "
          f"{info['launched_in_backward']}/{info['buckets']} buckets launched inside backward")
    assert err < 1e-4 and l2 < 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_flat_grad_sync_batch2_matches_per_sample_mean(dtype):
    """per-GPU batch B = 2 (the reference's BATCH_SIZE // n_gpus, finetune_fully.py:77) calls the block
    function once per sample, so every Earth-specific bias table feeds TWO autograd nodes of one backward.  With
    FlatGradSync registered the flat buffer must hold g(sample 0) + g(sample 1) of the batch loss -- the slot is handed to
    the first node only (ops.grad_slot), the second node's gradient is added by autograd."""
    import dp2_worker as W
    import cases
    import synth
    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import dist as D, train
    model = P.PanguModel(device="cuda").cuda().eval()      # autograd path, DropPath off (its draw ORDER differs between
    # a B = 2 batch -- per block -- and two B = 1 passes -- per sample)
    model.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    model.set_compute_dtype({"f32": torch.float32, "bf16": torch.bfloat16}[dtype])
    s0, s1 = W.sample(0), W.sample(1)
    stats, maps, const_h = s0[4:]
    order = [p for b in D.default_buckets(model) for p in b if p.requires_grad]
    # reference: the two samples one at a time (B = 1 each, no FlatGradSync), each with HALF the loss -- what the batch-mean
    # loss hands each sample.  (Halving afterwards instead is not bit-equivalent in bf16: a handful of gradient terms sit at
    # the bottom of the exponent range -- gelu'(-20) * 1e-7, softmax weights of masked keys -- and flush differently at the
    # two scales; one flipped bf16 rounding in a (N, 3C) tensor is then amplified 2-3x per block by the deliberately
    # non-contractive golden weights: 3e-3 at the first block.  Same scale on both sides: equal to the last bit.)
    want = torch.zeros(sum(p.numel() for p in order))
    for smp in (s0, s1):
        model.zero_grad(set_to_none=True)
        out, out_s = model(smp[0], smp[1], stats, maps, const_h)
        (train.weighted_l1_loss(out, out_s, smp[2], smp[3]) * 0.5).backward()
        off = 0
        for p in order:
            if p.grad is not None:
                want[off:off + p.numel()] += p.grad.float().flatten().cpu()
            off += p.numel()
    model.zero_grad(set_to_none=True)
    sync = D.FlatGradSync(model)
    try:
        cat = lambda a, b: torch.cat((a, b), 0)
        out, out_s = model(cat(s0[0], s1[0]), cat(s0[1], s1[1]), stats, maps, const_h)
        train.weighted_l1_loss(out, out_s, cat(s0[2], s1[2]), cat(s0[3], s1[3])).backward()
        sync.finish()
        torch.cuda.synchronize()
        got = torch.cat([sync._slot[p][1].flatten() for p in order]).cpu()      # (bucket ends of the flat buffer are padded)
    finally:
        sync.remove()
    # per bias table (the slot-written tensors) and overall
    names = {id(p): n for n, p in model.named_parameters()}
    off, worst = 0, (0.0, "")
    for p in order:
        if p.dim() == 5:
            a, b = got[off:off + p.numel()], want[off:off + p.numel()]
            worst = max(worst, (((a - b).norm() / b.norm().clamp_min(1e-30)).item(), names[id(p)]))
        off += p.numel()
    l2 = ((got - want).norm() / want.norm()).item()
    print(f"B=2 FlatGradSync {dtype}: rel-L2 {l2:.2e}; worst bias table {worst}")
    assert l2 < 1e-5 and worst[0] < 1e-6      # identical kernels at identical scales: only the fp32 atomics of the small tensors differ

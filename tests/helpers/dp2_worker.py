"""Worker of tests/test_gpu_dp2.py: one rank of an N-rank data-parallel training step of the REAL model on one GPU
(gloo backend, all ranks share cuda:0).  Usage: torchrun --nproc-per-node N dp2_worker.py <out_dir> <dtype> [all_reduce|reduce_scatter]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cases  # noqa: E402
import synth  # noqa: E402
import pangu_pytorch_amd as P  # noqa: E402
from pangu_pytorch_amd import dist as D, train  # noqa: E402


def sample(rank):
    inp, inp_s, stats, maps, const_h = cases.model_inputs("cuda")
    g = lambda n, s: synth.uniform(s, synth.name_seed(n), device="cuda")
    if rank:      # a different sample per rank
        inp, inp_s = g(f"input_r{rank}", inp.shape), g(f"input_surface_r{rank}", inp_s.shape)
    tgt, tgt_s = g(f"target_r{rank}", inp.shape), g(f"target_surface_r{rank}", inp_s.shape)
    return inp, inp_s, tgt, tgt_s, stats, maps, const_h


def one_backward(model, rank, sync=None):
    """forward + backward of rank `rank`'s sample with that rank's DropPath draws (host RNG seeded per rank)."""
    inp, inp_s, tgt, tgt_s, stats, maps, const_h = sample(rank)
    torch.manual_seed(4321 + rank)
    out, out_s = model(inp, inp_s, stats, maps, const_h)
    train.weighted_l1_loss(out, out_s, tgt, tgt_s).backward()
    if sync is not None:
        sync.launched_in_backward = sync._next       # buckets whose all-reduce was issued from the backward hooks
        sync.finish()


def main():
    out_dir, dtype = sys.argv[1], {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[2]]
    torch.cuda.set_device(0)
    D.init_dist("pytorch", backend="gloo")
    rank, world = D.get_dist_info()
    model = P.PanguModel(device="cuda").cuda().train()            # train(): DropPath ON (rates up to 0.2)
    model.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), "cuda"))
    model.set_compute_dtype(dtype)
    sync = D.FlatGradSync(model, mode=sys.argv[3] if len(sys.argv) > 3 else "all_reduce")
    launches = []                       # bucket indices in the order this rank issued their collectives
    launch = sync._launch
    sync._launch = lambda bi: (launches.append(bi), launch(bi))[1]
    one_backward(model, rank, sync)
    torch.cuda.synchronize()
    names = {id(p): n for n, p in model.named_parameters()}
    assert all(p.grad.data_ptr() == sync._slot[p][1].data_ptr() for p in model.parameters())
    total = sync.flat.numel() * 4
    dropped = sum(getattr(m, "n_dropped", 0) for m in model.modules())
    pattern = [(n, m.n_dropped) for n, m in model.named_modules() if getattr(m, "n_dropped", 0)]
    info = {"copied_bytes": sync.copied_bytes, "flat_bytes": total, "launched_in_backward": sync.launched_in_backward,
            "buckets": len(sync.buckets), "dropped_branches": dropped, "pattern": pattern,
            "order": [names[id(b[2][0][0])] for b in sync.buckets][:3], "launches": launches, "world": world,
            "peak_gb": torch.cuda.max_memory_allocated() / 2**30, "mode": sync.mode,
            "offsets": {names[id(p)]: sync._slot[p][1].storage_offset() for p in model.parameters()}}      # (buckets are padded)
    torch.save(info, os.path.join(out_dir, f"dp_info_r{rank}.pt"))
    if rank == 0:
        torch.save({"flat": sync.flat.cpu(), "info": info}, os.path.join(out_dir, "dp2.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

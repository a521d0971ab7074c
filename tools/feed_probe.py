#!/usr/bin/env python3
"""Per-step account of the host-fed training loop (tools/feed_probe.py [f32|bf16] [steps]): what the training thread waited for a
batch, the step's wall time, the copy engine's time for the batch -- to see WHERE a fed step loses against the resident one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import pangu_pytorch_amd as P
from pangu_pytorch_amd import data, train
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = P.PanguModel(device=dev).to(dev).train()
model.set_compute_dtype(dt)
_, _, stats, maps, const_h = bench.synthetic_inputs(dev, 1000)
opt = train.make_optimizer(model)
g = torch.Generator().manual_seed(3000)
shapes = ((1, 5, 13, 721, 1440), (1, 4, 721, 1440), (1, 5, 13, 721, 1440), (1, 4, 721, 1440))
host = [tuple(torch.rand(sh, generator=g) * 2 - 1 for sh in shapes) for _ in range(3)]
res = [tuple(t.to(dev) for t in b) for b in host]
for mode in ("resident", "fed", "fed_noitem"):
    torch.manual_seed(1)          # the same DropPath draws in every mode
    rows = []
    if mode == "resident":
        it = iter([res[i % 3] for i in range(n)])
        pf = None
    else:
        pf = data.DevicePrefetcher([host[i % 3] for i in range(n)], dev, flip_levels=True, fuse_flip=True, depth=2, reuse_device_buffers=True)
        it = iter(pf)
    torch.cuda.synchronize()
    t_prev = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        loss = train.train_step(model, opt, b, stats, maps, const_h, levels_reversed=True)
        t2 = time.perf_counter()
        if mode != "fed_noitem":
            loss.item()
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t_prev))
        t_prev = t3
    torch.cuda.synchronize()
    print(f"== {mode} {dt}: per step ms [wait for batch | enqueue step | item() | total]")
    for r in rows:
        print("   " + "  ".join(f"{x * 1e3:7.2f}" for x in r))
    if pf is not None:
        h = [(e0.elapsed_time(e1), nb) for e0, e1, nb in pf.stats["h2d_events"]]
        print("   copy-engine ms per batch:", " ".join(f"{a:.1f}" for a, _ in h))
        print("   ", pf.summary())

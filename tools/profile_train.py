"""Run a few training steps (fwd + bwd + Adam, DropPath on) for rocprofv3 / wall-clock inspection:
  python tools/profile_train.py [bf16|f32|both] [steps] [warmup steps]
Prints the wall time per step and the GPU time per step (events around the steps): when the two differ the step is
bound by the host (launch overhead), not by the kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import pangu_pytorch_amd as P  # noqa: E402
from pangu_pytorch_amd import train  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "f32"           # f32 | bf16 | both
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mode = sys.argv[4] if len(sys.argv) > 4 else "droppath"      # droppath (the default step) | nodrop (eager, DropPath off) | graph (captured, DropPath off)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = P.PanguModel(device=dev).to(dev).train()
if mode != "droppath":
    model.eval()            # stochastic depth off (the only thing eval() changes in this model): what a captured step requires
inp, inp_s, stats, maps, const_h = bench.synthetic_inputs(dev, 1000)
tgt, tgt_s, *_ = bench.synthetic_inputs(dev, 2000)
opt = train.make_optimizer(model)
batch = (inp, inp_s, tgt, tgt_s)
for dt in ([torch.float32, torch.bfloat16] if which == "both" else [torch.bfloat16 if which == "bf16" else torch.float32]):
    model.set_compute_dtype(dt)
    if mode == "graph":
        gts = train.GraphedTrainStep(model, opt, batch, stats, maps, const_h)
        one_step = lambda: gts.step()
    else:
        one_step = lambda: train.train_step(model, opt, batch, stats, maps, const_h)
    torch.manual_seed(1234)      # bench.py seeds the DropPath draws (host RNG) the same way before its training loops
    for _ in range(warm):
        one_step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        loss = one_step()
        host += time.perf_counter() - h0
    b.record()
    torch.cuda.synchronize()
    if steps == 0:
        print("warm-up only")
        continue
    wall = (time.perf_counter() - t0) / steps * 1e3
    from pangu_pytorch_amd.layers import DropPath
    nd = [sum(m.n_dropped_branch[i] for m in model.modules() if isinstance(m, DropPath)) for i in (0, 1)]
    print(f"dropped branches so far (attention, MLP): {nd}")
    print(f"{'bf16' if dt == torch.bfloat16 else 'f32'} train ({mode}): wall {wall:.2f} ms/step, GPU (events) {a.elapsed_time(b) / steps:.2f} ms/step, "
          f"host time spent issuing a step {host / steps * 1e3:.2f} ms, loss {float(loss):.4f}")

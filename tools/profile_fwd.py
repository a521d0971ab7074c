"""Run a few forward steps (fp32 or bf16) for rocprofv3:  python tools/profile_fwd.py [bf16|f32] [steps]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import pangu_pytorch_amd as P  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = P.PanguModel(device=dev).to(dev).eval()
model.set_compute_dtype(dt)
inp, inp_s, stats, maps, const_h = bench.synthetic_inputs(dev, 1000)
with torch.no_grad():
    for _ in range(steps):
        out = model(inp, inp_s, stats, maps, const_h)
torch.cuda.synchronize()
print("ok", float(out[0].float().abs().mean()))

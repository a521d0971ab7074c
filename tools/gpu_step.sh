#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline none --no-bf16 --no-train 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', b['ms_per_step'], b['roofline']['attention']['frac'])"; }
for rep in 1 2 3; do
  cp scratch/lib_nt0.so pangu-pytorch_amd/libpangu_hip.so; run nt0
  cp scratch/lib_nt1.so pangu-pytorch_amd/libpangu_hip.so; run nt1
done

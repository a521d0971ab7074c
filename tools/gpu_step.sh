#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
for nt in 0 1; do
  PANGU_ATTN_BWD_NT=$nt timeout 600 python bench.py --steps 2 --warmup 1 --cpu-baseline none --no-bf16 --train-steps 4 > gpurun_out/bench_nt$nt.json 2> gpurun_out/bench_nt$nt.err
  python - <<PY
import json
b=json.loads(open('gpurun_out/bench_nt$nt.json').read().strip().split('\n')[-1])
print("nt=$nt", b['ms_per_step'], b['ddp_train']['ms_per_step'], b['ddp_train_bf16']['ms_per_step'])
PY
done
done

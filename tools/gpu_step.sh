#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
for v in 0 1; do
  echo "stagger=$v" >> gpurun_out/stag.log
  PANGU_BWD_LIB=libbwd_stag$v.so timeout 300 python tools/ablate_attn_bwd.py 2>&1 | grep "^C=" | cut -c1-40 >> gpurun_out/stag.log
done
done
cat gpurun_out/stag.log

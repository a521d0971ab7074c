#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do for v in 1 0; do echo "helper sched_barrier=$v"; PANGU_BWD_LIB=libbwdf_sb$v.so timeout 300 python tools/ablate_attn_bwd.py f32 2>&1 | grep "^f32" | cut -c1-230; done; done

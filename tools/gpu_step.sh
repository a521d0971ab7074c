#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/ablate_attn_bwd.py 2>&1 | tail -4 > gpurun_out/bwd_stamp.log
cat gpurun_out/bwd_stamp.log

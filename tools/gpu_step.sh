#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
timeout 900 python scratch/soak.py 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/soak.log
cat gpurun_out/soak.log

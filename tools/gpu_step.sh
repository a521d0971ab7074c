#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/profile_train.py bf16 6 2>&1 | tail -1
timeout 600 python tools/profile_train.py f32 4 2>&1 | tail -1
timeout 600 python scratch/prof_host.py 2>&1 | head -22

#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -k "wgrad or block_backward" 2>&1 | tail -2 > gpurun_out/wgf.log
timeout 300 python tools/bench_kernels.py wgrad 2>&1 | grep wgrad >> gpurun_out/wgf.log
cat gpurun_out/wgf.log

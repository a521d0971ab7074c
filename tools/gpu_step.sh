#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -k "wgrad or block_backward or full_training or full_backward" 2>&1 | tail -3 > gpurun_out/wg_test.log
timeout 300 python tools/bench_kernels.py wgrad_bf16 2>&1 | grep wgrad > gpurun_out/wg_new.log
timeout 600 python tools/profile_train.py bf16 6 2 2>&1 | tail -1 >> gpurun_out/wg_new.log
cat gpurun_out/wg_test.log gpurun_out/wg_new.log

#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16.py -x -q -k "wgrad or backward or training" 2>&1 | tail -3 > gpurun_out/wg_test.log
timeout 300 python tools/bench_kernels.py wgrad 2>&1 | grep wgrad > gpurun_out/wgf_new.log
timeout 600 python tools/profile_train.py f32 4 2 2>&1 | tail -1 >> gpurun_out/wgf_new.log
cat gpurun_out/wg_test.log gpurun_out/wgf_new.log

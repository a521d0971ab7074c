#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16.py tests/test_gpu_dp2.py -x -q 2>&1 | tail -5 > gpurun_out/t_bwd.log
timeout 600 python tools/profile_train.py both 4 2 2>&1 | tail -2 > gpurun_out/train_wall.log
cat gpurun_out/t_bwd.log gpurun_out/train_wall.log

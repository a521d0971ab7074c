#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -k "proj_mlp or mlp_ln_residual" 2>&1 | tail -8 > gpurun_out/pm_test.log
timeout 300 python tools/bench_kernels.py mlp_fused 2>&1 | grep -v amdgpu > gpurun_out/pm_bench.log
cat gpurun_out/pm_test.log gpurun_out/pm_bench.log

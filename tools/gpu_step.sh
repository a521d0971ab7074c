#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -k "linear or block or full" 2>&1 | tail -3 > gpurun_out/big_test.log
PANGU_BF16_BIG=0 timeout 300 python tools/bench_kernels.py gemm_bf16 --lib-compare 2>&1 | grep "K=1536\|K= 768" > gpurun_out/big_off.log
PANGU_BF16_BIG=1 timeout 300 python tools/bench_kernels.py gemm_bf16 2>&1 | grep "K=1536\|K= 768" > gpurun_out/big_on.log
cat gpurun_out/big_test.log gpurun_out/big_off.log gpurun_out/big_on.log

#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q -k "window_attention_bwd_bf16 or block_backward_bf16" 2>&1 | tail -5 > gpurun_out/bwd2_test.log
PANGU_ATTN_BWD_V=2 timeout 300 python tools/bench_kernels.py attn_bwd 2>&1 | grep bf16 > gpurun_out/bwd_v2.log
timeout 300 python tools/ablate_attn_bwd.py 2>&1 | tail -4 > gpurun_out/bwd_stamp.log
cat gpurun_out/bwd2_test.log gpurun_out/bwd_v2.log gpurun_out/bwd_stamp.log

#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -k "attention" 2>&1 | tail -5 > gpurun_out/bwdf_test.log
PANGU_ATTN_BWD_V=2 timeout 300 python tools/bench_kernels.py attn_bwd 2>&1 | grep "f32" > gpurun_out/bwdf_v2.log
timeout 300 python tools/ablate_attn_bwd.py f32 2>&1 | tail -4 > gpurun_out/bwdf_stamp.log
cat gpurun_out/bwdf_test.log gpurun_out/bwdf_v2.log gpurun_out/bwdf_stamp.log

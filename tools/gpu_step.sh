#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extras.py -x -q 2>&1 | tail -5 > gpurun_out/compact_test.log
timeout 300 python tools/bench_kernels.py attn 2>&1 | grep attn > gpurun_out/compact_bench.log
cat gpurun_out/compact_test.log gpurun_out/compact_bench.log

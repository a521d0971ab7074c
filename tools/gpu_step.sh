#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/full_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/full_gpu_tests.log
cat gpurun_out/full_gpu_tests.log

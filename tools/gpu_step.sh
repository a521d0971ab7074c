#!/bin/bash
# scratch driver for one gpurun call (rewritten per run)
cd /root/repo
mkdir -p gpurun_out
timeout 900 python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -c 300 gpurun_out/r02_bench.json

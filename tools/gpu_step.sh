cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1700 python -m pytest tests/test_gpu_dp2.py tests/test_gpu_backward.py tests/test_gpu_rccl.py tests/test_gpu_extras.py -x -q -s 2>&1 | grep -E "^dp2 |passed|failed|Error|error" | tail -20) > gpurun_out/r2_s16_test.log 2>&1
cat gpurun_out/r2_s16_test.log | cut -c1-400

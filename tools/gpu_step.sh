cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1700 python -m pytest tests/test_gpu_dp2.py tests/test_gpu_backward.py -x -q -s -k "two_rank or full_training_step_golden or smooth_golden" 2>&1 | grep -E "dp2|passed|failed|Error|error|assert" | tail -20) > gpurun_out/r2_s10_test.log 2>&1
cat gpurun_out/r2_s10_test.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_bf16.py -x -q -s -k "reference_init or smooth_bf16" 2>&1 | grep -E "BF16GRAD|drift per step|passed|failed|Error" | tail -25) > gpurun_out/r2_s6_test.log 2>&1
cat gpurun_out/r2_s6_test.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_bf16.py -k "qkv_fused" -x -q 2>&1 | tail -3) > gpurun_out/r2_s12.log 2>&1
(timeout 300 python tools/bench_kernels.py attn_qkv_bf16 2>&1 | tail -4) >> gpurun_out/r2_s12.log 2>&1
cat gpurun_out/r2_s12.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_bf16.py -k "qkv_fused" -x -q 2>&1 | tail -3) > gpurun_out/r2_s17.log 2>&1
for m in 21 12 22 21 12; do
echo "mode $m" >> gpurun_out/r2_s17.log
(PANGU_ATTN_QKV_MODE=$m timeout 300 python tools/bench_kernels.py attn_qkv_bf16 2>&1 | grep attn_qkv | cut -c1-60) >> gpurun_out/r2_s17.log 2>&1
done
cat gpurun_out/r2_s17.log

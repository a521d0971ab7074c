cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(PANGU_MLP_NW192=8 timeout 900 python -m pytest tests/test_gpu_bf16.py -k "mlp_ln_residual_fused" -x -q 2>&1 | tail -3) > gpurun_out/r2_s14.log 2>&1
for i in 1 2; do
(PANGU_MLP_NW192=4 timeout 300 python tools/bench_kernels.py mlp_fused 2>&1 | grep "s0 mlp") >> gpurun_out/r2_s14.log 2>&1
(PANGU_MLP_NW192=8 timeout 300 python tools/bench_kernels.py mlp_fused 2>&1 | grep "s0 mlp") >> gpurun_out/r2_s14.log 2>&1
done
cat gpurun_out/r2_s14.log

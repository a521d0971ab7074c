cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_bf16.py -k "mlp_ln_residual_fused or qkv_fused or full_model_bf16" -x -q 2>&1 | tail -5) > gpurun_out/r2_s9_test.log 2>&1
(timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['bf16_forward']))") > gpurun_out/r2_s9_fwd.log 2>&1
cat gpurun_out/r2_s9_test.log gpurun_out/r2_s9_fwd.log

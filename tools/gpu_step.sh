cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_bf16.py -k "qkv_fused or mlp_ln_residual_fused" -x -q 2>&1 | tail -5) > gpurun_out/r2_s5_test.log 2>&1
(timeout 300 python tools/bench_kernels.py attn_qkv_bf16 2>&1 | tail -4) > gpurun_out/r2_s5_bench.log 2>&1
(PANGU_ATTN_QKV_RING=3 timeout 300 python tools/bench_kernels.py attn_qkv_bf16 2>&1 | tail -4) >> gpurun_out/r2_s5_bench.log 2>&1
(timeout 300 python tools/bench_kernels.py mlp_fused 2>&1 | tail -2) >> gpurun_out/r2_s5_bench.log 2>&1
(timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['bf16_forward']))") > gpurun_out/r2_s5_fwd.log 2>&1
cat gpurun_out/r2_s5_test.log gpurun_out/r2_s5_bench.log gpurun_out/r2_s5_fwd.log

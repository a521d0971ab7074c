cd $GRAFT_REPO_ROOT
export PANGU_COMMIT=dad9924
mkdir -p gpurun_out
(timeout 900 python bench.py 2>&1 | tail -1) > gpurun_out/r02_bench.json
bash tools/pmc_traffic.sh r02 > gpurun_out/r2_pmc_traffic.log 2>&1
(echo "# Round 2 -- per-kernel micro-benchmarks on one MI355X (tools/bench_kernels.py, model shapes), commit dad9924"; echo '```'; for k in mlp_fused attn_qkv_bf16 attn_bf16 gemm_bf16 gemm_ln_bf16 attn gemm; do echo "## $k"; timeout 300 python tools/bench_kernels.py $k 2>&1 | grep -v amdgpu.ids; done; echo '```') > gpurun_out/r02_kernel_microbench.md
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_train_trace -- python3 bench.py --steps 2 --warmup 1 --no-bf16 --no-cpu-baseline --train-steps 3 > gpurun_out/r02_train_trace.log 2>&1
cp $(ls gpurun_out/r02_train_trace/*/*kernel_stats.csv | head -1) gpurun_out/r02_train_f32_bf16_kernel_stats.csv
cut -c1-300 gpurun_out/r02_bench.json; tail -30 gpurun_out/r02_kernel_microbench.md

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r2_s15.log
for v in 0 1 0 1; do
(PANGU_BF16_FUSE_LN384=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-train 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LN384=$v', d['bf16_forward']['ms_per_step'], d['ms_per_step'])") >> gpurun_out/r2_s15.log 2>&1
done
cat gpurun_out/r2_s15.log

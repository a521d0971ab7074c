cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/ablate_mlp.py 2>&1 | tail -30) > gpurun_out/r2_s2_ablate.log 2>&1
cat gpurun_out/r2_s2_ablate.log

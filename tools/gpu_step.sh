cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 300 python tools/ablate_attn.py 2>&1 | grep "C=") > gpurun_out/r2_s18_attn_stamp.log 2>&1
cat gpurun_out/r2_s18_attn_stamp.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r2_s13_suite.log 2>&1
cat gpurun_out/r2_s13_suite.log

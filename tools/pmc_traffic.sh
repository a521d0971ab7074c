#!/bin/bash
# One command that regenerates profiles/pmc_traffic_f32.json (and the bf16 twin) on the GPU box:
#   bash tools/pmc_traffic.sh            (run through gpurun; every rocprofv3 pass under `timeout`)
# 1. kernel trace of the profiled command -> the kernel names the forward launches (profiles/<tag>_kernel_stats.csv)
# 2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_INSTS_VALU SQ_WAVES), as MI355X_MICROARCH.md prescribes
# 3. tools/pmc_to_json.py: per-launch HBM bytes (FETCH_SIZE x2), MFMA busy, the git commit, sha256 of the kernel sources, and a
#    check that every kernel family priced from the counters was launched by the traced command (fails otherwise)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:-r02}
for DT in f32 bf16; do
  O=gpurun_out/pmc_${TAG}_${DT}
  rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/profile_fwd.py $DT 3 > $O/trace.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/profile_fwd.py $DT 3 > $O/fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/profile_fwd.py $DT 3 > $O/write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 tools/profile_fwd.py $DT 3 > $O/mfma.log 2>&1
  # vector-ALU instructions issued (wave-level, MFMAs included) and waves launched: the VALU-issue roof of the bf16 attention kernel
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/valu -- python3 tools/profile_fwd.py $DT 3 > $O/valu.log 2>&1
  python3 tools/pmc_to_json.py $O $DT > $O/pmc_traffic_${DT}.json || { echo "pmc_to_json failed for $DT"; tail -3 $O/*.log; }
  cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${TAG}_fwd_${DT}_kernel_stats.csv 2>/dev/null
  head -c 1500 $O/pmc_traffic_${DT}.json
done

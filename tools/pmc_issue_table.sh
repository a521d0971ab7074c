#!/bin/bash
# Counter-backed account of where the kernels' cycles go (the fused bf16 forward kernels; VERDICT r3 item 1: the
# bf16 TRAINING step's kernels):
#   bash tools/pmc_issue_table.sh            > profiles/<tag>_fwd_bf16_issue_table.md      (through gpurun; each pass under `timeout`)
#   bash tools/pmc_issue_table.sh train bf16 > profiles/<tag>_train_bf16_issue_table.md    (tools/profile_train.py bf16 2 1)
# Separate rocprofv3 --pmc passes of the profiled command (kernel-trace for the durations of the SAME pass):
#   pass 1: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY   (disjoint: parked | issue-stalled | issuing)
#   pass 2: SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS
#   pass 3: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE                          (MFMA busy, effective clock)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=/tmp/pmc_issue; rm -rf $O; mkdir -p $O
if [ "${1:-fwd}" = train ]; then CMD="tools/profile_train.py ${2:-bf16} 2 1"; else CMD="tools/profile_fwd.py bf16 3"; fi
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -- python3 $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $CMD > $O/p3.log 2>&1
python3 tools/pmc_issue_table.py $O "$CMD"

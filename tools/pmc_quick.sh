#!/bin/bash
# Issue / MFMA / LDS / L1 counters of the kernels matching a name filter, for any probe command:
#   bash tools/pmc_quick.sh "<python3 args..>" "<kernel name substring[,substring]>"  > table.md
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
P="python3 $1"; F=${2:-kernel}
O=/tmp/pmc_quick; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p3 -- $P > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/p4 -- $P > $O/p4.log 2>&1
echo "Counters per dispatch (mean; rocprofv3 --pmc passes of \`$P\`), kernels matching \"$F\":"
echo
python3 tools/pmc_counter_table.py $O "$F"

#!/bin/bash
# Counters of the fused QKV attention, (window, head) kernel vs the longitude-walking kernel ("settle it with
# counters": the L1 -> L2 read requests per launch before / after, next to the time):  bash experiments/tools/pmc_walk_ab.sh [variants] > profiles/<tag>.md
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
V=${1:-0,40,21}
O=/tmp/pmc_walk; rm -rf $O; mkdir -p $O
P="python3 experiments/tools/pmc_walk_probe.py $V 3"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p3 -- $P > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/p4 -- $P > $O/p4.log 2>&1
echo "# fused QKV attention (bf16): (window, head) kernel vs longitude-walking kernel, variants $V"
echo
echo "Counters per dispatch (mean over 3 dispatches per instantiation x shift; rocprofv3 --pmc passes of \`$P\`); TCP_TCC_READ_REQ = L1 -> L2 read requests (64 B each);"
echo "FETCH_SIZE in KiB as reported (x2 for bytes on gfx950).  Instantiations: <SHIFTED, C[, pipelines, bias mode]>."
echo
python3 tools/pmc_counter_table.py $O
for p in p1 p2 p3 p4; do grep -il "error\|invalid\|not supported" $O/$p.log >/dev/null 2>&1 && { echo; echo "($p log tail)"; tail -3 $O/$p.log; }; done

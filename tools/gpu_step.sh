cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_mlp2; mkdir -p gpurun_out/pmc_mlp2
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_mlp2
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -- python3 tools/pmc_mlp_probe.py > $O/p1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 tools/pmc_mlp_probe.py > $O/p2.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/p3 -- python3 tools/pmc_mlp_probe.py > $O/p3.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4 -- python3 tools/pmc_mlp_probe.py > $O/p4.log 2>&1
python3 tools/pmc_summary.py $O $O/summary.md > /dev/null 2>&1
ls $O

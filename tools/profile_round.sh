#!/bin/bash
# Regenerates the round's committed measurements on the GPU box (run through gpurun; every rocprofv3 pass under `timeout`):
#   bash tools/profile_round.sh r02
# -> gpurun_out/<tag>_*: forward kernel stats + PMC traffic JSONs (tools/pmc_traffic.sh), training-step kernel stats per dtype,
#    the training PMC table (FETCH_SIZE | WRITE_SIZE | MFMA busy, three separate passes), the kernel micro-benchmarks and the
#    default bench line.  Copy what is to be judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:-r02}
mkdir -p gpurun_out
bash tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.log 2>&1
# bench.py reads roofline.traffic from profiles/pmc_traffic_<dtype>.json (and checks the kernel sources' hashes): let the
# bench line at the end of this script see the tables just measured
cp gpurun_out/pmc_${TAG}_f32/pmc_traffic_f32.json gpurun_out/pmc_${TAG}_bf16/pmc_traffic_bf16.json profiles/ 2>/dev/null
for DT in f32 bf16; do
  rm -rf /tmp/tr_$DT
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$DT -- python3 tools/profile_train.py $DT 3 1 > gpurun_out/${TAG}_train_${DT}.log 2>&1
  cp "$(find /tmp/tr_$DT -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_train_${DT}_kernel_stats.csv
done
# per dtype (the two steps share kernel names -- Adam, row kernels, the reduce launch -- so each gets its own passes): per-kernel
# table + whole-step byte totals (profiles/pmc_train.json: bench.py's training roofline blocks)
for DT in f32 bf16; do
  P=/tmp/trpmc_$DT; rm -rf $P; mkdir -p $P
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/p1 -- python3 tools/profile_train.py $DT 2 1 > $P/p1.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/p2 -- python3 tools/profile_train.py $DT 2 1 > $P/p2.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $P/p3 -- python3 tools/profile_train.py $DT 2 1 > $P/p3.log 2>&1
  { echo "## $DT training step"; python3 tools/pmc_train_table.py $P; } >> gpurun_out/${TAG}_train_pmc_table.md 2>> gpurun_out/${TAG}_train_pmc_table.err
  # the same command with ZERO measured steps (model construction + init + the warm-up step): subtracted from the byte totals
  mkdir -p ${P}_w0
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${P}_w0/p1 -- python3 tools/profile_train.py $DT 0 1 > ${P}_w0/p1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${P}_w0/p2 -- python3 tools/profile_train.py $DT 0 1 > ${P}_w0/p2.log 2>&1
done
python3 tools/pmc_train_table.py --json gpurun_out/pmc_train.json /tmp/trpmc_f32 /tmp/trpmc_bf16 2 && cp gpurun_out/pmc_train.json profiles/
{
  for sec in mlp_fused mlp_train attn_qkv_bf16 attn_bf16 gemm_bf16 wgrad_bf16 gemm_ln_bf16 attn attn_bwd gemm wgrad; do
    echo "## $sec"
    timeout 400 python3 tools/bench_kernels.py $sec --lib-compare 2>&1 | grep -v "amdgpu.ids"
  done
} > gpurun_out/${TAG}_kernel_microbench.txt
# training-step A/B of the one remaining training knob (one process per arm, interleaved, two rounds)
{
  for r in 1 2; do
    for m in 0 1; do echo "PANGU_BF16_TRAIN_MLP=$m: $(PANGU_BF16_TRAIN_MLP=$m timeout 300 python3 tools/profile_train.py bf16 6 2 2>&1 | grep train)"; done
  done
} > gpurun_out/${TAG}_train_ab.txt 2>&1
timeout 900 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json

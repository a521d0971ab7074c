#!/bin/bash
# Everything the round commits under profiles/, regenerated on the GPU box in ONE gpurun call (every rocprofv3 pass under `timeout`,
# counters in their own passes):   bash tools/profile_all.sh r04
#   tools/profile_round.sh  -> forward kernel stats + PMC traffic JSONs, training kernel stats, training PMC table + pmc_train.json,
#                              kernel micro-benchmarks, training A/B, the default bench line
#   tools/pmc_per_shape.py  -> per-shape clock / MFMA-busy of the forward kernels
#   tools/pmc_issue_table.sh (forward bf16, training bf16, training f32) -> issuing / stalled / parked, LDS busy + conflicts, MFMA busy
#   tools/per_step_kernels.sh (bf16, f32) -> exact per-step kernel lists of the training step
#   1-rank RCCL run of the bench's training section (PANGU_DIST_FORCE=1)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:-r04}
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1
for DT in f32 bf16; do
  python3 tools/pmc_per_shape.py gpurun_out/pmc_${TAG}_${DT}/mfma > gpurun_out/${TAG}_fwd_${DT}_per_shape.md 2>> gpurun_out/${TAG}_round.log
done
bash tools/pmc_issue_table.sh > gpurun_out/${TAG}_fwd_bf16_issue_table.md 2>> gpurun_out/${TAG}_round.log
bash tools/pmc_issue_table.sh train bf16 > gpurun_out/${TAG}_train_bf16_issue_table.md 2>> gpurun_out/${TAG}_round.log
bash tools/pmc_issue_table.sh train f32 > gpurun_out/${TAG}_train_f32_issue_table.md 2>> gpurun_out/${TAG}_round.log
bash tools/per_step_kernels.sh bf16 > gpurun_out/${TAG}_train_bf16_per_step.md 2>> gpurun_out/${TAG}_round.log
bash tools/per_step_kernels.sh f32 > gpurun_out/${TAG}_train_f32_per_step.md 2>> gpurun_out/${TAG}_round.log
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 PANGU_DIST_FORCE=1 timeout 600 python3 bench.py --no-bf16 --cpu-baseline none --steps 3 --warmup 1 > gpurun_out/${TAG}_rccl_1rank.json 2> gpurun_out/${TAG}_rccl_1rank.err
ls -la gpurun_out/${TAG}_* | head -40

#!/bin/bash
# Timing-only ablation builds of csrc/mlp_fused_f32.hip (whole library re-linked with the one object replaced) and their micro-benchmark:
#   bash tools/ablate_mlp_f32.sh build      (in the build container: scratch/libpangu_mlp_<tag>.so)
#   bash tools/ablate_mlp_f32.sh run        (on the GPU box)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
C=pangu-pytorch_amd/csrc
TAGS="base:-DMLP_PIN=1 nopin:-DMLP_PIN=0 nogelu:-DMLP_NOGELU=1 nodma:-DMLP_NODMA=1 dual:-DMLP_DUAL=1 nogelu_nodma:-DMLP_NOGELU=1,-DMLP_NODMA=1 dual_nopin:-DMLP_DUAL=1,-DMLP_PIN=0"
if [ "${1:-run}" = build ]; then
  mkdir -p scratch
  OBJS=$(ls $C/*.o | grep -v mlp_fused_f32.o)
  for t in $TAGS; do
    tag=${t%%:*}; fl=$(echo ${t#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $fl -c $C/mlp_fused_f32.hip -o scratch/mlp_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libpangu_mlp_$tag.so $OBJS scratch/mlp_$tag.o && rm scratch/mlp_$tag.o
  done
  ls -la scratch/libpangu_mlp_*.so
else
  for rnd in 1 2; do for t in $TAGS; do
    tag=${t%%:*}
    echo "== $tag"; PANGU_HIP_LIB=$PWD/scratch/libpangu_mlp_$tag.so python3 tools/bench_kernels.py mlp_f32 2>&1 | grep mlp_f32 | tail -1
  done; done
fi

#!/bin/bash
# Timing-only ablation builds of experiments/csrc/mlp_fused_f32.hip (the experiments library re-linked with that one object rebuilt
# under extra flags) and their micro-benchmark:
#   bash experiments/tools/ablate_mlp_f32.sh build      (in the build container: scratch/libexp_mlp_<tag>.so)
#   bash experiments/tools/ablate_mlp_f32.sh run        (on the GPU box)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
C=pangu-pytorch_amd/csrc
E=experiments/csrc
TAGS="base:-DMLP_PIN=1 nopin:-DMLP_PIN=0 nogelu:-DMLP_NOGELU=1 nodma:-DMLP_NODMA=1 dual:-DMLP_DUAL=1 nogelu_nodma:-DMLP_NOGELU=1,-DMLP_NODMA=1 dual_nopin:-DMLP_DUAL=1,-DMLP_PIN=0"
if [ "${1:-run}" = build ]; then
  mkdir -p scratch
  make -C experiments > /dev/null
  for t in $TAGS; do
    tag=${t%%:*}; fl=$(echo ${t#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$C $fl -c $E/mlp_fused_f32.hip -o scratch/mlp_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libexp_mlp_$tag.so $E/attn_walk_bf16.o scratch/mlp_$tag.o $C/capi.hip -I$C && rm scratch/mlp_$tag.o
  done
  ls -la scratch/libexp_mlp_*.so
else
  for rnd in 1 2; do for t in $TAGS; do
    tag=${t%%:*}
    echo "== $tag"; PANGU_EXP_LIB=$PWD/scratch/libexp_mlp_$tag.so python3 experiments/tools/bench_mlp_f32.py 2>&1 | grep mlp_f32 | tail -1
  done; done
fi

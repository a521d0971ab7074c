#!/bin/bash
# Round-6 follow-up on the GPU box: ring depth of the persistent projection + LayerNorm kernel, and the issue tables of both A/Bs.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
S=$PWD/scratch
echo "## parity: default build (ring of 4 at N = 384) and the ring-of-3 build"
timeout 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "linear_ln_residual" 2>&1 | tail -2
PANGU_HIP_LIB=$S/libpangu_lnr3.so timeout 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "linear_ln_residual" 2>&1 | tail -2
echo "## projection + LayerNorm + residual, bf16: ring4 (default) | ring3 | register-staged"
for r in 1 2 3; do for t in ring4 lnr3 lnreg; do L=""; [ $t != ring4 ] && L=$S/libpangu_$t.so
  echo "== $t"; PANGU_HIP_LIB=$L timeout 200 python tools/bench_kernels.py gemm_ln_bf16 2>&1 | grep "LN"; done; done
for t in base occ4; do L=""; [ $t = occ4 ] && L=$S/libpangu_occ4.so
  echo "## issue table, fused QKV + attention: $t"; PANGU_HIP_LIB=$L bash tools/pmc_quick.sh "tools/pmc_fused_probe.py attn_qkv 3" "window_attn_qkv"; done
for t in ring4 lnreg; do L=""; [ $t != ring4 ] && L=$S/libpangu_$t.so
  echo "## issue table, projection + LayerNorm: $t"; PANGU_HIP_LIB=$L bash tools/pmc_quick.sh "tools/pmc_fused_probe.py gemm_ln 3" "gemm_ln_residual"; done

#!/bin/bash
# Round-6 kernel A/Bs on the GPU box (interleaved, one process per arm; libraries built by tools/ab_lib.sh in the build container):
#   occ4   attn_bf16.hip  -DPANGU_ATTN_QKV_MIN_WAVES=4  (128 VGPRs, key halves + online softmax, one live bias row: 5 workgroups / CU)
#   lnreg  gemm_ln_bf16.hip -DPANGU_GEMM_LN_DMA=0       (the register-staged projection + LayerNorm kernel of rounds 2-5)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
S=$PWD/scratch
echo "## parity of the variant builds"
PANGU_HIP_LIB=$S/libpangu_occ4.so timeout 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "qkv_fused" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -k "linear_ln_residual" 2>&1 | tail -2
echo "## fused QKV + attention, bf16: base vs occ4"
for r in 1 2 3; do for t in base occ4; do L=""; [ $t = occ4 ] && L=$S/libpangu_occ4.so
  echo "== $t"; PANGU_HIP_LIB=$L timeout 200 python tools/bench_kernels.py attn_qkv_bf16 2>&1 | grep attn_qkv; done; done
echo "## projection + LayerNorm + residual, bf16: persistent LDS-DMA kernel (default build) vs register-staged (lnreg)"
for r in 1 2 3; do for t in dma lnreg; do L=""; [ $t = lnreg ] && L=$S/libpangu_lnreg.so
  echo "== $t"; PANGU_HIP_LIB=$L timeout 200 python tools/bench_kernels.py gemm_ln_bf16 2>&1 | grep "LN"; done; done

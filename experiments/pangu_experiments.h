/* Entry points of the kernels that were built, are parity-green and LOST their A/B (profiles/r05_walk_attn_ab.md,
 * profiles/r05_mlp_f32_ab.md).  They are NOT part of libpangu_hip.so / include/pangu_hip.h (round 6: pruned to the product);
 * `make -C experiments` builds experiments/libpangu_experiments.so, `python -m pytest experiments -m experiments` runs their
 * parity tests on a GPU box.  Conventions as in include/pangu_hip.h. */
#ifndef PANGU_EXPERIMENTS_H
#define PANGU_EXPERIMENTS_H
#include "../include/pangu_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The whole MLP branch of a block in ONE launch, fp32 inference (reference layers.py:251 with :264-270 inside):
 *   out[M,C] = x[M,C] + branch_scale * (LayerNorm(GELU(x @ W1^T + b1) @ W2^T + b2) * gamma + beta)
 * W1 [4C][C], b1 [4C], W2 [C][4C], b2 [C] in the reference's torch layouts (read in place: nothing is packed); x / out row strides
 * ldx / ldo (multiples of 4); C = 192 (the stage-0 / stage-3 width).  The (M x 4C) hidden activation stays on chip.  Replaces pangu_linear_fwd
 * (PANGU_ACT_GELU) + pangu_linear_ln_residual_fwd. */
int pangu_mlp_ln_residual_fwd(pangu_stream_t stream, const float* x, int ldx, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* gamma, const float* beta, float* out, int ldo, int M, int C,
                              float branch_scale);

/* The same operator (same arguments, same result) in its LONGITUDE-WALKING form: one persistent workgroup per (window type,
 * head) keeps that head's 96 linear1 rows in LDS (and, variant % 10 == 1, the wave's Earth-specific bias rows in registers) and
 * walks the nLon longitude windows that share them (reference layers.py:306-311: one bias per (type, head), broadcast over
 * longitude, :395).  variant = 10 * pipelines + bias_mode: 40, 30, 20 (bias rows re-read from L2 per window), 21, 11 (resident). */
int pangu_window_attn_qkv_walk_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_qkv, const float* b_qkv,
                                        const void* esb, void* out, float* lse, int Z, int H, int W, int C, int heads,
                                        int shifted, int variant);

#ifdef __cplusplus
}
#endif
#endif

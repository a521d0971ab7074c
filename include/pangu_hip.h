/*
 * pangu_hip.h — C ABI of the MI355X (gfx950) kernels behind the Pangu-Weather hot path.
 *
 * Drop-in boundary.  The reference (zhaoshan2/pangu-pytorch) has no native layer: its "operator API" for
 * this path is the nn.Module surface of models/pangu_model.py:8-87 and the layer classes of
 * models/layers.py.  This library sits UNDER that surface: each entry point replaces the ATen-op chain of
 * one reference method (cited per function) and is what a ctypes/cffi binding on the reference side binds
 * (see INTEGRATION.md).  Plain pointers and sizes only — no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; tensors are dense row-major fp32 (dtype-tagged
 *     variants take PANGU_F32 / PANGU_BF16 where noted);
 *   - `stream` is a hipStream_t passed as void*; nothing allocates, synchronises or branches on device
 *     data on the host => every call is hipGraph-capturable;
 *   - return value: PANGU_OK (0), a negative PANGU_E_* for bad arguments (nothing launched), or a positive
 *     hipError_t if the launch itself failed;
 *   - token tensors are (N, C) with N = Z*H*W tokens in (z,h,w) order (reference layers.py:188,247),
 *     row stride given explicitly as `ld*` (in elements) where a kernel supports strided rows.
 */
#ifndef PANGU_HIP_H
#define PANGU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANGU_ABI_VERSION 1

#define PANGU_OK 0
#define PANGU_E_SHAPE (-1)     /* unsupported shape / divisibility */
#define PANGU_E_NULL (-2)      /* required pointer is NULL */
#define PANGU_E_DTYPE (-3)     /* unsupported dtype tag */
#define PANGU_E_ARG (-4)       /* other invalid argument */
#define PANGU_E_RANGE (-5)     /* a row-strided matrix spans 4 GB or more: the linear / wgrad entries use 32-bit byte
                                  offsets (range-checked buffer addressing); split the call by rows */

#define PANGU_F32 0
#define PANGU_BF16 1

#define PANGU_ACT_NONE 0
#define PANGU_ACT_GELU 1       /* exact erf GELU, reference layers.py:261; aux (optional) receives the pre-activation */
#define PANGU_ACT_GELU_BWD 2   /* C = (A @ W^T) * gelu'(aux): backward through the GELU, aux = saved pre-activation */
#define PANGU_ACT_ADD 3        /* C = A @ W^T + bias + aux: residual-gradient accumulation fused into the data-gradient GEMM */
#define PANGU_ACT_GELU_BWD_H 4 /* bf16 only (pangu_linear_gelu_bwd_bf16): PANGU_ACT_GELU_BWD, and h = GELU(aux) is written too */

typedef void* pangu_stream_t;  /* hipStream_t */

int pangu_abi_version(void);
const char* pangu_error_string(int code);

/* ---- integer contract (bit-exact vs oracle) ------------------------------------------------------- */

/* out[nLon][types][144] int32: source token of every window slot, -1 = zero pad.
 * Replaces view/pad/roll/partition of EarthSpecificBlock.forward, reference layers.py:188-221. */
int pangu_window_index_export(pangu_stream_t stream, int32_t* out, int Z, int H, int W, int shifted);

/* out[types][144][144] fp32 in {0,-100}: shifted-window mask, reference layers.py:153-181 (gen_mask). */
int pangu_window_mask_export(pangu_stream_t stream, float* out, int Z, int H, int W);

/* ---- dense projections ---------------------------------------------------------------------------- */

/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]).  W is the torch nn.Linear / Conv1d(k=1) weight as stored
 * (out,in).  bias may be NULL.  K % 16 == 0, N % 4 == 0.  aux [M][N] (dense): see PANGU_ACT_*; NULL otherwise.
 * Replaces nn.Linear / nn.Conv1d calls at reference layers.py:68,86,265-268,365,418,457,476,498,520,536.
 * The input-gradient of a projection is the same call with W^T: dA[M,K] = dC[M,N] @ (W^T)[K,N]^T. */
int pangu_linear_fwd(pangu_stream_t stream, const float* A, int lda, const float* W, const float* bias,
                     float* C, int ldc, int M, int N, int K, int act, float* aux);

/* Weight/bias gradient of a projection (autograd of the calls above; the training step of reference
 * models/pangu_sample.py:71 `loss.backward()`):
 *   dW[N,K] += dC[M,N]^T @ A[M,K]      db[N] += sum_m dC[m,:]   (db may be NULL)
 * ACCUMULATES with fp32 atomics into caller-initialised buffers (zero them, or pass live .grad buffers). */
int pangu_linear_wgrad(pangu_stream_t stream, const float* dC, int lddc, const float* A, int lda, float* dW,
                       float* db, int M, int N, int K);
/* The same product with a caller-owned scratch buffer (see pangu_linear_wgrad_bf16_ws): partial tiles of the token slabs
 * through the buffer + one reduce launch instead of fp32 atomics; NULL / too small = the entry above. */
int pangu_linear_wgrad_ws(pangu_stream_t stream, const float* dC, int lddc, const float* A, int lda, float* dW,
                          float* db, int M, int N, int K, float* workspace, long long workspace_bytes);

/* ---- Earth-specific window attention -------------------------------------------------------------- */

/* Fused roll + window partition + (q*scale)k^T + earth_specific_bias + shift mask + softmax + .v +
 * window reverse + roll back + crop, one (window, head) per workgroup.
 *   qkv  [N][3C]  linear1 output on the UNPADDED tokens (channel = which*C + head*32 + d, layers.py:368-371)
 *   qkv_bias [3C] linear1.bias: the q/k/v of zero-pad tokens (layers.py:192 pads before linear1)
 *   esb  [types][heads][144][144] earth_specific_bias (layers.py:306-311)
 *   out  [N][C]   attention output before linear2 (channel = head*32 + d, layers.py:413-415)
 *   lse  [N][heads] optional (may be NULL): log-sum-exp of each query row, saved for backward
 * Replaces reference layers.py:192-247 + :368-415. */
int pangu_window_attn_fwd(pangu_stream_t stream, const float* qkv, const float* qkv_bias, const float* esb,
                          float* out, float* lse, int Z, int H, int W, int C, int heads, int shifted);
/* Inference mode on the paper's COMPACT Earth-specific bias: esb_compact is [types][heads][3312] fp32 (the (3312, types, heads)
 * table of reference layers.py:306-357 with the index axis last); the kernel gathers each score's bias through the
 * position index in closed form (layers.py:319-357, :384-391) -- 13 KB per (type, head) instead of 83 KB, 10 MB instead
 * of 62 MB per block.  Bit-identical to pangu_window_attn_fwd on the expanded table of the same values. */
int pangu_window_attn_fwd_compact(pangu_stream_t stream, const float* qkv, const float* qkv_bias,
                                  const float* esb_compact, float* out, float* lse, int Z, int H, int W, int C,
                                  int heads, int shifted);

/* EarthAttention3D.forward's own calling convention (reference layers.py:360-421, the part between linear1 and linear2) for
 * callers that use the module OUTSIDE EarthSpecificBlock: the tensor is already partitioned and the mask is explicit.
 *   qkv  [n_lon*types*144][3C]  linear1 output of the window slots in (lon window, type, slot) order (every slot an ordinary
 *                               token: nothing is a pad here), channel = which*C + head*32 + d
 *   esb  [types][heads][144][144]
 *   mask NULL (layers.py:404-405) or fp32 [n_lon][types][144][144] with mask_lon_stride = types*144*144, or one
 *        [types][144][144] table shared by all longitude windows with mask_lon_stride = 0 (layers.py:401-402)
 *   out  [n_lon*types*144][C]   channel = head*32 + d (layers.py:413-415)
 * Not on the hot path (plain VALU kernel, one (window, head) per workgroup). */
int pangu_attn_windows_fwd(pangu_stream_t stream, const float* qkv, const float* esb, const float* mask,
                           long long mask_lon_stride, float* out, int n_lon, int types, int heads, int C);

/* Backward of pangu_attn_windows_fwd: dout [rows][C] -> dqkv [rows][3C] (every row written), d_esb [types][heads][144][144]
 * (overwritten: summed over the n_lon windows inside the kernel, no atomics).  The mask gets no gradient (layers.py:153-181 builds
 * it from constants). */
int pangu_attn_windows_bwd(pangu_stream_t stream, const float* qkv, const float* esb, const float* mask,
                           long long mask_lon_stride, const float* dout, float* dqkv, float* d_esb, int n_lon, int types,
                           int heads, int C);

/* Backward of pangu_window_attn_fwd.  One workgroup per (window type, head) walks the nLon longitude windows
 * and keeps the bias gradient d_esb[t][head] = sum_l dS in registers (no atomics, written once).
 *   out, lse: the forward's outputs;  dout [N][C]: gradient w.r.t. out
 *   dqkv [N][3C]: gradient w.r.t. qkv (every real token is written exactly once per q/k/v)
 *   dqkv_bias [3C]: ACCUMULATED (atomics) gradient reaching linear1.bias through the zero-pad slots
 *   d_esb [types][heads][144][144]: overwritten */
int pangu_window_attn_bwd(pangu_stream_t stream, const float* qkv, const float* qkv_bias, const float* esb,
                          const float* out, const float* lse, const float* dout, float* dqkv, float* dqkv_bias,
                          float* d_esb, int Z, int H, int W, int C, int heads, int shifted);

/* ---- row kernels ----------------------------------------------------------------------------------- */

/* out[r] = shortcut[r] + branch_scale * (LayerNorm(y[r]) * gamma + beta)
 * (post-norm residual, reference layers.py:250-251; branch_scale = 1 in eval, the DropPath keep factor
 * 1/(1-p) in training, timm DropPath semantics).  eps = 1e-5.  shortcut/out may have row strides != C
 * (lds/ldo).  mean_rstd (may be NULL): [N][2] saved statistics for backward. */
int pangu_ln_residual_fwd(pangu_stream_t stream, const float* y, const float* shortcut, int lds,
                          const float* gamma, const float* beta, float* out, int ldo, float* mean_rstd,
                          int N, int C, float branch_scale);

/* Projection + post-norm residual in one launch (inference path of layers.py:250-251), fp32, N = 192 only (the GEMM tile
 * spans the whole row):  out[M,N] = shortcut[M,N] + branch_scale * (LayerNorm(A[M,K] @ W[N,K]^T + bias) * gamma + beta).
 * shortcut / out row strides lds / ldo; bias may be NULL; K % 16 == 0.  Replaces pangu_linear_fwd + pangu_ln_residual_fwd. */
int pangu_linear_ln_residual_fwd(pangu_stream_t stream, const float* A, int lda, const float* W, const float* bias,
                                 const float* shortcut, int lds, const float* gamma, const float* beta, float* out, int ldo,
                                 int M, int N, int K, float branch_scale);

/* Backward of the LayerNorm branch of pangu_ln_residual_fwd (the shortcut's gradient is dout itself):
 *   dy [N][C] overwritten;  dgamma[C], dbeta[C] ACCUMULATED (atomics).  dout may be row-strided (lddo). */
int pangu_ln_residual_bwd(pangu_stream_t stream, const float* dout, int lddo, const float* y, const float* gamma,
                          float* dy, float* dgamma, float* dbeta, int N, int C, float branch_scale);

/* DownSample gather + LayerNorm(4C): x[Z][H][W][C] (row stride ldx) -> out[Z*(H+1)/2*(W/2)][4C],
 * channel = dh*2C + dw*C + c, zero row for h == H (pad).  Reference layers.py:436-454. */
int pangu_downsample_ln_fwd(pangu_stream_t stream, const float* x, int ldx, const float* gamma,
                            const float* beta, float* out, float* mean_rstd, int Z, int H, int W, int C);

/* Backward: dout [rows][4C] -> dx [Z*H*W][C] (overwritten, every token once); dgamma/dbeta [4C] ACCUMULATED.
 * dx_add (may be NULL): a second gradient of the same tokens, dense [Z*H*W][C] -- the skip connection's (reference
 * pangu_model.py:62,81: layer 0's output feeds both the down-sampling and the channel concat) -- added to dx in this pass
 * instead of by a separate elementwise kernel. */
int pangu_downsample_ln_bwd(pangu_stream_t stream, const float* dout, const float* x, int ldx, const float* gamma,
                            float* dx, float* dgamma, float* dbeta, int Z, int H, int W, int C, const float* dx_add);

/* UpSample pixel-shuffle + crop + LayerNorm(Co): y[Z][H2][W2][4*Co] -> out[Z][H][2*W2][Co] with
 * out[z][2h+dh][2w+dw][c] = y[z][h][w][dh*2Co + dw*Co + c], rows h >= H dropped.  Reference layers.py:480-495.
 * pre (may be NULL): the shuffled rows before LayerNorm, saved for backward. */
int pangu_upsample_ln_fwd(pangu_stream_t stream, const float* y, const float* gamma, const float* beta,
                          float* out, float* mean_rstd, int Z, int H2, int W2, int H, int Co);

/* Backward: dout [Z*H*2W2][Co] -> dy [Z*H2*W2][4Co] (overwritten; cropped rows get 0); dgamma/dbeta ACCUMULATED. */
int pangu_upsample_ln_bwd(pangu_stream_t stream, const float* dout, const float* y, const float* gamma, float* dy,
                          float* dgamma, float* dbeta, int Z, int H2, int W2, int H, int Co);

/* ---- patch embedding / recovery --------------------------------------------------------------------- */

/* Normalise + zero-pad + patchify the raw fields into GEMM A-matrices (reference layers.py:48-85):
 *   a_surface [H4*W4][112]  col = c*16 + ph*4 + pw,  c<4: (input_surface-mean)/std, c in 4..6: maps
 *   a_upper [7*H4*W4][192]  col = c*32 + pz*16 + ph*4 + pw, c<5: (input-mean_used)/std_used, c==5: const_h
 * with mean_used[c][l] = upper_mean[12-l][c] (level-reversed statistics, layers.py:73-76).
 * input [5][13][LAT][LON], input_surface [4][LAT][LON], maps [3][4*H4][LON], const_h [13][LAT][LON],
 * surface_mean/std [4], upper_mean/std [13][5].  LAT=721, LON=1440 -> H4=181, W4=360.
 * levels_reversed != 0: `input` is stored with its level axis in the file's (ascending) order and the reader's reversal
 * (reference era5_data/utils_data.py:117, `[::-1]` on the host) is done by this kernel's addressing: logical level l is read
 * from plane 12 - l.  Statistics, const_h and the outputs are unaffected. */
int pangu_patch_embed_gather(pangu_stream_t stream, const float* input, const float* input_surface,
                             const float* surface_mean, const float* surface_std, const float* upper_mean,
                             const float* upper_std, const float* maps, const float* const_h,
                             float* a_surface, float* a_upper, int LAT, int LON, int levels_reversed);

/* Adjoint of the gather above w.r.t. the raw fields (autograd of reference layers.py:48-55,71-76 when a caller sets
 * input.requires_grad): d_input [5][13][LAT][LON] = da_upper[token][c*32+pz*16+ph*4+pw] / upper_std[12-l][c],
 * d_input_surface [4][LAT][LON] = da_surface[token][c*16+ph*4+pw] / surface_std[c]; every element of both fields is
 * written.  da_surface [H4*W4][64], da_upper [7*H4*W4][160] fp32: the gradient of the A-matrices' field columns only (the
 * first 64 of 112 / 160 of 192; the columns of maps / const_h have no field behind them).
 * levels_reversed as in the forward: d_input is laid out like the `input` it belongs to. */
int pangu_patch_embed_gather_bwd(pangu_stream_t stream, const float* da_surface, const float* da_upper,
                                 const float* surface_std, const float* upper_std, float* d_input,
                                 float* d_input_surface, int LAT, int LON, int levels_reversed);

/* Un-patchify + crop (reference layers.py:522-543):
 *   y_upper [7*H4*W4][160] col = v*32 + pz*16 + ph*4 + pw -> output [5][13][LAT][LON]
 *   y_surface [H4*W4][64]  col = v*16 + ph*4 + pw          -> output_surface [4][LAT][LON] */
int pangu_patch_recover_scatter(pangu_stream_t stream, const float* y_upper, const float* y_surface,
                                float* output, float* output_surface, int LAT, int LON);
/* The same scatter with the reference's `normBackData` (era5_data/utils_data.py:324-330) folded in for the rollout: besides
 * the normalised fields every element is also written in physical units, phys = out * std + mean (a multiply, then an add,
 * as the reference's expression rounds) of its (variable, level) plane, into phys [5][13][LAT][LON] / phys_surface
 * [4][LAT][LON] -- the next step's input buffers.  upper_mean / upper_std: [5][13] (the `weather_statistics_last` layout,
 * utils_data.py:214-236), surface_mean / surface_std: [4]. */
int pangu_patch_recover_scatter_denorm(pangu_stream_t stream, const float* y_upper, const float* y_surface, float* output,
                                       float* output_surface, float* phys, float* phys_surface, const float* upper_mean,
                                       const float* upper_std, const float* surface_mean, const float* surface_std, int LAT,
                                       int LON);

/* Backward of the scatter: gradients of the two field tensors -> dy_upper [7*H4*W4][160], dy_surface [H4*W4][64]
 * (cropped positions get 0). */
int pangu_patch_recover_gather_bwd(pangu_stream_t stream, const float* d_output, const float* d_output_surface,
                                   float* dy_upper, float* dy_surface, int LAT, int LON);

/* ---- evaluation ---------------------------------------------------------------------------- */

/* Latitude-weighted sums behind RMSE / ACC (reference era5_data/score.py:92-105,123-135), per plane of pred/target
 * [planes][H][W]:  out[plane] = { sum w (p-t)^2, sum w p t, sum w p^2, sum w t^2 } with lat_weight w[H].
 * ACCUMULATES (atomics) into a zero-initialised out[planes][4].  W % 4 == 0. */
int pangu_lat_weighted_sums(pangu_stream_t stream, const float* pred, const float* target, const float* lat_weight,
                            float* out, int planes, int H, int W);

/* ---- bf16 variants (BASELINE configs[2], [4]) --------------------------------------------------------------
 * Activations and weight shadows are bf16 (raw uint16 bit patterns); biases, LayerNorm parameters, softmax and all
 * accumulation stay fp32.  Same semantics and layouts as the fp32 entry points above. */

/* C = act(A @ W^T + bias): A [M][K] bf16 (row stride lda), W [N][K] bf16, bias fp32 (may be NULL), C bf16 or fp32
 * (out_dtype = PANGU_BF16 / PANGU_F32), aux bf16 [M][N].  K % 8 == 0, N % 8 == 0. */
int pangu_linear_fwd_bf16(pangu_stream_t stream, const void* A, int lda, const void* W, const float* bias, void* C,
                          int ldc, int M, int N, int K, int act, void* aux, int out_dtype);

/* qkv, qkv_bias, esb, out bf16; lse fp32 (may be NULL). */
int pangu_window_attn_fwd_bf16(pangu_stream_t stream, const void* qkv, const void* qkv_bias, const void* esb, void* out,
                               float* lse, int Z, int H, int W, int C, int heads, int shifted);

/* The same attention with the QKV projection fused in (reference layers.py:365-374 + :378-415): q, k, v are computed
 * per (window, head) from the window's input rows x [tokens][C] (row stride ldx, bf16) and linear1 (w_qkv [3C][C] bf16,
 * b_qkv [3C] fp32) inside the kernel; the (tokens x 3C) qkv tensor is never materialised.  C = 192 or 384. */
int pangu_window_attn_qkv_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_qkv, const float* b_qkv,
                                   const void* esb, void* out, float* lse, int Z, int H, int W, int C, int heads,
                                   int shifted);

int pangu_ln_residual_fwd_bf16(pangu_stream_t stream, const void* y, const void* shortcut, int lds, const float* gamma,
                               const float* beta, void* out, int ldo, int N, int C, float branch_scale);
/* Projection + post-norm residual in one launch (inference path of layers.py:250-251):
 *   out[M,N] = shortcut[M,N] + LayerNorm(A[M,K] @ W[N,K]^T + bias) * gamma + beta,   N = 192 or 384 (the tile spans the row).
 * A, W, shortcut (dense, ld = N), out (row stride ldo) bf16; bias (may be NULL), gamma, beta fp32; statistics in fp32 on the
 * accumulators.  K % 8 == 0.  Replaces pangu_linear_fwd_bf16 + pangu_ln_residual_fwd_bf16 (branch scale 1). */
int pangu_linear_ln_residual_fwd_bf16(pangu_stream_t stream, const void* A, int lda, const void* W, const float* bias,
                                      const void* shortcut, const float* gamma, const float* beta, void* out, int ldo, int M,
                                      int N, int K);
/* Whole MLP branch + post-norm residual in one launch (inference path of reference models/layers.py:251 with
 * Mlp.forward :264-270 inside):
 *   out[M,C] = x[M,C] + branch_scale * ( LayerNorm( GELU(x W1^T + b1) W2^T + b2 ) * gamma + beta ),   C = 192 or 384.
 * x (row stride ldx), out (row stride ldo) bf16; b1 [4C], b2/gamma/beta [C] fp32; w_packed = the bf16 chunk image of
 * (W1 [4C][C], W2 [C][4C]) laid out as csrc/mlp_fused_bf16.hip documents (4C/32 chunks of 64*C elements).  The hidden
 * activation (M x 4C) never reaches memory.  Replaces two pangu_linear_fwd_bf16 + pangu_ln_residual_fwd_bf16. */
int pangu_mlp_ln_residual_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_packed, const float* b1,
                                   const float* b2, const float* gamma, const float* beta, void* out, int ldo, int M,
                                   int C, float branch_scale);
/* Training forward of the same branch (the bf16 counterpart of reference models/pangu_sample.py:45-77's forward through
 * layers.py:251, :264-270): as above, and the two tensors the backward needs leave from the registers they live in --
 *   pre [M][4C] bf16 (row stride ldp): x W1^T + b1 BEFORE the GELU;
 *   m   [M][C]  bf16 (row stride ldm): GELU(pre) W2^T + b2 BEFORE the LayerNorm.
 * The hidden activation h = GELU(pre) is not stored: pangu_linear_gelu_bwd_bf16 re-creates it in the backward. */
int pangu_mlp_ln_residual_train_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const void* w_packed,
                                         const float* b1, const float* b2, const float* gamma, const float* beta, void* out,
                                         int ldo, void* pre, int ldp, void* m, int ldm, int M, int C, float branch_scale);
/* Backward through Mlp.linear2 + GELU in one launch (reference layers.py:264-270 under autograd):
 *   dpre[M,N] = (A[M,K] @ W[N,K]^T) * gelu'(pre[M,N]),   h[M,N] = GELU(pre[M,N])
 * A = dm (row stride lda), W = linear2.weight^T laid out (N = 4C, K = C), pre dense [M][N]; dpre (row stride ldc) and h
 * (dense, may be NULL) bf16.  h feeds the weight gradient of linear2 (dW2 = dm^T h) and is transient. */
int pangu_linear_gelu_bwd_bf16(pangu_stream_t stream, const void* A, int lda, const void* W, void* dpre, int ldc, int M,
                               int N, int K, const void* pre, void* h);
int pangu_downsample_ln_fwd_bf16(pangu_stream_t stream, const void* x, int ldx, const float* gamma, const float* beta,
                                 void* out, int Z, int H, int W, int C);
int pangu_upsample_ln_fwd_bf16(pangu_stream_t stream, const void* y, const float* gamma, const float* beta, void* out,
                               int Z, int H2, int W2, int H, int Co);

/* fp32 fields -> bf16 GEMM operands; a_surface is [H4*W4][128] (columns 112..127 zero: K padded to a multiple of 64). */
int pangu_patch_embed_gather_bf16(pangu_stream_t stream, const float* input, const float* input_surface,
                                  const float* surface_mean, const float* surface_std, const float* upper_mean,
                                  const float* upper_std, const float* maps, const float* const_h, void* a_surface,
                                  void* a_upper, int LAT, int LON, int levels_reversed);

/* bf16 backward (configs[2]): operands / saved activations / activation gradients bf16; parameter gradients,
 * statistics and all accumulation fp32 (they feed the fp32 master weights).  Semantics as the fp32 entry points. */
int pangu_linear_wgrad_bf16(pangu_stream_t stream, const void* dC, int lddc, const void* A, int lda, float* dW,
                            float* db, int M, int N, int K);
/* The same product with a caller-owned scratch buffer (16-B aligned device memory, contents irrelevant before and after):
 * when it holds one fp32 partial tile per resident workgroup (N rounded up to whole tiles x K per token slab: <= 80 MB at the
 * model's shapes) the partial tiles are written there with plain stores (in MFMA register order) and summed into dW by a second
 * launch instead of 75 MB of fp32 atomics; too small or NULL = the entry above. */
int pangu_linear_wgrad_bf16_ws(pangu_stream_t stream, const void* dC, int lddc, const void* A, int lda, float* dW,
                               float* db, int M, int N, int K, float* workspace, long long workspace_bytes);
int pangu_window_attn_bwd_bf16(pangu_stream_t stream, const void* qkv, const void* qkv_bias, const void* esb,
                               const void* out, const float* lse, const void* dout, void* dqkv, float* dqkv_bias,
                               float* d_esb, int Z, int H, int W, int C, int heads, int shifted);
int pangu_ln_residual_bwd_bf16(pangu_stream_t stream, const void* dout, int lddo, const void* y, const float* gamma,
                               void* dy, float* dgamma, float* dbeta, int N, int C, float branch_scale);
int pangu_downsample_ln_bwd_bf16(pangu_stream_t stream, const void* dout, const void* x, int ldx, const float* gamma,
                                 void* dx, float* dgamma, float* dbeta, int Z, int H, int W, int C, const void* dx_add);
int pangu_upsample_ln_bwd_bf16(pangu_stream_t stream, const void* dout, const void* y, const float* gamma, void* dy,
                               float* dgamma, float* dbeta, int Z, int H2, int W2, int H, int Co);
int pangu_patch_recover_gather_bwd_bf16(pangu_stream_t stream, const float* d_output, const float* d_output_surface,
                                        void* dy_upper, void* dy_surface, int LAT, int LON);

/* bf16 weight shadows: every bf16 derived copy of the fp32 master weights re-made by ONE launch (the bf16 paths of
 * PanguModel compute on bf16 images of the reference's fp32 parameters -- pangu_model.py:9-48 -- which go stale at every
 * optimizer step, finetune_fully.py:121 / pangu_sample.py:75).  `jobs`: device memory, (n_jobs + 1) rows of 8 int64:
 *   [0] src0 (float*)  [1] src1 (float*, gather)  [2] dst (bf16*)  [3] idx (int32*, gather)  [4] n0  [5] n1  [6] mode
 *   [7] first block of the job; row n_jobs is a sentinel whose [7] = total_blocks.
 *   mode 0: dst[i] = bf16(src0[i]), i < n0 (4096 elements per block; src0 / dst 16-B aligned)
 *   mode 1: dst[c][r] = bf16(src0[r][c]), r < n0, c < n1 (one 64 x 64 tile per block)
 *   mode 2: e = idx[i]; dst[i] = bf16(e < n0 ? src0[e] : src1[e - n0]), i < n1 (4096 elements per block)
 * Rounding: to nearest even, as torch's float32 -> bfloat16 cast. */
int pangu_shadow_refresh_bf16(pangu_stream_t stream, const void* jobs, int n_jobs, long long total_blocks);

/* Training loss of the reference (models/pangu_sample.py:61-67, weights era5_data/config.py:45-46) in one pass per direction:
 *   loss = mean(|out - target| * w_upper[var]) + 0.25 * mean(|out_surface - target_surface| * w_surface[var])
 * out / target: [B][Vu][plane_u] fp32 (plane_u = levels * H * W), out_surface / target_surface: [B][Vs][plane_s]; w_upper [Vu],
 * w_surface [Vs] device fp32.  fwd: `partial` = scratch of pangu_weighted_l1_loss_blocks(..) floats; loss[0] = the loss,
 * loss[1] / loss[2] = the two means.  bwd: grad = device scalar (d loss); d_out = sign(out - target) * ((grad / n) * w[var])
 * in the order torch's autograd multiplies (surface: grad * 0.25 first), sign(0) = 0.
 * levels: the upper-air level count (plane_u % levels == 0): blocks never straddle a (sample, variable, level) plane.
 * The target side of the loop body is folded in (no separate pass over the 286 MB target):
 *   target_levels_reversed != 0: `target` is stored with its level axis in the file's (ascending) order (the reader's
 *     `[::-1]`, era5_data/utils_data.py:117, becomes an address: logical level l = plane levels-1-l);
 *   t_mean_upper / t_std_upper [Vu][levels] (logical level order), t_mean_surface / t_std_surface [Vs], all four or none:
 *     the targets arrive in physical units and are normalised on the fly, (t - mean) / std -- `normData`,
 *     era5_data/utils_data.py:315-321, called at models/pangu_sample.py:57.  NULL: the targets are already normalised. */
long long pangu_weighted_l1_loss_blocks(int B, int Vu, long long plane_u, int Vs, long long plane_s, int levels);
int pangu_weighted_l1_loss_fwd(pangu_stream_t stream, const float* out, const float* target, const float* out_surface,
                               const float* target_surface, const float* w_upper, const float* w_surface, float* partial,
                               float* loss, int B, int Vu, long long plane_u, int Vs, long long plane_s, int levels,
                               int target_levels_reversed, const float* t_mean_upper, const float* t_std_upper,
                               const float* t_mean_surface, const float* t_std_surface);
int pangu_weighted_l1_loss_bwd(pangu_stream_t stream, const float* out, const float* target, const float* out_surface,
                               const float* target_surface, const float* w_upper, const float* w_surface, const float* grad,
                               float* d_out, float* d_out_surface, int B, int Vu, long long plane_u, int Vs, long long plane_s,
                               int levels, int target_levels_reversed, const float* t_mean_upper, const float* t_std_upper,
                               const float* t_mean_surface, const float* t_std_surface);

/* Host side of the input pipeline (the idea of reference era5_data/utils_data.py:16-51 and the four
 * `.to(device)` of models/pangu_sample.py:41-43): copy `bytes` from the loader's pageable memory into a page-locked staging
 * buffer with up to `threads` host threads (each one contiguous 4 KB-aligned span; < 4 MB per thread is not split).  Pure
 * host code, no stream, returns when the copy is complete. */
int pangu_host_copy(void* dst, const void* src, long long bytes, int threads);

/* Adam over a whole list of tensors in ONE launch (reference finetune_fully.py:121 torch.optim.Adam, stepped at
 * models/pangu_sample.py:75): p, grad, exp_avg, exp_avg_sq fp32, updated in place with the arithmetic of torch's fused Adam
 * (L2 weight decay added to the gradient, bias corrections, eps outside the root; doubles where that code uses doubles) -- bit
 * identical to torch.optim.Adam(fused=True).  `jobs`: device memory, (n_jobs + 1) rows of 8 int64:
 *   [0] param  [1] grad (0 = an all-zero gradient that is never read)  [2] exp_avg  [3] exp_avg_sq  [4] bf16 image of the updated param to write as well, or 0  [5] n
 *   [6] 0 = the call's bias corrections, else float bits of (1 - beta1^step) | float bits of sqrt(1 - beta2^step) << 32 (tensors
 *   at different step counts)   [7] first block (4096 elements per block); row n_jobs: sentinel, [7] = total_blocks. */
int pangu_adam_step_multi(pangu_stream_t stream, const void* jobs, int n_jobs, long long total_blocks, double lr, double beta1,
                          double beta2, double weight_decay, double eps, float bias_correction1, float bias_correction2_sqrt);

/* Rehearsal tool (not on the product path): copy `bytes` (multiple of 16, 16-B aligned pointers) device to device with a grid of
 * exactly `workgroups` 256-thread workgroups -- the HBM traffic / CU footprint of a collective on its own stream, for measuring how
 * much a bucketed gradient all-reduce (reference era5_data/utils_dist.py:125-134 semantics) slows the backward kernels it overlaps. */
int pangu_traffic_copy(pangu_stream_t stream, const void* src, void* dst, long long bytes, int workgroups);

#ifdef __cplusplus
}
#endif
#endif /* PANGU_HIP_H */

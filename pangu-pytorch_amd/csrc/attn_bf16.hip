// Earth-specific 3D window attention forward, bf16 operands / fp32 softmax, gfx950.
//
// Same decomposition as attn_f32.hip (one (window, head) per 3-wave workgroup, transposed scores so that the
// probability registers are the B operand of the second product), on v_mfma_f32_16x16x32_bf16:
//   S^T tile [16 keys][16 queries] = K[16][32] . Q^T[32][16]  is ONE MFMA (K = head_dim = 32);
//   O^T[16 d][16 q] += V^T[16 d][32 keys] . P^T[32 keys][16 q]: five MFMAs per d-tile (144 keys padded to 160).
// With 16x fewer matrix cycles than fp32 the kernel is bound by the softmax VALU work and by HBM (qkv read,
// out write, 31 MB bias), so: K is staged as [key][32] with a chunk swizzle (conflict-free b128 fragment reads),
// V is staged TRANSPOSED ([d][160 keys], 336-B rows: conflict-free b64 fragment reads) so that the P^T accumulator
// quads of two adjacent key tiles form the 8-element B fragment with no data movement (the MFMA k index is
// permuted identically in both operands), scale is folded into the bias add (s = acc*scale + bias), and the
// bf16 bias tile is read as 8-byte quads.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int VT_LD = 336;      // bytes per V^T row (160 keys * 2 B + 16 pad)

__device__ inline u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
__device__ inline unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
__device__ inline float bflo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ inline float bfhi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

// K image: [144 keys][4 chunks of 16 B], chunk XOR F[(row>>2)&3], F = {0,2,3,1}: every ds_read_b128 lane group sees 16
// distinct 16-B slots (rows r with chunk g and rows r' with chunk g^1).
__device__ inline int kswz(int row, int chunk) {
  const int f = (0x78 >> (((row >> 2) & 3) * 2)) & 3;      // packed table F = {0,2,3,1} (2 bits each, q = 0 lowest)
  return row * 64 + ((chunk ^ f) << 4);
}

template <bool SHIFTED>
__global__ __launch_bounds__(192, 3) void window_attn_bf16_kernel(const u16* __restrict__ qkv,
                                                                  const u16* __restrict__ qkv_bias,
                                                                  const u16* __restrict__ esb, u16* __restrict__ out,
                                                                  float* __restrict__ lse, WinGeom g, int C, int heads,
                                                                  int n_pairs) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[PANGU_WTOK * 64];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[32 * VT_LD];
  __shared__ int tok_s[PANGU_WTOK];

  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int pair = (local / g.nLon) * 8 + xcd;
  const int l = local % g.nLon;
  if (pair >= n_pairs) return;
  const int t = pair / heads, hd = pair - t * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C3 = 3 * C;
  const float scale = 0.17677669529663687f;

  if (tid < PANGU_WTOK) tok_s[tid] = win_src_token(g, l, t, tid, SHIFTED);
  // zero the 16 pad keys of V^T (keys 144..159): they meet P = 0 but must be finite
  if (tid < 32) {
#pragma unroll
    for (int c = 0; c < 2; ++c) *reinterpret_cast<u32x4*>(Vt + tid * VT_LD + 288 + 16 * c) = u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  // ---- stage K ([key][32]) and V^T ([d][key]): 144 rows x 4 chunks of 8 bf16
  for (int f = tid; f < PANGU_WTOK * 4; f += 192) {
    const int n = f >> 2, ch = f & 3;
    const int tok = tok_s[n];
    const u16* src = tok >= 0 ? qkv + (size_t)tok * C3 : qkv_bias;
    const u32x4 kv = *reinterpret_cast<const u32x4*>(src + C + hd * 32 + ch * 8);
    const u32x4 vv = *reinterpret_cast<const u32x4*>(src + 2 * C + hd * 32 + ch * 8);
    *reinterpret_cast<u32x4*>(Ks + kswz(n, ch)) = kv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *reinterpret_cast<u16*>(Vt + (ch * 8 + 2 * e) * VT_LD + n * 2) = (u16)(vv[e] & 0xFFFFu);
      *reinterpret_cast<u16*>(Vt + (ch * 8 + 2 * e + 1) * VT_LD + n * 2) = (u16)(vv[e] >> 16);
    }
  }
  __syncthreads();

  const int lq = lane & 15, lg = lane >> 4;
  const u16* bias_tile = esb + (size_t)pair * PANGU_WTOK * PANGU_WTOK;
  bool zcut = false, hcut = false;
  unsigned long long kz_bits = 0ull, kh_bits = 0ull;
  if (SHIFTED) {
    const int zwin = t / g.nHw, hwin = t - zwin * g.nHw;
    zcut = zwin == g.nZw - 1;
    hcut = hwin == g.nHw - 1;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kn = j * 16 + lg * 4 + r;
        if (kn >= 72) kz_bits |= 1ull << (4 * j + r);
        if (((kn / 12) % 6) < 3) kh_bits |= 1ull << (4 * j + r);
      }
  }

  for (int qt = wave; qt < 9; qt += 3) {
    const int qn = qt * 16 + lq;
    const int qtok = tok_s[qn];
    int lz = 0;
    asm volatile("" : "+v"(lz));                 // keep the K / V^T fragment reads inside this loop (see attn_f32.hip)
    const unsigned char* Ksq = Ks + lz;
    const unsigned char* Vtq = Vt + lz;
    // Q fragment (B operand): Q[query qn][d = 8lg .. 8lg+7]
    const bf16x8 qf = *reinterpret_cast<const bf16x8*>((qtok >= 0 ? qkv + (size_t)qtok * C3 : qkv_bias) + hd * 32 + lg * 8);
    // ---- S^T = K Q^T, one MFMA per key tile; then s = acc*scale + bias^T
    f32x4 s[9];
    const u16* brow = bias_tile + (size_t)qn * PANGU_WTOK + lg * 4;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ksq + kswz(j * 16 + lq, lg));
      s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const u32x2 bq = *reinterpret_cast<const u32x2*>(brow + j * 16);
      s[j][0] = fmaf(s[j][0], scale, bflo(bq[0]));
      s[j][1] = fmaf(s[j][1], scale, bfhi(bq[0]));
      s[j][2] = fmaf(s[j][2], scale, bflo(bq[1]));
      s[j][3] = fmaf(s[j][3], scale, bfhi(bq[1]));
    }
    float mx = -INFINITY;
    if (SHIFTED) {
      if (zcut || hcut) {
        const bool zq = qn >= 72, hq = ((qn / 12) % 6) < 3;
        const unsigned long long zsel = zq ? ~kz_bits : kz_bits;
        const unsigned long long hsel = hq ? ~kh_bits : kh_bits;
        const unsigned long long cut = (zcut ? zsel : 0ull) | (hcut ? hsel : 0ull);
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((cut >> (4 * j + r)) & 1ull) s[j][r] += -100.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[j][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[j][r] - mx);
        s[j][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // ---- O^T = V^T P^T.  k-step u covers key tiles 2u, 2u+1: B fragment element e <-> key 32u + (e<4 ? 4lg+e : 16+4lg+e-4)
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      u32x4 pb;
      pb[0] = pack2(s[2 * u][0], s[2 * u][1]);
      pb[1] = pack2(s[2 * u][2], s[2 * u][3]);
      if (u < 4) {
        pb[2] = pack2(s[2 * u + 1][0], s[2 * u + 1][1]);
        pb[3] = pack2(s[2 * u + 1][2], s[2 * u + 1][3]);
      } else {
        pb[2] = 0u; pb[3] = 0u;
      }
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pb);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned char* vrow = Vtq + (dt * 16 + lq) * VT_LD + (32 * u + 4 * lg) * 2;
        const u32x2 va = *reinterpret_cast<const u32x2*>(vrow);
        const u32x2 vb = *reinterpret_cast<const u32x2*>(vrow + 32);
        const bf16x8 vf = __builtin_bit_cast(bf16x8, u32x4{va[0], va[1], vb[0], vb[1]});
        if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o0, 0, 0, 0);
        else o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o1, 0, 0, 0);
      }
    }
    // lane holds O^T[d = 16dt + 4lg + r][query = qn]
    if (qtok >= 0) {
      const float inv = 1.0f / sum;
      u16* dst = out + (size_t)qtok * C + hd * 32 + lg * 4;
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack2(o0[0] * inv, o0[1] * inv), pack2(o0[2] * inv, o0[3] * inv)};
      *reinterpret_cast<u32x2*>(dst + 16) = u32x2{pack2(o1[0] * inv, o1[1] * inv), pack2(o1[2] * inv, o1[3] * inv)};
      if (lse && lg == 0) lse[(size_t)qtok * heads + hd] = mx + __logf(sum);
    }
  }
}

}  // namespace

extern "C" int pangu_window_attn_fwd_bf16(pangu_stream_t stream, const void* qkv, const void* qkv_bias, const void* esb,
                                          void* out, float* lse, int Z, int H, int W, int C, int heads, int shifted) {
  if (!qkv || !qkv_bias || !esb || !out) return PANGU_E_NULL;
  if (Z <= 0 || H <= 0 || W <= 0 || Z % PANGU_WZ || (H + PANGU_PAD_H) % PANGU_WH || W % PANGU_WW) return PANGU_E_SHAPE;
  if (heads <= 0 || C != heads * PANGU_HEAD_DIM) return PANGU_E_SHAPE;
  const WinGeom g = make_geom(Z, H, W);
  const int n_pairs = g.types * heads;
  const int grid = ((n_pairs + 7) / 8) * 8 * g.nLon;
  hipStream_t s = (hipStream_t)stream;
  if (shifted)
    hipLaunchKernelGGL(window_attn_bf16_kernel<true>, dim3(grid), dim3(192), 0, s, (const u16*)qkv, (const u16*)qkv_bias,
                       (const u16*)esb, (u16*)out, lse, g, C, heads, n_pairs);
  else
    hipLaunchKernelGGL(window_attn_bf16_kernel<false>, dim3(grid), dim3(192), 0, s, (const u16*)qkv,
                       (const u16*)qkv_bias, (const u16*)esb, (u16*)out, lse, g, C, heads, n_pairs);
  return pangu_launch_status();
}

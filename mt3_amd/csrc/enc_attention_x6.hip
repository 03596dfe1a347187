// Encoder self-attention of the f32 engine on the bf16 matrix pipes (round 4; the attention of mt3/layers.py:134-157 inside
// `MultiHeadDotProductAttention`, layers.py:160-314, as mt3/network.py:44-66 calls it: no mask, no bias, no 1/sqrt(d)).
//
// The f32 encoder's dense layers already multiply that way (gemm.hip: gemm_x6_kernel): an f32 value is EXACTLY
// hi + mid + lo + r with three bf16 terms and |r| <= 2^-27 |x|, so a product of two f32 operands is six bf16 products
// (mm, hl, lh, hm, mh, hh; smallest first) accumulated in f32 -- at least as exact as v_mfma_f32_16x16x4_f32 at 2.7x its
// rate (profiles/r4_mfma_bf16_accuracy.txt).  The attention was the part of the f32 encoder still on the f32 instruction:
// enc_attn_kernel<float, 256> holds K and V^T of a head in 141 KB of LDS (one four-wave workgroup per CU) and took 0.53 ms
// per layer, a fifth of the f32 encoder (profiles/r4_f32_kernel_stats.csv), bound by that instruction (its pipes 0.47
// busy, profiles/r4_pmc_encoder_summary.json; 4 / 8 / 16 waves: within 2 %, r4_ab_f32_encoder_attention_waves.txt).
//
// Here Q, K, V and the probabilities P are split into their three bf16 planes (Q when a wave loads its query tile, K and V
// while they are staged, P in registers after the softmax) and both products run as 6 x v_mfma_f32_16x16x32_bf16.  The
// scores, the online softmax (running maximum, running sum) and the output stay f32, in the same order over the keys as
// the f32 kernel (64-key chunks, 16-key blocks inside).
// Layout: a workgroup owns T / QS queries of one (batch, head) -- one 16-query tile per wave -- and walks over the keys in
// parts of TK = 64 (K planes [3][64][64 + 16] and V^T planes [3][64][64 + 8] bf16: 58 KB of LDS, two workgroups per CU, so
// one's staging overlaps the other's products).  S^T = K Q^T comes out of the MFMA with keys along the rows, which is the
// A-operand layout of P for P V without a transpose (as in enc_attn_kernel).
#include <hip/hip_runtime.h>

#include "common.h"
#include "device.h"
#include "kernels.h"

namespace mt3k {

namespace {

// x = hi + mid + lo (+ r, |r| <= 2^-27 |x|): hi = rne(x), mid = rne(x - hi), lo = rne(x - hi - mid); both differences exact
__device__ __forceinline__ void split3(const float (&x)[8], u32x4* h, u32x4* m, u32x4* l) {
  float hf[8], mf[8], lf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hi = static_cast<__bf16>(x[i]);
    const float r1 = x[i] - static_cast<float>(hi);
    const __bf16 mi = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(mi);
    hf[i] = static_cast<float>(hi);
    mf[i] = static_cast<float>(mi);
    lf[i] = r2;
  }
  *h = pack_bf16x8(hf);
  *m = pack_bf16x8(mf);
  *l = pack_bf16x8(lf);
}

// acc += a . b for f32 operands given as planes {hi, mid, lo}: six bf16 products, smallest terms first
__device__ __forceinline__ void mfma_x6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4& acc) {
  mfma_chunk<__bf16>(a[1], b[1], acc);   // mid . mid
  mfma_chunk<__bf16>(a[0], b[2], acc);   // hi  . lo
  mfma_chunk<__bf16>(a[2], b[0], acc);   // lo  . hi
  mfma_chunk<__bf16>(a[0], b[1], acc);   // hi  . mid
  mfma_chunk<__bf16>(a[1], b[0], acc);   // mid . hi
  mfma_chunk<__bf16>(a[0], b[0], acc);   // hi  . hi
}

template <int T, int QS, int NW, int TK = 64>
__global__ __launch_bounds__(NW * 64) void enc_attn_x6_kernel(const float* __restrict__ qkv, float* __restrict__ out, int H) {
  constexpr int D = 64, KH = T / TK;
  constexpr int ROWK = D + 16;                  // K plane row (bf16 elements): 160 B = 32 mod 64: conflict-free b128 reads
  constexpr int ROWV = TK + 8;                  // V^T plane row (bf16 elements)
  static_assert(T / QS == NW * 16, "one 16-query tile per wave");
  static_assert(TK % 64 == 0 && T % TK == 0, "keys are staged in whole 64-key chunks");

  __shared__ __attribute__((aligned(16))) __bf16 Ks[3][TK * ROWK];
  __shared__ __attribute__((aligned(16))) __bf16 Vt[3][D * ROWV];

  const int bh = blockIdx.x / QS, qs = blockIdx.x % QS;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int RS = 3 * H * D, HD = H * D;         // qkv row stride; offset of K (V at 2 HD) inside a row
  const float* base = qkv + static_cast<size_t>(b) * T * RS + h * D;
  const int fr = lane & 15, fg = lane >> 4;

  // Q^T as the B operand: col n = query q0 + fr, k-group c covers d = c * 32 + fg * 8 .. + 7
  const int q0 = qs * (T / QS) + wave * 16;
  u32x4 qf[2][3];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float4* src = reinterpret_cast<const float4*>(base + static_cast<size_t>(q0 + fr) * RS + c * 32 + fg * 8);
    const float4 x0 = src[0], x1 = src[1];
    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    split3(f, &qf[c][0], &qf[c][1], &qf[c][2]);
  }
  float m = -3.0e38f, l = 0.f;                   // running max (uniform over the 4 lane groups), lane-partial sum
  f32x4 o[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) o[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int part = 0; part < KH; ++part) {
    if (part) __syncthreads();                  // every wave is done with the previous part
    const int key0 = part * TK;
    // ---- K: thread -> (key row, 8-element piece), split into the three planes
    for (int w = tid; w < TK * D / 8; w += NW * 64) {
      const int row = w >> 3, ch = w & 7;
      const float4* src = reinterpret_cast<const float4*>(base + static_cast<size_t>(key0 + row) * RS + HD + ch * 8);
      const float4 x0 = src[0], x1 = src[1];
      const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      u32x4 p[3];
      split3(f, &p[0], &p[1], &p[2]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(&Ks[pl][row * ROWK + ch * 8]) = p[pl];
    }
    // ---- V transposed: work item = (key pair, 8-element piece of d); planes written as bf16 pairs {key, key + 1}
    for (int w = tid; w < (TK / 2) * 8; w += NW * 64) {
      const int rp = w % (TK / 2), ch = w / (TK / 2);
      const float* src = base + static_cast<size_t>(key0 + 2 * rp) * RS + 2 * HD + ch * 8;
      const float4 a0 = reinterpret_cast<const float4*>(src)[0], a1 = reinterpret_cast<const float4*>(src)[1];
      const float4 c0 = reinterpret_cast<const float4*>(src + RS)[0], c1 = reinterpret_cast<const float4*>(src + RS)[1];
      const float fa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float fc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      u32x4 pa[3], pc[3];
      split3(fa, &pa[0], &pa[1], &pa[2]);
      split3(fc, &pc[0], &pc[1], &pc[2]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // element j of a packed chunk: half (j & 1) of dword j / 2
          const unsigned ea = (pa[pl][j >> 1] >> ((j & 1) * 16)) & 0xffffu, ec = (pc[pl][j >> 1] >> ((j & 1) * 16)) & 0xffffu;
          *reinterpret_cast<unsigned*>(&Vt[pl][(ch * 8 + j) * ROWV + 2 * rp]) = ea | (ec << 16);
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int k64 = 0; k64 < TK / 64; ++k64) {
    // ---- S^T block j: rows = keys (k64 * 4 + j) * 16 + fg * 4 + r of the part, col = query fr
    f32x4 sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 kf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          kf[pl] = *reinterpret_cast<const u32x4*>(&Ks[pl][((k64 * 4 + j) * 16 + fr) * ROWK + c * 32 + fg * 8]);
        mfma_x6(kf, qf[c], sc[j]);
      }
    }
    float cm = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cm = fmaxf(cm, sc[j][r]);
    cm = fmaxf(cm, __shfl_xor(cm, 16));
    cm = fmaxf(cm, __shfl_xor(cm, 32));
    // e^d as v_exp_f32(d * log2(e)) on the f32 difference d = score - max (<= 0): three VALU instructions per
    // probability where libm's expf takes ~30 -- with it the kernel was VALU-bound (17 exponentials per lane and chunk).
    // The difference is formed in f32 first, exactly as expf's argument would be; the extra rounding of d * log2(e) is
    // <= 2^-24 |d|, the size of the rounding already in d.
    constexpr float kLog2e = 1.4426950408889634f;
    const float mn = fmaxf(m, cm);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * kLog2e);   // 0 on the first chunk
    m = mn;
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f((sc[j][r] - mn) * kLog2e);
        sc[j][r] = p;
        ls += p;
      }
    l = l * alpha + ls;
    // rescale O: its rows are queries fg * 4 + r, whose alpha lives in lane fg * 4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(alpha, fg * 4 + r);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) o[nb][r] *= ar;
    }
    // ---- O += P V: A = P (row = query fr; this lane's 8 K-slots = its keys of blocks 2 kc and 2 kc + 1), B = V^T rows
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const float pv[8] = {sc[2 * kc][0],     sc[2 * kc][1],     sc[2 * kc][2],     sc[2 * kc][3],
                           sc[2 * kc + 1][0], sc[2 * kc + 1][1], sc[2 * kc + 1][2], sc[2 * kc + 1][3]};
      u32x4 pa[3];
      split3(pv, &pa[0], &pa[1], &pa[2]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        u32x4 vb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const __bf16* vrow = &Vt[pl][(nb * 16 + fr) * ROWV + k64 * 64 + fg * 4];
          const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + (2 * kc) * 16);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (2 * kc + 1) * 16);
          vb[pl] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
        mfma_x6(pa, vb, o[nb]);
      }
    }
    }
  }

  // normalise and store the wave's 16-query tile: O fragment row = query fg * 4 + r, col = d = nb * 16 + fr; 1 / l of that
  // query lives in lane fg * 4 + r
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float linv = 1.f / l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float li = __shfl(linv, fg * 4 + r);
    float* dst = out + (static_cast<size_t>(b) * T + q0 + fg * 4 + r) * HD + h * D + fr;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) dst[nb * 16] = o[nb][r] * li;
  }
}

}  // namespace

// f32 qkv [B * T][3 * H * 64] -> f32 out [B * T][H * 64]; T = 256 or 512
int launch_encoder_attention_x6(const void* qkv, void* out, int B, int T, int H, hipStream_t s) {
  if (!qkv || !out || B <= 0 || H <= 0) return mt3::fail(MT3_ERR_INVALID, "encoder_attention_x6: bad arguments");
  const float* q = static_cast<const float*>(qkv);
  float* o = static_cast<float*>(out);
  if (T == 256) {
    hipLaunchKernelGGL((enc_attn_x6_kernel<256, 2, 8>), dim3(B * H * 2), dim3(512), 0, s, q, o, H);
  } else if (T == 512) {
    hipLaunchKernelGGL((enc_attn_x6_kernel<512, 4, 8>), dim3(B * H * 4), dim3(512), 0, s, q, o, H);
  } else {
    return mt3::fail(MT3_ERR_INVALID, "encoder_attention_x6: supported T are 256 and 512");
  }
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

}  // namespace mt3k

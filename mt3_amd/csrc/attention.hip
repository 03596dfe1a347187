// Attention kernels of the MT3 path on gfx950.
//
// Reference semantics (mt3/layers.py:85-157 `dot_product_attention`, called from
// MultiHeadDotProductAttention, layers.py:164-355): logits = Q K^T with NO 1/sqrt(d)
// scaling (layers.py:230-234), f32 softmax over keys, then P V.  The encoder mask is all ones
// (network.py:283-289); the cached decoder attends positions 0..t (layers.py:297-305), which
// here is a loop bound instead of a -1e10 bias.
//
// (1) enc_attn_kernel: one workgroup per (batch, head), T = 256/512 keys, head_dim 64.
//     K [T][64] and V^T [64][T] of the head are staged into LDS ONCE (each K/V byte is read
//     from HBM exactly once); every wave then owns 16-query tiles:
//       S^T = K Q^T on the matrix pipe with K as the A operand, so that a lane ends up holding,
//       for ITS query (lane & 15), the scores of keys kb*16 + (lane>>4)*4 + r -- the softmax
//       row reduction is in-lane plus two cross-lane steps (xor 16, 32), and the probabilities
//       are already laid out as the A operand of the P V product (two 16-key blocks = one
//       32-wide K-group): no LDS round trip for P.
//       V is consumed as V^T rows so the B operand is K-contiguous (8-byte LDS reads).
// (2) dec_attn_kernel: single-query attention of one decode step -- pure HBM streaming of the
//     K/V cache [B][H][cap][64]; LPK lanes share a key (16 bytes each), a wave covers 64/LPK
//     consecutive keys per load (1 KiB contiguous), 4-deep unrolled so every lane keeps 8 non-temporal
//     loads in flight.  The arithmetic is kept thin so that it never throttles the stream: q.k on the
//     packed bf16 operands (v_dot2c_f32_bf16), key-group sums on the DPP network, base-2 online softmax
//     with ONE rescale per 4 keys, a wave-uniform branch-free key loop (out-of-range slots re-read the
//     last cached key with weight 0); lane groups merged by shuffles, waves through LDS.  With APPEND the
//     step's new K/V row is folded in from registers and written to the cache at position step[b] by
//     the same kernel (the reference rewrites the WHOLE cache per step with a one-hot multiply-add:
//     layers.py:272-292).
#include <hip/hip_runtime.h>

#include "common.h"
#include "device.h"
#include "kernels.h"

namespace mt3k {

// ------------------------------------------------------------------------ encoder attention
// stage keys [key0, key0 + TK) of one head into LDS: K row-major [TK][ROWK], V transposed [64][ROWV]
template <typename CT, int TK, int ROWK, int ROWV, int NT = 256>
__device__ __forceinline__ void enc_stage_kv(const CT* base, int RS, int HD, int key0, CT* Ks, CT* Vt, int tid) {
  constexpr int KPL = CTraits<CT>::KPL;
  constexpr int D = 64;
  constexpr int CH = D / KPL;                   // chunks per K/V row
  // ---- K (row-major) : chunk c -> row c / CH, piece c % CH
  for (int c = tid; c < TK * CH; c += NT) {
    const int row = c / CH, ch = c % CH;
    const u32x4 v = *reinterpret_cast<const u32x4*>(base + static_cast<size_t>(key0 + row) * RS + HD + ch * KPL);
    *reinterpret_cast<u32x4*>(&Ks[row * ROWK + ch * KPL]) = v;
  }
  // ---- V transposed: work item = (key pair, piece); a wave takes 64 consecutive key pairs
  for (int w = tid; w < (TK / 2) * CH; w += NT) {
    const int rp = w % (TK / 2), ch = w / (TK / 2);
    const CT* src = base + static_cast<size_t>(key0 + 2 * rp) * RS + 2 * HD + ch * KPL;
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(src);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(src + RS);
    if constexpr (KPL == 8) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, v0), c = __builtin_bit_cast(bf16x8, v1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 pr = {a[j], c[j]};
        *reinterpret_cast<bf16x2*>(&Vt[(ch * 8 + j) * ROWV + 2 * rp]) = pr;
      }
    } else {
      const f32x4 a = __builtin_bit_cast(f32x4, v0), c = __builtin_bit_cast(f32x4, v1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float2*>(&Vt[(ch * 4 + j) * ROWV + 2 * rp]) = make_float2(a[j], c[j]);
    }
  }
}

// one 64-key chunk (keys k64*64 .. +63 of the STAGED range) folded into the online-softmax state (m, l, o) of
// the wave's 16-query tile whose Q^T fragments are qf
template <typename CT, int ROWK, int ROWV>
__device__ __forceinline__ void enc_attn_chunk(const CT* Ks, const CT* Vt, int k64,
                                               const u32x4 (&qf)[64 / CTraits<CT>::KGROUP], float& m, float& l,
                                               f32x4 (&o)[4], int fr, int fg) {
  constexpr int KPL = CTraits<CT>::KPL;
  constexpr int KG = CTraits<CT>::KGROUP;
  constexpr int NC = 64 / KG;
  // S^T block j: rows = keys (k64*4 + j)*16 + fg*4 + r, col = query fr
  f32x4 sc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const u32x4 kf = *reinterpret_cast<const u32x4*>(&Ks[((k64 * 4 + j) * 16 + fr) * ROWK + c * KG + fg * KPL]);
      mfma_chunk<CT>(kf, qf[c], sc[j]);
    }
  }
  float cm = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cm = fmaxf(cm, sc[j][r]);
  cm = fmaxf(cm, __shfl_xor(cm, 16));
  cm = fmaxf(cm, __shfl_xor(cm, 32));
  float alpha, ls = 0.f;
  if constexpr (sizeof(CT) == 2) {
    // bf16 path: the kernel is VALU-bound (16 MFMAs against ~300 VALU instructions per chunk with libm's expf), so
    // the running maximum lives in the base-2 domain and every probability is ONE v_fma_f32 + ONE v_exp_f32
    constexpr float kLog2e = 1.4426950408889634f;
    const float mn = fmaxf(m, cm * kLog2e);
    alpha = __builtin_amdgcn_exp2f(m - mn);    // 0 on the first chunk
    m = mn;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[j][r], kLog2e, -mn));
        sc[j][r] = p;
        ls += p;
      }
  } else {
    const float mn = fmaxf(m, cm);
    alpha = expf(m - mn);                      // 0 on the first chunk
    m = mn;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(sc[j][r] - mn);
        sc[j][r] = p;
        ls += p;
      }
  }
  l = l * alpha + ls;
  // rescale O: its rows are queries fg*4 + r, whose alpha lives in lane fg*4 + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ar = __shfl(alpha, fg * 4 + r);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb][r] *= ar;
  }
  // O += P V : A = P (row = query fr, K-slots = this lane's keys), B = V^T rows (col = d)
  if constexpr (KPL == 8) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const float pv[8] = {sc[2 * kc][0],     sc[2 * kc][1],     sc[2 * kc][2],     sc[2 * kc][3],
                           sc[2 * kc + 1][0], sc[2 * kc + 1][1], sc[2 * kc + 1][2], sc[2 * kc + 1][3]};
      const u32x4 pa = pack_bf16x8(pv);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const CT* vrow = &Vt[(nb * 16 + fr) * ROWV + k64 * 64 + fg * 4];
        const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + (2 * kc) * 16);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (2 * kc + 1) * 16);
        const u32x4 vb = u32x4{lo.x, lo.y, hi.x, hi.y};
        mfma_chunk<CT>(pa, vb, o[nb]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 pa = pack_f32x4(sc[j][0], sc[j][1], sc[j][2], sc[j][3]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const u32x4 vb = *reinterpret_cast<const u32x4*>(&Vt[(nb * 16 + fr) * ROWV + (k64 * 4 + j) * 16 + fg * 4]);
        mfma_chunk<CT>(pa, vb, o[nb]);
      }
    }
  }
}

// normalise and store the wave's 16-query output tile: O fragment row = query fg*4 + r, col = d = nb*16 + fr;
// 1/l of that query lives in lane fg*4 + r
template <typename CT>
__device__ __forceinline__ void enc_attn_store(CT* out, size_t row0, int HD, int h, float l, const f32x4 (&o)[4],
                                               int fr, int fg) {
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float linv = 1.f / l;
  float li[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) li[r] = __shfl(linv, fg * 4 + r);
  if constexpr (sizeof(CT) == 2) {
    // dword stores: lanes fr and fr ^ 1 hold adjacent columns, they swap over DPP and the even lane writes rows
    // r = 0, 1, the odd lane rows r = 2, 3 of the pair (a 2-byte store costs a read-modify-write in the cache)
    const int odd = fr & 1;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      float mine[4], other[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mine[r] = o[nb][r] * li[r];
        other[r] = lane_xor1(mine[r]);
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float lo = odd ? other[2 + hh] : mine[hh], hi = odd ? mine[2 + hh] : other[hh];
        CT* dst = out + (row0 + fg * 4 + odd * 2 + hh) * HD + h * 64 + nb * 16 + fr - odd;
        *reinterpret_cast<unsigned*>(dst) = pack_bf16x2(lo, hi);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      CT* dst = out + (row0 + fg * 4 + r) * HD + h * 64 + fr;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) dst[nb * 16] = to_ct<CT>(o[nb][r] * li[r]);
    }
  }
}

// NW waves per (batch, head) workgroup: 4, or (round 3, bf16 T = 256) 8 -- 114 VGPRs and 74.7 KB of LDS allow two such
// workgroups per CU = 16 waves instead of 8: twice the loads in flight while K / V^T are staged and twice the waves to
// hide the softmax's VALU chains behind (the kernel is latency-, not MFMA-bound: matrix pipes 13 % busy)
template <typename CT, int T, int NW = 4>
__global__ __launch_bounds__(NW * 64) void enc_attn_kernel(const CT* __restrict__ qkv, CT* __restrict__ out, int H) {
  constexpr int KPL = CTraits<CT>::KPL;
  constexpr int KG = CTraits<CT>::KGROUP;       // head-dim / key elements per chunk-MFMA
  constexpr int D = 64;
  constexpr int ROWK = D + 2 * KPL;             // K tile row (elements): stride 32 mod 64 bytes, conflict-free b128 reads
  constexpr int ROWV = T + 8;                   // V^T row (elements): keeps 8/16-byte alignment
  constexpr int NC = D / KG;                    // K-groups across the head dim

  __shared__ __attribute__((aligned(16))) CT Ks[T * ROWK];
  __shared__ __attribute__((aligned(16))) CT Vt[D * ROWV];

  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int RS = 3 * H * D;                     // qkv row stride (elements)
  const CT* base = qkv + static_cast<size_t>(b) * T * RS + h * D;

  enc_stage_kv<CT, T, ROWK, ROWV, NW * 64>(base, RS, H * D, 0, Ks, Vt, tid);
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int qt = wave; qt < T / 16; qt += NW) {
    const int q0 = qt * 16;
    // Q^T as the B operand: col n = query q0 + fr, K-group chunk c
    u32x4 qf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
      qf[c] = *reinterpret_cast<const u32x4*>(base + static_cast<size_t>(q0 + fr) * RS + c * KG + fg * KPL);

    // online softmax over 64-key chunks (4 blocks of 16 keys): keeps the live score registers at 16
    // per lane instead of T/4, so T = 512 and the f32 path stay out of scratch.
    float m = -3.0e38f, l = 0.f;                 // running max (uniform over the 4 lane groups), lane-partial sum
    f32x4 o[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int k64 = 0; k64 < T / 64; ++k64) enc_attn_chunk<CT, ROWK, ROWV>(Ks, Vt, k64, qf, m, l, o, fr, fg);
    enc_attn_store<CT>(out, static_cast<size_t>(b) * T + q0, H * D, h, l, o, fr, fg);
  }
}

// The same attention when K and V^T of a whole head do not fit the 160 KB of LDS (f32 operands at T = 512: 280 KB):
// the keys are staged in KH parts, the workgroup owns only T / QS queries (grid = B * H * QS) so that the
// online-softmax state of ALL its query tiles (TPW per wave) stays in registers across the parts.  Same chunk
// order per query as the one-pass kernel, i.e. the same arithmetic.
template <typename CT, int T, int KH, int QS>
__global__ __launch_bounds__(256) void enc_attn_split_kernel(const CT* __restrict__ qkv, CT* __restrict__ out, int H) {
  constexpr int KPL = CTraits<CT>::KPL;
  constexpr int KG = CTraits<CT>::KGROUP;
  constexpr int D = 64;
  constexpr int TK = T / KH;                    // keys staged at a time
  constexpr int ROWK = D + 2 * KPL;
  constexpr int ROWV = TK + 8;
  constexpr int NC = D / KG;
  constexpr int TPW = T / QS / 16 / 4;          // query tiles per wave
  static_assert(TPW >= 1 && TK % 64 == 0, "split attention: tile/part sizes");

  __shared__ __attribute__((aligned(16))) CT Ks[TK * ROWK];
  __shared__ __attribute__((aligned(16))) CT Vt[D * ROWV];

  const int bh = blockIdx.x / QS, qs = blockIdx.x % QS;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int RS = 3 * H * D;
  const CT* base = qkv + static_cast<size_t>(b) * T * RS + h * D;
  const int fr = lane & 15, fg = lane >> 4;

  u32x4 qf[TPW][NC];
  float m[TPW], l[TPW];
  f32x4 o[TPW][4];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int q0 = qs * (T / QS) + (ti * 4 + wave) * 16;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      qf[ti][c] = *reinterpret_cast<const u32x4*>(base + static_cast<size_t>(q0 + fr) * RS + c * KG + fg * KPL);
    m[ti] = -3.0e38f;
    l[ti] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) o[ti][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll 1
  for (int part = 0; part < KH; ++part) {
    if (part) __syncthreads();                  // every wave is done with the previous part
    enc_stage_kv<CT, TK, ROWK, ROWV>(base, RS, H * D, part * TK, Ks, Vt, tid);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
#pragma unroll 1
      for (int k64 = 0; k64 < TK / 64; ++k64)
        enc_attn_chunk<CT, ROWK, ROWV>(Ks, Vt, k64, qf[ti], m[ti], l[ti], o[ti], fr, fg);
    }
  }
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int q0 = qs * (T / QS) + (ti * 4 + wave) * 16;
    enc_attn_store<CT>(out, static_cast<size_t>(b) * T + q0, H * D, h, l[ti], o[ti], fr, fg);
  }
}

int launch_encoder_attention(int dtype, const void* qkv, void* out, int B, int T, int H, hipStream_t s) {
  if (!qkv || !out || B <= 0 || H <= 0) return mt3::fail(MT3_ERR_INVALID, "encoder_attention: bad arguments");
  const dim3 grid(B * H), block(256);
  if (dtype == MT3_BF16 && T == 256) {
    // eight waves per (batch, head) workgroup: two such workgroups per CU (r3: 78 against 104 us per launch with four)
    hipLaunchKernelGGL((enc_attn_kernel<__bf16, 256, 8>), grid, dim3(512), 0, s, static_cast<const __bf16*>(qkv),
                       static_cast<__bf16*>(out), H);
  } else if (dtype == MT3_BF16 && T == 512) {
    hipLaunchKernelGGL((enc_attn_kernel<__bf16, 512>), grid, block, 0, s, static_cast<const __bf16*>(qkv),
                       static_cast<__bf16*>(out), H);
  } else if (dtype == MT3_F32 && T == 256) {
    hipLaunchKernelGGL((enc_attn_kernel<float, 256>), grid, block, 0, s, static_cast<const float*>(qkv),
                       static_cast<float*>(out), H);
  } else if (dtype == MT3_F32 && T == 512) {
    // ismir2021 preset at the reference's precision: keys in two parts, four query quarters per (batch, head)
    hipLaunchKernelGGL((enc_attn_split_kernel<float, 512, 2, 4>), dim3(B * H * 4), block, 0, s,
                       static_cast<const float*>(qkv), static_cast<float*>(out), H);
  } else {
    return mt3::fail(MT3_ERR_INVALID, "encoder_attention: supported T are 256 and 512 (bf16 and f32)");
  }
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// ------------------------------------------------------------------------- decode attention
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// q . k over one 16-byte chunk.  bf16: 4 x v_dot2c_f32_bf16 straight on the packed operands (exact
// products, f32 accumulate) -- neither q nor k is unpacked; f32: 4 FMAs.
template <typename CT>
__device__ __forceinline__ float chunk_dot(const u32x4& q, const u32x4& k);
template <>
__device__ __forceinline__ float chunk_dot<__bf16>(const u32x4& q, const u32x4& k) {
  const bf16x8 a = __builtin_bit_cast(bf16x8, q), b = __builtin_bit_cast(bf16x8, k);
  float s = 0.f;
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), s, false);
  return s;
}
template <>
__device__ __forceinline__ float chunk_dot<float>(const u32x4& q, const u32x4& k) {
  const f32x4 a = __builtin_bit_cast(f32x4, q), b = __builtin_bit_cast(f32x4, k);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s = __builtin_fmaf(a[j], b[j], s);
  return s;
}

// sum over the LPK (8 or 16) consecutive lanes that share a key, on the DPP network (no LDS traffic):
// xor 1 and xor 2 inside a quad, then the mirrored half-row / row holds the other partial sum
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LPK>
__device__ __forceinline__ float key_group_sum(float v) {
  v = dpp_add<0xB1>(v);                          // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                          // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);                         // row_half_mirror: lane i <-> 7 - i
  if constexpr (LPK == 16) v = dpp_add<0x140>(v);   // row_mirror: lane i <-> 15 - i
  return v;
}

// 1 / rms of residual row b from its n (<= 64, a multiple of 4) partial sums of squares: lanes 0 .. n/4 - 1 of the
// wave's first DPP row fetch a float4 each, the row is summed on the DPP network, lane 0's value is broadcast through
// an SGPR -- the same order in every wave, a few VALU cycles instead of a six-step ds_bpermute butterfly
__device__ __forceinline__ float row_rs_from_partials(const float* ss, int n, int lane) {
  float part = 0.f;
  if (lane < (n >> 2)) {
    const float4 v = reinterpret_cast<const float4*>(ss)[lane];
    part = ((v.x + v.y) + v.z) + v.w;
  }
  part = row16_sum(part);
  const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, part)));
  return rsqrtf(tot / static_cast<float>(16 * n) + 1e-6f);
}


// Every field of DecAttnArgs in SGPRs behind ONE scalar-memory round trip (see gemm.hip: gemm_kernel): left to itself the
// compiler sinks the s_loads next to their first use -- shape, then the retirement pointers, then the cache pointers,
// each a dependent trip to the kernel-argument segment in front of the first K/V request.  (MT3_EXP bit 6: off, for A/B.)
#if defined(MT3_EXP) && (MT3_EXP & 64)
#define MT3_PIN_DEC_ATTN_ARGS(a)
#else
#define MT3_PIN_DEC_ATTN_ARGS(a)                                                                                        \
  asm volatile("" ::"s"((a).q), "s"((a).q_stride), "s"((a).kcache), "s"((a).vcache), "s"((a).cap), "s"((a).new_k),       \
               "s"((a).new_v), "s"((a).kv_stride), "s"((a).step), "s"((a).n_keys), "s"((a).out), "s"((a).B), "s"((a).H), \
               "s"((a).kv_scale), "s"((a).q_f32), "s"((a).q_ss), "s"((a).q_ss_n), "s"((a).done), "s"((a).cache_row))
#endif
template <typename CT, bool APPEND, int NW, bool QF32 = false>
__global__ __launch_bounds__(NW * 64) void dec_attn_kernel(DecAttnArgs a) {
  constexpr int KPL = CTraits<CT>::KPL;
  constexpr int D = 64;
  constexpr int LPK = D / KPL;            // lanes sharing one key (8 bf16 / 16 f32)
  constexpr int KPW = 64 / LPK;           // keys per wave per load
  constexpr int STRIDE = NW * KPW;        // keys per block iteration
  constexpr int UNROLL = 4;
  constexpr float kLog2e = 1.4426950408889634f;

  __shared__ float s_m[NW][LPK], s_l[NW][LPK], s_acc[NW][LPK][KPL];

  MT3_PIN_DEC_ATTN_ARGS(a);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, slot = lane / LPK;
  // Row retirement (a.done != nullptr; uniform over the launch, both loads scalar): a finished slot costs no cache
  // byte -- the workgroup returns before its first request (its `out` row keeps the previous step's finite values; only
  // the slot's own, ignored, residual row ever sees them); a live slot finds its cache row through the slot map.
  int crow = b;
  if (a.done) {
    if (a.done[b]) return;
    if (a.cache_row) crow = a.cache_row[b];
  }
  const CT* kc = static_cast<const CT*>(a.kcache) + (static_cast<size_t>(crow) * a.H + h) * a.cap * D;
  const CT* vc = static_cast<const CT*>(a.vcache) + (static_cast<size_t>(crow) * a.H + h) * a.cap * D;
  // The first group of keys is requested BEFORE the row's position counter is known (its load would otherwise sit
  // in front of every cache load: one dependent memory round trip per launch): positions past the row's length are
  // discarded below, the memory behind them is always addressable (the cache is allocated to `cap` rows); its
  // CONTENTS do not matter (tests/test_gpu_kernels.py poisons them with NaN patterns).
  u32x4 kv0[UNROLL], vv0[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const int key = min(wave * KPW + slot + u * STRIDE, a.cap - 1);
    kv0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kc + static_cast<size_t>(key) * D + sub * KPL));
    vv0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vc + static_cast<size_t>(key) * D + sub * KPL));
  }
  const int n_keys = a.step ? (a.step[b] + 1) : a.n_keys;      // per-row position counter
  const int pos = n_keys - 1;
  const int n_cache = APPEND ? pos : n_keys;                   // keys that come from the cache
  // what the speculative loads fetched from beyond the row's length is DISCARDED BY POSITION: the V chunk is replaced
  // by zeros (a zero weight alone would turn a stale Inf / NaN pattern into NaN through 0 * x), the score is never
  // looked at (selects in `fold`).  Later groups clamp their addresses to the row's own last cached key instead.
#pragma unroll
  for (int u = 0; u < UNROLL; ++u)
    if (wave * KPW + slot + u * STRIDE >= n_cache) vv0[u] = u32x4{0u, 0u, 0u, 0u};

  // QF32: q (and with APPEND the step's new K/V row) arrive as UNNORMALISED f32 products plus the partial sums of
  // squares of the residual row they were projected from (the projection rode in the neighbouring GEMM launches):
  // 1/rms is applied here, then the values are rounded to the compute type exactly where the GEMM epilogue would have.
  // ALL raw loads are issued before the reduction that yields 1/rms (that reduction waits on ITS load; anything issued
  // after it would cost a second dependent memory round trip per launch -- measured: 0.9 us).
  constexpr int NRAW = QF32 ? (APPEND ? 3 : 1) : 1;
  float4 raw[NRAW][2];
  if constexpr (QF32) {
    const float* src[3] = {a.q_f32 + static_cast<size_t>(b) * a.q_stride + h * D + sub * KPL,
                           APPEND ? static_cast<const float*>(a.new_k) + static_cast<size_t>(b) * a.kv_stride + h * D + sub * KPL : nullptr,
                           APPEND ? static_cast<const float*>(a.new_v) + static_cast<size_t>(b) * a.kv_stride + h * D + sub * KPL : nullptr};
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      raw[i][0] = *reinterpret_cast<const float4*>(src[i]);
      raw[i][1] = KPL == 8 ? *reinterpret_cast<const float4*>(src[i] + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float rs = 1.f;
  if constexpr (QF32) rs = row_rs_from_partials(a.q_ss + static_cast<size_t>(b) * a.q_ss_n, a.q_ss_n, lane);
  auto scaled_chunk = [&](const float4 (&r)[2]) -> u32x4 {
    if constexpr (KPL == 8) {
      const float f[8] = {r[0].x * rs, r[0].y * rs, r[0].z * rs, r[0].w * rs, r[1].x * rs, r[1].y * rs, r[1].z * rs, r[1].w * rs};
      return pack_bf16x8(f);
    } else {
      return pack_f32x4(r[0].x * rs, r[0].y * rs, r[0].z * rs, r[0].w * rs);
    }
  };
  u32x4 new_k = {0u, 0u, 0u, 0u}, new_v = {0u, 0u, 0u, 0u};
  if constexpr (APPEND) {
    // this step's K/V row: folded in below from registers, persisted for the later steps
    if constexpr (QF32) {
      new_k = scaled_chunk(raw[NRAW - 2]);
      new_v = scaled_chunk(raw[NRAW - 1]);
    } else {
      new_k = *reinterpret_cast<const u32x4*>(static_cast<const CT*>(a.new_k) + static_cast<size_t>(b) * a.kv_stride +
                                              h * D + sub * KPL);
      new_v = *reinterpret_cast<const u32x4*>(static_cast<const CT*>(a.new_v) + static_cast<size_t>(b) * a.kv_stride +
                                              h * D + sub * KPL);
    }
    if (tid < LPK) {
      const size_t at = ((static_cast<size_t>(crow) * a.H + h) * a.cap + pos) * D + sub * KPL;
      *reinterpret_cast<u32x4*>(static_cast<CT*>(a.kcache) + at) = new_k;
      *reinterpret_cast<u32x4*>(static_cast<CT*>(a.vcache) + at) = new_v;
    }
  }

  u32x4 qc;
  if constexpr (QF32) {
    qc = scaled_chunk(raw[0]);
  } else {
    qc = *reinterpret_cast<const u32x4*>(static_cast<const CT*>(a.q) + static_cast<size_t>(b) * a.q_stride + h * D +
                                         sub * KPL);
  }

  // Online softmax in base 2 (scores scaled by log2 e once; v_exp_f32 is 2^x) with ONE rescale of the
  // running state per group of UNROLL keys.  Out-of-range keys re-read the row's last cached key and get
  // weight 0, so the loop is wave-uniform and branch-free; q.k runs on the packed operands.
  float m = -1.0e30f, l = 0.f, acc[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) acc[j] = 0.f;

  // one group of UNROLL x KPW x NW keys starting at `base`, already in registers
  auto fold = [&](const u32x4 (&kv)[UNROLL], const u32x4 (&vv)[UNROLL], int base) {
    float sc[UNROLL];
    float mn = m;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      sc[u] = key_group_sum<LPK>(chunk_dot<CT>(qc, kv[u])) * kLog2e;
      if (base + slot + u * STRIDE < n_cache) mn = fmaxf(mn, sc[u]);
    }
    const float rs = __builtin_amdgcn_exp2f(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) acc[j] *= rs;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float p = base + slot + u * STRIDE < n_cache ? __builtin_amdgcn_exp2f(sc[u] - mn) : 0.f;
      psum += p;
      float vf[KPL];
      unpack_chunk<CT>(vv[u], vf);
#pragma unroll
      for (int j = 0; j < KPL; ++j) acc[j] = __builtin_fmaf(p, vf[j], acc[j]);
    }
    l = l * rs + psum;
    m = mn;
  };
  if (wave * KPW < n_cache) fold(kv0, vv0, wave * KPW);
  for (int base = wave * KPW + STRIDE * UNROLL; base < n_cache; base += STRIDE * UNROLL) {
    u32x4 kv[UNROLL], vv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int key = min(base + slot + u * STRIDE, n_cache - 1);
      // streamed once per step by exactly one CU: non-temporal, so the K/V stream (up to 3.2 GB per
      // step) does not evict the decoder weights / activations the GEMMs re-read from L2 / MALL
      kv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kc + static_cast<size_t>(key) * D + sub * KPL));
      vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vc + static_cast<size_t>(key) * D + sub * KPL));
    }
    fold(kv, vv, base);
  }
  if constexpr (APPEND) {
    // the new key: one lane group of the block carries it (weight 0 everywhere else)
    const float s_new = key_group_sum<LPK>(chunk_dot<CT>(qc, new_k)) * kLog2e;
    const bool mine = tid < LPK;
    const float mn = mine ? fmaxf(m, s_new) : m;
    const float rs = __builtin_amdgcn_exp2f(m - mn);
    const float p = mine ? __builtin_amdgcn_exp2f(s_new - mn) : 0.f;
    float vf[KPL];
    unpack_chunk<CT>(new_v, vf);
#pragma unroll
    for (int j = 0; j < KPL; ++j) acc[j] = __builtin_fmaf(p, vf[j], acc[j] * rs);
    l = l * rs + p;
    m = mn;
  }
  // merge the key slots of this wave (lanes with equal `sub`)
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o), lo = __shfl_xor(l, o);
    const float mn = fmaxf(m, mo);
    const float ca = __builtin_amdgcn_exp2f(m - mn), cb = __builtin_amdgcn_exp2f(mo - mn);
    l = l * ca + lo * cb;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const float ao = __shfl_xor(acc[j], o);
      acc[j] = acc[j] * ca + ao * cb;
    }
    m = mn;
  }
  if (lane < LPK) {
    s_m[wave][lane] = m;
    s_l[wave][lane] = l;
#pragma unroll
    for (int j = 0; j < KPL; ++j) s_acc[wave][lane][j] = acc[j];
  }
  __syncthreads();
  if (tid < LPK) {
    float M = s_m[0][tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, s_m[w][tid]);
    float L = 0.f, o[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) o[j] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float c = __builtin_amdgcn_exp2f(s_m[w][tid] - M);
      L += s_l[w][tid] * c;
#pragma unroll
      for (int j = 0; j < KPL; ++j) o[j] += s_acc[w][tid][j] * c;
    }
    const float inv = 1.f / L;
    CT* dst = static_cast<CT*>(a.out) + static_cast<size_t>(b) * a.H * D + h * D + tid * KPL;
    if constexpr (KPL == 8) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = o[j] * inv;
      *reinterpret_cast<u32x4*>(dst) = pack_bf16x8(r);
    } else {
      *reinterpret_cast<u32x4*>(dst) = pack_f32x4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    }
  }
}

// ------------------------------------------------------------- decode attention, fp8 (OCP e4m3fn) K/V cache
// The decode step is HBM-bound on the K/V stream, so the cache is the one place where fp8 pays directly: rows are
// stored as 64 e4m3 bytes plus ONE power-of-two scale per (row, head, position) for K and for V (float2
// {k_scale, v_scale} in a side array [B][H][cap]): half the bytes of the bf16 cache (+6 % for the scales).
// A power-of-two scale only moves the exponent, so quantisation error is the pure 3-bit-mantissa rounding of e4m3.
// 4 lanes share a key (16 bytes = 16 elements each), a wave covers 16 consecutive keys per load (1 KiB
// contiguous); q, the probabilities and the accumulators are f32, activations in memory stay bf16.
// The new row of the step is quantised HERE (per-head amax over the quad), written to the cache, and attended in
// its DEQUANTISED form, so a position contributes the same values in the step that creates it and in every later one.
__device__ __forceinline__ void fp8x16_to_f32(const u32x4& c, float* out) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(c[w]), false);
    const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(static_cast<int>(c[w]), true);
    out[4 * w + 0] = lo[0];
    out[4 * w + 1] = lo[1];
    out[4 * w + 2] = hi[0];
    out[4 * w + 3] = hi[1];
  }
}
__device__ __forceinline__ u32x4 f32x16_to_fp8(const float* v) {
  u32x4 c;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int d = 0;
    d = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * w + 0], v[4 * w + 1], d, false);
    d = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * w + 2], v[4 * w + 3], d, true);
    c[w] = static_cast<unsigned>(d);
  }
  return c;
}
// power-of-two scale s with amax / s in [128, 256) (e4m3fn finite max = 448); amax = 0 -> 1
__device__ __forceinline__ float fp8_row_scale(float amax) {
  const unsigned e = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;      // biased exponent of amax
  const unsigned se = e > 8u ? e - 7u : 1u;                                    // 2^(exp - 7), clamped to normal f32
  return amax > 0.f ? __builtin_bit_cast(float, se << 23) : 1.f;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  return v;
}
// quantise 16 f32 values of a 64-wide row held by a quad (4 lanes x 16): returns the chunk, the row scale, and
// leaves the DEQUANTISED values in v
__device__ __forceinline__ u32x4 fp8_quantize_quad(float* v, float* scale_out) {
  float am = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) am = fmaxf(am, fabsf(v[j]));
  am = fminf(quad_max(am), 3.0e38f);                                           // Inf / NaN inputs: a finite scale
  const float sc = fp8_row_scale(am), inv = 1.f / sc;                          // exact: powers of two
  float t[16];
  // (clamped to e4m3fn's finite range: a non-finite activation must not leave NaN bytes in the cache for every
  // later step to read; finite rows are untouched, their |v / sc| is < 256)
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = fminf(fmaxf(v[j] * inv, -448.f), 448.f);
  const u32x4 c = f32x16_to_fp8(t);
  fp8x16_to_f32(c, t);
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = t[j] * sc;
  *scale_out = sc;
  return c;
}

template <bool APPEND, int NW, int UNROLL, int SPEC>
__global__ __launch_bounds__(NW * 64) void dec_attn_fp8_kernel(DecAttnArgs a) {
  constexpr int D = 64, EPL = 16;         // elements (= bytes) per lane
  constexpr int LPK = 4;                  // lanes sharing one key
  constexpr int KPW = 16;                 // keys per wave per load
  constexpr int STRIDE = NW * KPW;
  // UNROLL = keys per lane and iteration of the steady-state loop, SPEC = keys per lane requested before the row's
  // length is known (self: 3 waves x 16 keys x 2 = 96; round 3's group of 4 x 48 = 192 keys over-read a short cache by
  // half: traffic 1.49 x algorithmic at 129 keys).
  constexpr float kLog2e = 1.4426950408889634f;

  __shared__ float s_m[NW][LPK], s_l[NW][LPK], s_acc[NW][LPK][EPL];
  // The step's own (dequantised) K / V row waits here for the end of the key loop instead of in 32 VGPRs of every
  // lane: with UNROLL 3 that is 120 -> 92 VGPRs = five waves per SIMD, all six workgroups a CU gets at B x H = 1536
  // resident together.  Measured (profiles/r4_ab_fp8_attention_variants.txt): self-attention 23.3 -> 23.0 us at 6 heads,
  // 43.5 -> 42.2 at 12; the other combinations (4 keys per lane = four waves per SIMD again, 2, four waves per
  // workgroup; cross: three waves, 3 keys) are equal or slower.  The launch costs what a line through the bf16 and f32
  // kernels predicts for its bytes (6.2 us + bytes / 7.0 TB/s = 21.5 us): what keeps the e4m3 cache at 0.55-0.56 of the
  // HBM peak is that fixed part on half the bytes, not residency, the tail merge or the conversions.
  __shared__ float s_new[APPEND ? 2 : 1][LPK][EPL];

  MT3_PIN_DEC_ATTN_ARGS(a);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 3, slot = lane >> 2;
  int crow = b;                                        // row retirement: see dec_attn_kernel
  if (a.done) {
    if (a.done[b]) return;
    if (a.cache_row) crow = a.cache_row[b];
  }
  const size_t head = (static_cast<size_t>(crow) * a.H + h) * a.cap;
  const uint8_t* kc = static_cast<const uint8_t*>(a.kcache) + head * D;
  const uint8_t* vc = static_cast<const uint8_t*>(a.vcache) + head * D;
  const float2* sc2 = a.kv_scale + head;
  // first key group requested before the row's position counter is known (see dec_attn_kernel); discarded by
  // position below, so the contents of the cache and of its scale array past the row's length do not matter
  u32x4 kv0[SPEC], vv0[SPEC];
  float2 ss0[SPEC];
#pragma unroll
  for (int u = 0; u < SPEC; ++u) {
    const int key = min(wave * KPW + slot + u * STRIDE, a.cap - 1);
    kv0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kc + static_cast<size_t>(key) * D + sub * EPL));
    vv0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vc + static_cast<size_t>(key) * D + sub * EPL));
    ss0[u] = sc2[key];
  }
  const int n_keys = a.step ? (a.step[b] + 1) : a.n_keys;
  const int pos = n_keys - 1;
  const int n_cache = APPEND ? pos : n_keys;
  // discard by position what the speculative loads fetched from beyond the row's length (see dec_attn_kernel)
#pragma unroll
  for (int u = 0; u < SPEC; ++u)
    if (wave * KPW + slot + u * STRIDE >= n_cache) {
      vv0[u] = u32x4{0u, 0u, 0u, 0u};
      ss0[u] = make_float2(0.f, 0.f);
    }

  // q: 16 bf16 of this lane's slice -> f32, pre-multiplied by log2(e) (base-2 softmax)
  float q[EPL];
  float nk[EPL], nv[EPL];
  if (a.q_f32) {     // (wave-uniform) folded projections: unnormalised f32 query (+ new K/V row) + the row's partial sums
    const float* qp = a.q_f32 + static_cast<size_t>(b) * a.q_stride + h * D + sub * EPL;
#pragma unroll
    for (int j = 0; j < EPL; j += 4) {
      const float4 v = *reinterpret_cast<const float4*>(qp + j);
      q[j] = v.x, q[j + 1] = v.y, q[j + 2] = v.z, q[j + 3] = v.w;
    }
    if constexpr (APPEND) {   // raw loads BEFORE the 1/rms reduction (one memory round trip for all of them)
      const float* kp = static_cast<const float*>(a.new_k) + static_cast<size_t>(b) * a.kv_stride + h * D + sub * EPL;
      const float* vp = static_cast<const float*>(a.new_v) + static_cast<size_t>(b) * a.kv_stride + h * D + sub * EPL;
#pragma unroll
      for (int j = 0; j < EPL; j += 4) {
        const float4 kx = *reinterpret_cast<const float4*>(kp + j), vx = *reinterpret_cast<const float4*>(vp + j);
        nk[j] = kx.x, nk[j + 1] = kx.y, nk[j + 2] = kx.z, nk[j + 3] = kx.w;
        nv[j] = vx.x, nv[j + 1] = vx.y, nv[j + 2] = vx.z, nv[j + 3] = vx.w;
      }
    }
    const float rs1 = row_rs_from_partials(a.q_ss + static_cast<size_t>(b) * a.q_ss_n, a.q_ss_n, lane);
    const float rs = rs1 * kLog2e;
#pragma unroll
    for (int j = 0; j < EPL; ++j) q[j] *= rs;
    if constexpr (APPEND) {   // scale, round to bf16 as the GEMM epilogue would have
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        nk[j] = static_cast<float>(static_cast<__bf16>(nk[j] * rs1));
        nv[j] = static_cast<float>(static_cast<__bf16>(nv[j] * rs1));
      }
    }
  } else {
    const __bf16* qp = static_cast<const __bf16*>(a.q) + static_cast<size_t>(b) * a.q_stride + h * D + sub * EPL;
    const u32x4 q0 = *reinterpret_cast<const u32x4*>(qp), q1 = *reinterpret_cast<const u32x4*>(qp + 8);
    unpack_chunk<__bf16>(q0, q);
    unpack_chunk<__bf16>(q1, q + 8);
#pragma unroll
    for (int j = 0; j < EPL; ++j) q[j] *= kLog2e;
    if constexpr (APPEND) {
      const __bf16* kp = static_cast<const __bf16*>(a.new_k) + static_cast<size_t>(b) * a.kv_stride + h * D + sub * EPL;
      const __bf16* vp = static_cast<const __bf16*>(a.new_v) + static_cast<size_t>(b) * a.kv_stride + h * D + sub * EPL;
      unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(kp), nk);
      unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(kp + 8), nk + 8);
      unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(vp), nv);
      unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(vp + 8), nv + 8);
    }
  }
  if constexpr (APPEND) {
    float ks, vs;
    const u32x4 kq = fp8_quantize_quad(nk, &ks), vq = fp8_quantize_quad(nv, &vs);     // nk / nv now dequantised
    if (tid < LPK) {
      const size_t at = (head + pos) * D + sub * EPL;
      *reinterpret_cast<u32x4*>(static_cast<uint8_t*>(a.kcache) + at) = kq;
      *reinterpret_cast<u32x4*>(static_cast<uint8_t*>(a.vcache) + at) = vq;
      if (tid == 0) const_cast<float2*>(sc2)[pos] = make_float2(ks, vs);
      // (written and read back by the same four lanes: no barrier)
#pragma unroll
      for (int j = 0; j < EPL; j += 4) {
        *reinterpret_cast<float4*>(&s_new[0][sub][j]) = make_float4(nk[j], nk[j + 1], nk[j + 2], nk[j + 3]);
        *reinterpret_cast<float4*>(&s_new[APPEND ? 1 : 0][sub][j]) = make_float4(nv[j], nv[j + 1], nv[j + 2], nv[j + 3]);
      }
    }
  }

  float m = -1.0e30f, l = 0.f, acc[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) acc[j] = 0.f;

  auto fold = [&](const auto& kv, const auto& vv, const auto& ss, int base) {
    constexpr int G = static_cast<int>(sizeof(ss) / sizeof(float2));            // keys per lane in this group
    float sc[G];
    float mn = m;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      float kf[EPL];
      fp8x16_to_f32(kv[u], kf);
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; j += 2) {
        d0 = __builtin_fmaf(q[j], kf[j], d0);
        d1 = __builtin_fmaf(q[j + 1], kf[j + 1], d1);
      }
      float d = d0 + d1;
      d = dpp_add<0xB1>(d);
      d = dpp_add<0x4E>(d);
      sc[u] = d * ss[u].x;
      if (base + slot + u * STRIDE < n_cache) mn = fmaxf(mn, sc[u]);
    }
    const float rs = __builtin_amdgcn_exp2f(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] *= rs;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const float p = base + slot + u * STRIDE < n_cache ? __builtin_amdgcn_exp2f(sc[u] - mn) : 0.f;
      psum += p;
      const float pv = p * ss[u].y;
      float vf[EPL];
      fp8x16_to_f32(vv[u], vf);
#pragma unroll
      for (int j = 0; j < EPL; ++j) acc[j] = __builtin_fmaf(pv, vf[j], acc[j]);
    }
    l = l * rs + psum;
    m = mn;
  };
  if (wave * KPW < n_cache) fold(kv0, vv0, ss0, wave * KPW);
  for (int base = wave * KPW + STRIDE * SPEC; base < n_cache; base += STRIDE * UNROLL) {
    u32x4 kv[UNROLL], vv[UNROLL];
    float2 ss[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int key = min(base + slot + u * STRIDE, n_cache - 1);
      kv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kc + static_cast<size_t>(key) * D + sub * EPL));
      vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vc + static_cast<size_t>(key) * D + sub * EPL));
      ss[u] = sc2[key];
    }
    fold(kv, vv, ss, base);
  }
  if constexpr (APPEND) {
    if (tid < LPK) {                                       // the quad that wrote the new row attends it
      float nkq[EPL], nvq[EPL];
#pragma unroll
      for (int j = 0; j < EPL; j += 4) {
        const float4 kx = *reinterpret_cast<const float4*>(&s_new[0][sub][j]);
        const float4 vx = *reinterpret_cast<const float4*>(&s_new[1][sub][j]);
        nkq[j] = kx.x, nkq[j + 1] = kx.y, nkq[j + 2] = kx.z, nkq[j + 3] = kx.w;
        nvq[j] = vx.x, nvq[j + 1] = vx.y, nvq[j + 2] = vx.z, nvq[j + 3] = vx.w;
      }
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < EPL; ++j) d = __builtin_fmaf(q[j], nkq[j], d);
      d = dpp_add<0xB1>(d);
      d = dpp_add<0x4E>(d);
      const float mn = fmaxf(m, d);
      const float rs = __builtin_amdgcn_exp2f(m - mn);
      const float p = __builtin_amdgcn_exp2f(d - mn);
#pragma unroll
      for (int j = 0; j < EPL; ++j) acc[j] = __builtin_fmaf(p, nvq[j], acc[j] * rs);
      l = l * rs + p;
      m = mn;
    }
  }
  // merge the 16 key slots of the wave (lanes of equal `sub`): inside a 16-lane row the partner arrives over the DPP
  // network (row_ror 4, then 8: every lane ends up with its row's total), across rows through ds_bpermute
  auto merge = [&](auto fetch) {
    const float mo = fetch(m), lo = fetch(l);
    const float mn = fmaxf(m, mo);
    const float ca = __builtin_amdgcn_exp2f(m - mn), cb = __builtin_amdgcn_exp2f(mo - mn);
    l = l * ca + lo * cb;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = acc[j] * ca + fetch(acc[j]) * cb;
    m = mn;
  };
  merge([](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true)); });
  merge([](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); });
  merge([](float v) { return __shfl_xor(v, 16); });
  merge([](float v) { return __shfl_xor(v, 32); });
  if (lane < LPK) {
    s_m[wave][lane] = m;
    s_l[wave][lane] = l;
#pragma unroll
    for (int j = 0; j < EPL; ++j) s_acc[wave][lane][j] = acc[j];
  }
  __syncthreads();
  if (tid < LPK) {
    float M = s_m[0][tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, s_m[w][tid]);
    float Lsum = 0.f, o[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) o[j] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float c = __builtin_amdgcn_exp2f(s_m[w][tid] - M);
      Lsum += s_l[w][tid] * c;
#pragma unroll
      for (int j = 0; j < EPL; ++j) o[j] += s_acc[w][tid][j] * c;
    }
    const float inv = 1.f / Lsum;
    __bf16* dst = static_cast<__bf16*>(a.out) + static_cast<size_t>(b) * a.H * D + h * D + tid * EPL;
    float r0[8], r1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r0[j] = o[j] * inv;
      r1[j] = o[8 + j] * inv;
    }
    *reinterpret_cast<u32x4*>(dst) = pack_bf16x8(r0);
    *reinterpret_cast<u32x4*>(dst + 8) = pack_bf16x8(r1);
  }
}

// bf16 rows [rows][64] (K rows then V rows, `rows` each, e.g. the cross-attention [2][B][H][T][64] of one layer)
// -> e4m3 rows + float2 {k_scale, v_scale} per row; one quad per (K row, V row) pair
__global__ __launch_bounds__(256) void kv_quantize_fp8_kernel(const __bf16* __restrict__ src, uint8_t* __restrict__ dst,
                                                               float2* __restrict__ scales, int rows) {
  const int r = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  if (r >= rows) return;
  float k[16], v[16];
  const __bf16* kp = src + static_cast<size_t>(r) * 64 + sub * 16;
  const __bf16* vp = kp + static_cast<size_t>(rows) * 64;
  unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(kp), k);
  unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(kp + 8), k + 8);
  unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(vp), v);
  unpack_chunk<__bf16>(*reinterpret_cast<const u32x4*>(vp + 8), v + 8);
  float ks, vs;
  const u32x4 kq = fp8_quantize_quad(k, &ks), vq = fp8_quantize_quad(v, &vs);
  *reinterpret_cast<u32x4*>(dst + static_cast<size_t>(r) * 64 + sub * 16) = kq;
  *reinterpret_cast<u32x4*>(dst + (static_cast<size_t>(rows) + r) * 64 + sub * 16) = vq;
  if (sub == 0) scales[r] = make_float2(ks, vs);
}

int launch_kv_quantize_fp8(const void* src_bf16, void* dst_fp8, void* scales, int rows, hipStream_t s) {
  if (!src_bf16 || !dst_fp8 || !scales || rows <= 0) return mt3::fail(MT3_ERR_INVALID, "kv_quantize_fp8: bad arguments");
  hipLaunchKernelGGL(kv_quantize_fp8_kernel, dim3((rows + 63) / 64), dim3(256), 0, s,
                     static_cast<const __bf16*>(src_bf16), static_cast<uint8_t*>(dst_fp8),
                     static_cast<float2*>(scales), rows);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

int launch_decode_attention(int dtype, const DecAttnArgs& a, hipStream_t s) {
  if ((!a.q && !a.q_f32) || !a.kcache || !a.vcache || !a.out || a.B <= 0 || a.H <= 0 || a.cap <= 0)
    return mt3::fail(MT3_ERR_INVALID, "decode_attention: bad arguments");
  if (a.q_f32 && (!a.q_ss || a.q_ss_n <= 0 || a.q_ss_n > 64 || (a.q_ss_n & 3)))
    return mt3::fail(MT3_ERR_INVALID, "decode_attention: the unnormalised f32 query form needs the row's partial sums "
                                      "of squares (4 .. 64 of them, a multiple of 4)");
  if (!a.step && (a.n_keys <= 0 || a.n_keys > a.cap)) return mt3::fail(MT3_ERR_INVALID, "decode_attention: n_keys");
  const bool append = a.new_k != nullptr;
  if (append && !a.new_v) return mt3::fail(MT3_ERR_INVALID, "decode_attention: new_k without new_v");
  // waves per (batch, head) workgroup.  B*H workgroups must all be resident for an even HBM stream:
  // at 84 VGPRs a CU holds 20 waves, so 3 waves per workgroup keeps B*H = 1536 groups (4608 waves)
  // co-resident on 256 CUs, while 4 would leave a 256-group second round running at 1/5 occupancy
  // (2 / 3 / 4 waves measured within 2 % of each other; r1).
  if (a.kv_scale) {
    // measured on MI355X at B = 256 (r2): the growing self-attention cache streams best with 3 waves (22.8 us against
    // 23.7 / 24.0 with 2 / 4 at the mean depth); the fixed 256-key cross-attention with 4 (one whole 64-key x 4 pass
    // per wave: 12.4 us against 13.4 with 3)
    // fp8 (e4m3) K/V cache; activations (q, new rows, out) are bf16
    if (dtype != MT3_BF16) return mt3::fail(MT3_ERR_INVALID, "decode_attention: the fp8 K/V cache needs bf16 activations");
    // (round 4, profiles/r4_ab_fp8_attention_variants.txt: keys per lane in the loop / in the speculative group, waves)
    if (append) hipLaunchKernelGGL((dec_attn_fp8_kernel<true, 3, 3, 2>), dim3(a.B * a.H), dim3(192), 0, s, a);
    else hipLaunchKernelGGL((dec_attn_fp8_kernel<false, 4, 2, 2>), dim3(a.B * a.H), dim3(256), 0, s, a);
    MT3_HIP_CHECK(hipGetLastError());
    return MT3_OK;
  }
  const dim3 grid(a.B * a.H), block(3 * 64);
#define MT3_LAUNCH_DEC(CT, AP) hipLaunchKernelGGL((dec_attn_kernel<CT, AP, 3>), grid, block, 0, s, a)
#define MT3_LAUNCH_DEC_Q(CT, AP) hipLaunchKernelGGL((dec_attn_kernel<CT, AP, 3, true>), grid, block, 0, s, a)
  if (dtype == MT3_BF16 && a.q_f32) {
    if (append) MT3_LAUNCH_DEC_Q(__bf16, true);
    else MT3_LAUNCH_DEC_Q(__bf16, false);
  } else if (dtype == MT3_BF16) {
    if (append) MT3_LAUNCH_DEC(__bf16, true);
    else MT3_LAUNCH_DEC(__bf16, false);
  } else if (dtype == MT3_F32 && a.q_f32) {
    if (append) MT3_LAUNCH_DEC_Q(float, true);
    else MT3_LAUNCH_DEC_Q(float, false);
  } else if (dtype == MT3_F32) {
    if (append) MT3_LAUNCH_DEC(float, true);
    else MT3_LAUNCH_DEC(float, false);
  } else {
    return mt3::fail(MT3_ERR_INVALID, "decode_attention: unknown dtype");
  }
#undef MT3_LAUNCH_DEC
#undef MT3_LAUNCH_DEC_Q
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

}  // namespace mt3k

extern "C" {

int mt3_op_encoder_attention(int32_t dtype, const void* d_qkv, void* d_out, int32_t B, int32_t T, int32_t H,
                             void* stream) {
  return mt3k::launch_encoder_attention(dtype, d_qkv, d_out, B, T, H, static_cast<hipStream_t>(stream));
}

int mt3_op_decode_attention(int32_t dtype, const void* d_q, int32_t q_stride, void* d_kcache, void* d_vcache,
                            int32_t cap, const void* d_new_k, const void* d_new_v, int32_t kv_stride,
                            const int32_t* d_step, int32_t n_keys, void* d_out, int32_t B, int32_t H, void* stream) {
  mt3k::DecAttnArgs a{};
  a.q = d_q;
  a.q_stride = q_stride;
  a.kcache = d_kcache;
  a.vcache = d_vcache;
  a.cap = cap;
  a.new_k = d_new_k;
  a.new_v = d_new_v;
  a.kv_stride = kv_stride;
  a.step = d_step;
  a.n_keys = n_keys;
  a.out = d_out;
  a.B = B;
  a.H = H;
  return mt3k::launch_decode_attention(dtype, a, static_cast<hipStream_t>(stream));
}

int mt3_op_decode_attention_fp8(const void* d_q, int32_t q_stride, void* d_kcache, void* d_vcache, void* d_kv_scale,
                                int32_t cap, const void* d_new_k, const void* d_new_v, int32_t kv_stride,
                                const int32_t* d_step, int32_t n_keys, void* d_out, int32_t B, int32_t H,
                                void* stream) {
  if (!d_kv_scale) return mt3::fail(MT3_ERR_INVALID, "decode_attention_fp8: null scale array");
  mt3k::DecAttnArgs a{};
  a.q = d_q;
  a.q_stride = q_stride;
  a.kcache = d_kcache;
  a.vcache = d_vcache;
  a.kv_scale = static_cast<float2*>(d_kv_scale);
  a.cap = cap;
  a.new_k = d_new_k;
  a.new_v = d_new_v;
  a.kv_stride = kv_stride;
  a.step = d_step;
  a.n_keys = n_keys;
  a.out = d_out;
  a.B = B;
  a.H = H;
  return mt3k::launch_decode_attention(MT3_BF16, a, static_cast<hipStream_t>(stream));
}

int mt3_op_kv_quantize_fp8(const void* d_src, void* d_dst, void* d_scales, int32_t rows, void* stream) {
  return mt3k::launch_kv_quantize_fp8(d_src, d_dst, d_scales, rows, static_cast<hipStream_t>(stream));
}

}  // extern "C"

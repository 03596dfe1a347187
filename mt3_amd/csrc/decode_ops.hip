// Small bandwidth/latency kernels around the decode loop (gfx950).
//
//   rmsnorm_kernel      T5 LayerNorm (mt3/layers.py:604-621) as a standalone pass -- used once per
//                       encode for `encoder_norm` (network.py:192); every other norm is fused into
//                       the GEMM that consumes it (gemm.hip NORM).
//   embed_kernel        Embed (one-hot matmul in the reference, layers.py:530-533 -> a row gather)
//                       + FixedEmbed decode slice pos[t] (layers.py:589-596).
//   argmax_step_kernel  greedy pick of the step (lowest id on ties), EOS bookkeeping, writes
//                       ids[b][t]; rows that already emitted EOS get 0 (pad).
//                       Positions are PER-ROW device counters (no cross-row sync), so ONE captured
//                       hipGraph serves every step; the block also writes the next step's embedding row.
//   ids_to_tokens_kernel GenericTokenVocabulary._decode_tf (mt3/vocabularies.py:241-271), bit-exact.
#include <hip/hip_runtime.h>

#include "common.h"
#include "device.h"
#include "kernels.h"

namespace mt3k {

template <typename CT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       CT* __restrict__ out_ct, float* __restrict__ out_f32,
                                                       int rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per row
  if (row >= rows) return;
  const float* xr = x + static_cast<size_t>(row) * dim;
  float ss = 0.f;
  for (int i = lane * 4; i < dim; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rs = rsqrtf(ss / static_cast<float>(dim) + 1e-6f);
  for (int i = lane * 4; i < dim; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    const float4 sc = *reinterpret_cast<const float4*>(scale + i);
    const float y0 = v.x * rs * sc.x, y1 = v.y * rs * sc.y, y2 = v.z * rs * sc.z, y3 = v.w * rs * sc.w;
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + static_cast<size_t>(row) * dim + i) = make_float4(y0, y1, y2, y3);
    if (out_ct) {
      CT* o = out_ct + static_cast<size_t>(row) * dim + i;
      o[0] = to_ct<CT>(y0);
      o[1] = to_ct<CT>(y1);
      o[2] = to_ct<CT>(y2);
      o[3] = to_ct<CT>(y3);
    }
  }
}

int launch_rmsnorm(int dtype, const float* x, const float* scale, void* out_ct, float* out_f32, int rows, int dim,
                   hipStream_t s) {
  if (!x || !scale || rows <= 0 || dim % 4 != 0) return mt3::fail(MT3_ERR_INVALID, "rmsnorm: bad arguments");
  const dim3 grid((rows + 3) / 4), block(256);
  if (dtype == MT3_BF16)
    hipLaunchKernelGGL((rmsnorm_kernel<__bf16>), grid, block, 0, s, x, scale, static_cast<__bf16*>(out_ct), out_f32,
                       rows, dim);
  else
    hipLaunchKernelGGL((rmsnorm_kernel<float>), grid, block, 0, s, x, scale, static_cast<float*>(out_ct), out_f32,
                       rows, dim);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

__global__ __launch_bounds__(128) void embed_kernel(const float* __restrict__ table, const float* __restrict__ pos,
                                                     const int* __restrict__ tok, const int* __restrict__ step,
                                                     float* __restrict__ y, int dim) {
  const int b = blockIdx.x;
  const float* e = table + static_cast<size_t>(tok[b]) * dim;
  const float* p = pos + static_cast<size_t>(step[b]) * dim;
  for (int i = threadIdx.x * 4; i < dim; i += 512) {
    const float4 a = *reinterpret_cast<const float4*>(e + i), c = *reinterpret_cast<const float4*>(p + i);
    *reinterpret_cast<float4*>(y + static_cast<size_t>(b) * dim + i) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  }
}

int launch_embed(const float* table, const float* pos, const int* tok, const int* step, float* y, int B, int dim,
                 hipStream_t s) {
  hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(128), 0, s, table, pos, tok, step, y, dim);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

__global__ __launch_bounds__(256) void argmax_step_kernel(const float* __restrict__ logits, int vocab,
                                                           int* __restrict__ ids, int ids_stride,
                                                           int* __restrict__ cur_tok, int* __restrict__ done,
                                                           int* __restrict__ n_done, int* __restrict__ step,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ pos_table, int max_pos,
                                                           float* __restrict__ y_next, int dim) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ int s_tok, s_t;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + static_cast<size_t>(b) * vocab;
  float best = -3.0e38f;
  int bi = 0x7fffffff;
  for (int i = tid; i < vocab; i += 256) {
    const float v = row[i];
    if (v > best) {            // ascending i: strict > keeps the lowest index on ties
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    s_v[wave] = best;
    s_i[wave] = bi;
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) {
        best = s_v[w];
        bi = s_i[w];
      }
    const int was_done = done[b];
    const int tok = was_done ? 0 : bi;
    const int t = step[b];                // this row's own position counter: no cross-row synchronisation
    ids[static_cast<size_t>(b) * ids_stride + t] = tok;
    cur_tok[b] = tok;
    step[b] = t + 1;
    if (!was_done && tok == 1) {          // EOS
      done[b] = 1;
      atomicAdd(n_done, 1);
    }
    s_tok = tok;
    s_t = t + 1;
  }
  __syncthreads();
  // the next step's decoder input row: Embed(tok) + FixedEmbed[t+1]  (saves the embed launch of every step)
  if (y_next) {
    const int tp = s_t < max_pos ? s_t : max_pos - 1;
    const float* e = table + static_cast<size_t>(s_tok) * dim;
    const float* p = pos_table + static_cast<size_t>(tp) * dim;
    for (int i = tid * 4; i < dim; i += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(e + i), c = *reinterpret_cast<const float4*>(p + i);
      *reinterpret_cast<float4*>(y_next + static_cast<size_t>(b) * dim + i) =
          make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
  }
}

int launch_argmax_step(const float* logits, int vocab, int* ids, int ids_stride, int* cur_tok, int* done,
                       int* n_done, int* step, const float* table, const float* pos_table, int max_pos,
                       float* y_next, int dim, int B, hipStream_t s) {
  hipLaunchKernelGGL(argmax_step_kernel, dim3(B), dim3(256), 0, s, logits, vocab, ids, ids_stride, cur_tok, done,
                     n_done, step, table, pos_table, max_pos, y_next, dim);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

__global__ __launch_bounds__(256) void ids_to_tokens_kernel(const int* __restrict__ ids, int L, int num_regular,
                                                             int* __restrict__ out) {
  __shared__ int s_first[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* row = ids + static_cast<size_t>(b) * L;
  int first = L;                                      // index of the first EOS (id == 1)
  for (int i = tid; i < L; i += 256)
    if (row[i] == 1) {
      first = i;
      break;                                          // ascending i per thread
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
  if (lane == 0) s_first[wave] = first;
  __syncthreads();
  first = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
  for (int i = tid; i < L; i += 256) {
    const int id = row[i];
    int t;
    if (i >= first) t = -1;                           // DECODED_EOS_ID from the first EOS onward
    else if (id >= 3 && id < 3 + num_regular) t = id - 3;
    else t = -2;                                      // DECODED_INVALID_ID
    out[static_cast<size_t>(b) * L + i] = t;
  }
}

int launch_ids_to_tokens(const int* ids, int B, int L, int num_regular, int* out, hipStream_t s) {
  if (!ids || !out || B <= 0 || L <= 0) return mt3::fail(MT3_ERR_INVALID, "ids_to_tokens: bad arguments");
  hipLaunchKernelGGL(ids_to_tokens_kernel, dim3(B), dim3(256), 0, s, ids, L, num_regular, out);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

}  // namespace mt3k

extern "C" int mt3_ids_to_tokens(const int32_t* d_ids, int32_t batch, int32_t length, int32_t num_regular,
                                 int32_t* d_tokens, void* stream) {
  return mt3k::launch_ids_to_tokens(d_ids, batch, length, num_regular, d_tokens, static_cast<hipStream_t>(stream));
}

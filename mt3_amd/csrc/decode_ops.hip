// Small bandwidth/latency kernels around the decode loop (gfx950).
//
//   rmsnorm_kernel      T5 LayerNorm (mt3/layers.py:604-621) as a standalone pass -- used once per
//                       encode for `encoder_norm` (network.py:192); every other norm is fused into
//                       the GEMM that consumes it (gemm.hip NORM).
//   embed_kernel        Embed (one-hot matmul in the reference, layers.py:530-533 -> a row gather)
//                       + FixedEmbed decode slice pos[t] (layers.py:589-596).
//   argmax_step_kernel  greedy pick of the step (lowest id on ties), EOS bookkeeping, writes
//                       ids[b][t]; rows that already emitted EOS get 0 (pad).  BEAM1 variant: one step
//                       of t5x beam_search with num_decodes=1 (top-2 of log_softmax, live/finished sets).
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

// one float4 of a decoder-input row: the f32 value, for the split residual form the exact sum of squares of its
// 16-column group (4 consecutive lanes = one quad own one group), and for the bf16 path also its compute-type copy
// (y_ct == nullptr with y_ss != nullptr: the f32 engine, whose compute-type rows ARE the f32 rows)
__device__ __forceinline__ void put_row_piece(float4 v, float* y, void* y_ct, float* y_ss, size_t row, int dim, int i) {
  *reinterpret_cast<float4*>(y + row * dim + i) = v;
  if (y_ss) {
    float t = __builtin_fmaf(v.w, v.w, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
    t = quad_sum(t);
    if ((threadIdx.x & 3) == 0) y_ss[row * (dim >> 4) + (i >> 4)] = t;
  }
  if (y_ct) {
    uint2 pk;
    pk.x = pack_bf16x2(v.x, v.y);
    pk.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(static_cast<__bf16*>(y_ct) + row * dim + i) = pk;
  }
}

// the split form of residual rows that did not come out of a RESID epilogue (the encoder's input projection):
// f32 rows -> bf16 copy + exact per-16-column sums of squares; one float4 per thread, a quad per 16-column group
__global__ __launch_bounds__(256) void residual_split_kernel(const float* __restrict__ x, void* __restrict__ x_ct,
                                                              float* __restrict__ x_ss, size_t n4, int dim) {
  const size_t i4 = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  const bool in = i4 < n4;
  const float4 v = in ? reinterpret_cast<const float4*>(x)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
  float t = __builtin_fmaf(v.w, v.w, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
  t = quad_sum(t);
  if (!in) return;
  if ((threadIdx.x & 3) == 0) x_ss[i4 >> 2] = t;         // [row][dim/16] == flat index / 16
  uint2 pk;
  pk.x = pack_bf16x2(v.x, v.y);
  pk.y = pack_bf16x2(v.z, v.w);
  reinterpret_cast<uint2*>(x_ct)[i4] = pk;
}

int launch_residual_split(const float* x, void* x_ct, float* x_ss, int rows, int dim, hipStream_t s) {
  if (!x || !x_ct || !x_ss || rows <= 0 || dim % 16) return mt3::fail(MT3_ERR_INVALID, "residual_split: bad arguments");
  const size_t n4 = static_cast<size_t>(rows) * dim / 4;
  hipLaunchKernelGGL(residual_split_kernel, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, s, x, x_ct,
                     x_ss, n4, dim);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// q_out[b] = ew[tok] + pw[t]: the first decoder layer's unnormalised q | k | v | cross-q of the row (RowProj)
__device__ __forceinline__ void put_row_projection(const RowProj& rp, int b, int tok, int t, int tid, int nthreads) {
  const float* e = rp.ew + static_cast<size_t>(tok) * rp.q_n;
  const float* p = rp.pw + static_cast<size_t>(t) * rp.q_n;
  float* o = rp.q_out + static_cast<size_t>(b) * rp.q_n;
  for (int i = tid * 4; i < rp.q_n; i += nthreads * 4) {
    const float4 a = *reinterpret_cast<const float4*>(e + i), c = *reinterpret_cast<const float4*>(p + i);
    *reinterpret_cast<float4*>(o + i) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  }
}

__global__ __launch_bounds__(128) void embed_kernel(const float* __restrict__ table, const float* __restrict__ pos,
                                                     const int* __restrict__ tok, const int* __restrict__ step,
                                                     float* __restrict__ y, void* __restrict__ y_ct,
                                                     float* __restrict__ y_ss, int dim, RowProj rp) {
  const int b = blockIdx.x;
  const float* e = table + static_cast<size_t>(tok[b]) * dim;
  const float* p = pos + static_cast<size_t>(step[b]) * dim;
  for (int i = threadIdx.x * 4; i < dim; i += 512) {
    const float4 a = *reinterpret_cast<const float4*>(e + i), c = *reinterpret_cast<const float4*>(p + i);
    put_row_piece(make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w), y, y_ct, y_ss, b, dim, i);
  }
  if (rp.q_out) put_row_projection(rp, b, tok[b], step[b], threadIdx.x, 128);
}

int launch_embed(const float* table, const float* pos, const int* tok, const int* step, float* y, void* y_ct,
                 float* y_ss, int B, int dim, const RowProj& rp, hipStream_t s) {
  if ((y_ct || y_ss) && (!y_ss || dim % 16)) return mt3::fail(MT3_ERR_INVALID, "embed: the split form needs y_ss and dim % 16 == 0");
  if (rp.q_out && (!rp.ew || !rp.pw || rp.q_n % 4)) return mt3::fail(MT3_ERR_INVALID, "embed: row projection tables missing");
  hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(128), 0, s, table, pos, tok, step, y, y_ct, y_ss, dim, rp);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// order of candidates: larger logit first, lower id on ties (lax.top_k / argmax convention)
__device__ inline bool cand_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

struct Top2 {
  float v1, v2;
  int i1, i2;
  __device__ inline void insert(float v, int i) {
    if (cand_better(v, i, v1, i1)) {
      v2 = v1;
      i2 = i1;
      v1 = v;
      i1 = i;
    } else if (cand_better(v, i, v2, i2)) {
      v2 = v;
      i2 = i;
    }
  }
};

// BEAM1 = false: greedy pick (the product default).
// BEAM1 = true : one step of t5x `beam_search` with num_decodes = 1: the two best
//   candidates of log_softmax are taken; the LIVE hypothesis follows the best non-EOS one; an EOS candidate
//   finishes `live prefix + EOS` with score (live_logp + logp_eos) / brevity_penalty(t + 1), kept if it
//   beats the best finished score of the row; the row is done once that score exceeds
//   live_logp / brevity_penalty(max_len + 1), after which no later finish can win.  beam_f holds
//   [live_logp | best_score] per row, beam_len the prefix length of the best finished hypothesis (-1: none);
//   beam_cfg[0] = brevity_penalty(max_len + 1) of this call, beam_cfg[1 + n] = brevity_penalty(n) =
//   ((5 + n) / 6) ^ alpha, tabulated on the host.
template <bool BEAM1>
__global__ __launch_bounds__(256) void argmax_step_kernel(float* __restrict__ logits, int vocab,
                                                           int* __restrict__ ids, int ids_stride,
                                                           int* __restrict__ cur_tok, int* __restrict__ done,
                                                           int* __restrict__ n_done, int* __restrict__ step,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ pos_table, int max_pos,
                                                           float* __restrict__ y_next, void* __restrict__ y_ct,
                                                           float* __restrict__ y_ss, int dim,
                                                           float* __restrict__ beam_f, int* __restrict__ beam_len,
                                                           int* __restrict__ beam_len_row,
                                                           const float* __restrict__ beam_cfg, int beam_rows,
                                                           const int* __restrict__ forced, int forced_stride,
                                                           RowProj rp, LogitScale ls, StepRetire rt) {
  __shared__ float s_v[8], s_sum[4];
  __shared__ int s_i[8];
  __shared__ int s_tok, s_t;
  __shared__ float s_rs;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // row retirement: a finished slot is not touched again (its ids stay at the 0 they were initialised to, its
  // position counter stops: nothing reads it any more)
  if (rt.retire && done[b]) return;
  const int out_row = rt.slot_row ? rt.slot_row[b] : b;        // row of `ids` / `eos_at` this slot decodes
  float* row = logits + static_cast<size_t>(b) * vocab;
  // Round 6: the whole row in registers (vocab <= 2048: eight values per thread), requested in ONE batch before anything
  // that has to be waited for.  The loops this replaces walked the row three times, one element per thread and trip, each
  // trip a dependent load (the compiler waited for every load before the next): a dozen memory round trips in a kernel
  // that every decode step ends with (rocprofv3: 14.3 us per group step).  Same values, same insertion order, same bits.
  constexpr int kPer = 8;
  const bool in_regs = vocab <= 256 * kPer;
  float xv[kPer];
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int i = tid + u * 256;
      xv[u] = row[i < vocab ? i : vocab - 1];
    }
  }
  // thread 0 issues its state loads up front so that their latency hides behind the reductions
  int was_done = 0, t = 0, blen = -1, eos_len = 0x7fffffff;
  float live = 0.f, best = 0.f, bp_max = 1.f, bp_t = 1.f;
  if (tid == 0) {
    was_done = done[b];
    t = step[b];
    if (rt.eos_at) eos_len = rt.eos_at[rt.slot_seg ? rt.slot_seg[b] : out_row];
    if (BEAM1) {
      live = beam_f[b];
      best = beam_f[beam_rows + b];
      blen = beam_len[b];
      bp_max = beam_cfg[0];
      bp_t = beam_cfg[1 + t + 1];
    }
  }
  if (ls.ss) {
    // folded logits projection: the row arrives unnormalised; 1/rms of the final residual row from its partial sums
    // (<= 64 of them: one wave), then the scaled logits replace the raw ones (callers read them: first-step /
    // per-step logits) -- arg-max would not need the scale, the beam's log-softmax and the parity outputs do
    if (wave == 0) {
      float p = lane < ls.n_ss ? ls.ss[static_cast<size_t>(b) * ls.n_ss + lane] : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
      if (lane == 0) s_rs = rsqrtf(p / static_cast<float>(ls.dim) + 1e-6f);
    }
    __syncthreads();
    const float rs = s_rs;
    if (in_regs) {
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int i = tid + u * 256;
        xv[u] *= rs;
        if (i < vocab) row[i] = xv[u];
      }
    } else {
      for (int i = tid; i < vocab; i += 256) row[i] *= rs;
      // (each thread re-reads below exactly the elements it just wrote: no barrier needed)
    }
  }
  Top2 t2{-3.0e38f, -3.0e38f, 0x7fffffff, 0x7fffffff};
  float acc = 0.f;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) {                               // ascending i per thread
      const int i = tid + u * 256;
      if (i < vocab) t2.insert(xv[u], i);
    }
    if (BEAM1) {
      // sum of exp(x - thread max) over this thread's elements; rescaled to the wave max below
#pragma unroll
      for (int u = 0; u < kPer; ++u)
        if (tid + u * 256 < vocab) acc += __expf(xv[u] - t2.v1);
    }
  } else {
    for (int i = tid; i < vocab; i += 256) t2.insert(row[i], i);   // ascending i per thread
    if (BEAM1) {
      for (int i = tid; i < vocab; i += 256) acc += __expf(row[i] - t2.v1);
    }
  }
  const float own_max = t2.v1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov1 = __shfl_xor(t2.v1, o), ov2 = __shfl_xor(t2.v2, o);
    const int oi1 = __shfl_xor(t2.i1, o), oi2 = __shfl_xor(t2.i2, o);
    t2.insert(ov1, oi1);
    if (BEAM1) t2.insert(ov2, oi2);
  }
  if (BEAM1) {
    acc *= __expf(own_max - t2.v1);                     // every lane now holds the wave max in t2.v1
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  }
  if (lane == 0) {
    s_v[wave] = t2.v1;
    s_i[wave] = t2.i1;
    s_v[4 + wave] = t2.v2;
    s_i[4 + wave] = t2.i2;
    s_sum[wave] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    float m = t2.v1;                                    // wave 0's max; the block max after the merge
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      t2.insert(s_v[w], s_i[w]);
      if (BEAM1) t2.insert(s_v[4 + w], s_i[4 + w]);
    }
    int tok;
    bool finished = false;                // this step finishes the slot
    if (!BEAM1) {
      if (forced) was_done = 0;           // teacher forcing: every step reports its own arg-max, no EOS bookkeeping
      tok = was_done ? 0 : (t + 1 >= eos_len ? 1 : t2.i1);      // synthetic EOS schedule: a point mass on EOS
      finished = !was_done && tok == 1 && !forced;   // EOS
    } else {
      tok = 0;
      if (!was_done) {
        m = t2.v1;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += s_sum[w] * __expf(s_v[w] - m);
        const float lse = m + __logf(sum);
        float lp1 = t2.v1 - lse, lp2 = t2.v2 - lse;
        int eos_slot = t2.i1 == 1 ? 1 : (t2.i2 == 1 ? 2 : 0);
        if (t + 1 >= eos_len) {           // synthetic EOS schedule: EOS with probability 1, everything else impossible
          eos_slot = 1;
          lp1 = 0.f;
          lp2 = -1.0e30f;
        }
        if (eos_slot) {
          const float score = (live + (eos_slot == 1 ? lp1 : lp2)) / bp_t;
          if (blen < 0 || score > best) {
            best = score;
            blen = t;
          }
        }
        tok = eos_slot == 1 ? t2.i2 : t2.i1;
        live += eos_slot == 1 ? lp2 : lp1;
        beam_f[b] = live;
        beam_f[beam_rows + b] = best;
        beam_len[b] = blen;
        beam_len_row[out_row] = blen;
        finished = blen >= 0 && best > live / bp_max;
      }
    }
    // in-flight batching: a slot that has written max_len ids is finished as well (the row ran out of positions)
    if (!was_done && rt.max_len > 0 && t + 1 >= rt.max_len) finished = true;
    if (finished) {
      done[b] = 1;
      atomicAdd(n_done, 1);
    }
    ids[static_cast<size_t>(out_row) * ids_stride + t] = tok;
    // teacher forcing (Transformer.decode on given decoder_input_tokens, network.py:303-361): the NEXT input is
    // the caller's token for position t + 1, whatever this step predicted
    if (!BEAM1 && forced) tok = forced[static_cast<size_t>(b) * forced_stride + t];
    cur_tok[b] = tok;
    step[b] = t + 1;
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
      put_row_piece(make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w), y_next, y_ct, y_ss, b, dim, i);
    }
    if (rp.q_out) put_row_projection(rp, b, s_tok, tp, tid, 256);
  }
}

int launch_argmax_step(float* logits, int vocab, int* ids, int ids_stride, int* cur_tok, int* done,
                       int* n_done, int* step, const float* table, const float* pos_table, int max_pos,
                       float* y_next, void* y_ct, float* y_ss, int dim, int B, const BeamState* beam,
                       const int* forced, int forced_stride, const RowProj& rp, const LogitScale& ls,
                       const StepRetire& rt, hipStream_t s) {
  if (beam && forced) return mt3::fail(MT3_ERR_INVALID, "argmax_step: teacher forcing is a greedy-path feature");
  if (beam && !beam->len_row) return mt3::fail(MT3_ERR_INVALID, "argmax_step: the beam state needs its row-indexed lengths");
  if (forced && (rt.retire || rt.slot_row || rt.eos_at || rt.slot_seg || rt.max_len))
    return mt3::fail(MT3_ERR_INVALID, "argmax_step: teacher forcing keeps every row live and in place");
  if (ls.ss && (ls.n_ss <= 0 || ls.n_ss > 64 || ls.dim <= 0))
    return mt3::fail(MT3_ERR_INVALID, "argmax_step: the row scale needs 1 .. 64 partial sums");
  if (rp.q_out && (!y_next || !rp.ew || !rp.pw || rp.q_n % 4))
    return mt3::fail(MT3_ERR_INVALID, "argmax_step: row projection needs the next-row output and its tables");
  if (beam)
    hipLaunchKernelGGL(argmax_step_kernel<true>, dim3(B), dim3(256), 0, s, logits, vocab, ids, ids_stride, cur_tok,
                       done, n_done, step, table, pos_table, max_pos, y_next, y_ct, y_ss, dim, beam->f, beam->len,
                       beam->len_row, beam->cfg, beam->rows, nullptr, 0, rp, ls, rt);
  else
    hipLaunchKernelGGL(argmax_step_kernel<false>, dim3(B), dim3(256), 0, s, logits, vocab, ids, ids_stride, cur_tok,
                       done, n_done, step, table, pos_table, max_pos, y_next, y_ct, y_ss, dim, nullptr, nullptr, nullptr,
                       nullptr, 0, forced, forced_stride, rp, ls, rt);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// ---------------------------------------------------------------------------------------------- row retirement
// Compaction of a row group's live slots (CompactArgs, kernels.h).  Three launches on the group's stream, issued at a
// poll of the early-exit loop when the live rows fit a smaller number of 32-row GEMM tiles:
//   plan    one wave: perm[i] = i-th live slot (ascending: the order of the rows never changes), perm[rows] = n_live
//   gather  block i < n_live: state of slot perm[i] -> scratch slot i
//   scatter block i: scratch slot i -> slot i, done = 0 (i < n_live);  done = 1 (i >= n_live)
// (two passes because perm[i] >= i: slot k is the source of one block and the destination of another)
__global__ __launch_bounds__(64) void compact_plan_kernel(const int* __restrict__ done, int* __restrict__ perm, int rows) {
  const int lane = threadIdx.x;
  int n = 0;
  for (int base = 0; base < rows; base += 64) {
    const int i = base + lane;
    const bool live = i < rows && done[i] == 0;
    const unsigned long long m = __ballot(live);
    if (live) perm[n + __popcll(m & ((1ull << lane) - 1ull))] = i;
    n += __popcll(m);
  }
  if (lane == 0) perm[rows] = n;
}

template <bool GATHER>
__global__ __launch_bounds__(128) void compact_move_kernel(CompactArgs c) {
  const int i = blockIdx.x, tid = threadIdx.x;
  const int n_live = c.perm[c.rows];
  if (i >= n_live) {
    if (!GATHER && tid == 0) {
      c.done[i] = 1;
      if (c.slot_seg) c.slot_seg[i] = -1;
    }
    return;
  }
  const size_t src = GATHER ? static_cast<size_t>(c.perm[i]) : static_cast<size_t>(i), dst = i;
  const float *y = GATHER ? c.y : c.s_y, *ss = GATHER ? c.y_ss : c.s_y_ss, *q = GATHER ? c.qkvf : c.s_qkvf;
  float *yo = GATHER ? c.s_y : c.y, *sso = GATHER ? c.s_y_ss : c.y_ss, *qo = GATHER ? c.s_qkvf : c.qkvf;
  const uint2* ct = static_cast<const uint2*>(GATHER ? c.y_ct : c.s_y_ct);
  uint2* cto = static_cast<uint2*>(GATHER ? c.s_y_ct : c.y_ct);
  for (int k = tid; k < c.emb / 4; k += 128) {
    reinterpret_cast<float4*>(yo + dst * c.emb)[k] = reinterpret_cast<const float4*>(y + src * c.emb)[k];
    if (ct) cto[dst * (c.emb / 4) + k] = ct[src * (c.emb / 4) + k];
  }
  if (ss)
    for (int k = tid; k < c.emb / 16; k += 128) sso[dst * (c.emb / 16) + k] = ss[src * (c.emb / 16) + k];
  if (q)
    for (int k = tid; k < c.q_n / 4; k += 128)
      reinterpret_cast<float4*>(qo + dst * c.q_n)[k] = reinterpret_cast<const float4*>(q + src * c.q_n)[k];
  if (tid == 0) {
    if (GATHER) {
      c.s_int[4 * dst + 0] = c.slot_row[src];
      c.s_int[4 * dst + 1] = c.step[src];
      c.s_int[4 * dst + 2] = c.cur_tok[src];
      c.s_int[4 * dst + 3] = c.beam_len ? c.beam_len[src] : 0;
      if (c.slot_seg) c.s_seg[dst] = c.slot_seg[src];
      if (c.beam_f) {
        c.s_beam[2 * dst + 0] = c.beam_f[src];
        c.s_beam[2 * dst + 1] = c.beam_f[c.beam_rows + src];
      }
    } else {
      c.slot_row[dst] = c.s_int[4 * dst + 0];
      c.step[dst] = c.s_int[4 * dst + 1];
      c.cur_tok[dst] = c.s_int[4 * dst + 2];
      if (c.beam_len) c.beam_len[dst] = c.s_int[4 * dst + 3];
      if (c.slot_seg) c.slot_seg[dst] = c.s_seg[dst];
      if (c.beam_f) {
        c.beam_f[dst] = c.s_beam[2 * dst + 0];
        c.beam_f[c.beam_rows + dst] = c.s_beam[2 * dst + 1];
      }
      c.done[dst] = 0;
    }
  }
}

int launch_compact(const CompactArgs& c, hipStream_t s) {
  if (!c.done || !c.slot_row || !c.step || !c.cur_tok || !c.y || !c.s_y || !c.s_int || !c.perm || c.rows <= 0 ||
      c.emb % 16 || c.q_n % 4 || (c.y_ct && !c.s_y_ct) || (c.y_ss && !c.s_y_ss) || (c.qkvf && !c.s_qkvf) ||
      (c.beam_f && (!c.s_beam || !c.beam_len)) || (c.slot_seg && !c.s_seg))
    return mt3::fail(MT3_ERR_INVALID, "compact: bad arguments");
  hipLaunchKernelGGL(compact_plan_kernel, dim3(1), dim3(64), 0, s, c.done, c.perm, c.rows);
  hipLaunchKernelGGL(compact_move_kernel<true>, dim3(c.rows), dim3(128), 0, s, c);
  hipLaunchKernelGGL(compact_move_kernel<false>, dim3(c.rows), dim3(128), 0, s, c);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// ---------------------------------------------------------------------------------------------- slot refill
// In-flight batching (RefillArgs, kernels.h).  Three launches on the group's stream at a poll of the group's loop:
//   plan   one wave: plan[i] = i-th FINISHED slot of the group (ascending), plan[rows] = how many; the group's counter of
//          finished slots drops by the number that restart
//   cross  block (i, layer x {K, V [, scales]}, part): cross-attention K/V of segment first_seg + i from the staging chunk
//          into the cache rows of slot plan[i]  (16-byte pieces, 256 lanes, a row of H*T*64 elements in `parts` pieces)
//   slot   block i: id row of slot plan[i] -> the caller's row of the segment it decoded (beam-1: the finished
//          hypothesis, as beam1_finalize_kernel would leave it); i < n_new: restart on segment first_seg + i
__global__ __launch_bounds__(64) void refill_plan_kernel(const int* __restrict__ done, int* __restrict__ plan,
                                                          int* __restrict__ n_done, int rows, int n_new) {
  const int lane = threadIdx.x;
  int n = 0;
  for (int base = 0; base < rows; base += 64) {
    const int i = base + lane;
    const bool fin = i < rows && done[i] != 0;
    const unsigned long long m = __ballot(fin);
    if (fin) plan[n + __popcll(m & ((1ull << lane) - 1ull))] = i;
    n += __popcll(m);
  }
  if (lane == 0) {
    plan[rows] = n;
    *n_done -= n_new < n ? n_new : n;
  }
}

__global__ __launch_bounds__(256) void refill_cross_kernel(RefillArgs a, int parts) {
  const int i = blockIdx.x;
  const int n_fin = a.plan[a.rows];
  if (i >= a.n_new || i >= n_fin) return;
  const int row = a.slot_row[a.plan[i]];
  const int l = blockIdx.y / 3, what = blockIdx.y % 3;          // 0: K rows, 1: V rows, 2: scale rows (e4m3 caches)
  const char* src;
  char* dst;
  size_t bytes;
  if (what < 2) {
    bytes = a.row_bytes;
    src = a.src[l] + (static_cast<size_t>(what) * a.src_batch + a.src_entry0 + i) * bytes;
    dst = a.dst[l] + (static_cast<size_t>(what) * a.dst_batch + row) * bytes;
  } else {
    if (!a.src_sc[l]) return;
    bytes = a.sc_bytes;
    src = a.src_sc[l] + static_cast<size_t>(a.src_entry0 + i) * bytes;
    dst = a.dst_sc[l] + static_cast<size_t>(row) * bytes;
  }
  const size_t n16 = bytes >> 4;                                   // rows are multiples of 16 bytes (64 elements per key)
  const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
  u32x4* d4 = reinterpret_cast<u32x4*>(dst);
  for (size_t k = static_cast<size_t>(blockIdx.z) * 256 + threadIdx.x; k < n16; k += static_cast<size_t>(parts) * 256)
    d4[k] = __builtin_nontemporal_load(s4 + k);
}

__global__ __launch_bounds__(128) void refill_slot_kernel(RefillArgs a) {
  const int i = blockIdx.x, tid = threadIdx.x;
  if (i >= a.plan[a.rows]) return;
  const int slot = a.plan[i];
  const int row = a.slot_row[slot];
  const int seg_old = a.slot_seg[slot];
  int* idrow = a.ids + static_cast<size_t>(row) * a.ids_stride;
  if (seg_old >= 0) {
    // the finished hypothesis of the beam-1 search: live[:n] + EOS + padding (beam1_finalize_kernel); n < 0: the live one
    const int n = a.beam_len ? a.beam_len_row[row] : -1;
    int* out = a.out_ids + static_cast<size_t>(seg_old) * a.ids_stride;
    for (int k = tid; k < a.ids_stride; k += 128) out[k] = (n >= 0 && k >= n) ? (k == n ? 1 : 0) : idrow[k];
  }
  const bool restart = i < a.n_new;
  __syncthreads();                  // every lane has read the old occupant's beam length before lane 0 resets it
  // (a thread zeroes exactly the ids it has just copied out)
  if (restart)
    for (int k = tid; k < a.ids_stride; k += 128) idrow[k] = 0;
  if (tid == 0) {
    a.slot_seg[slot] = restart ? a.first_seg + i : -1;
    if (restart) {
      a.step[slot] = 0;
      a.cur_tok[slot] = 0;                                       // BOS
      a.done[slot] = 0;
      if (a.beam_f) {                                            // t5x beam_search: live log-prob 0, nothing finished
        a.beam_f[slot] = 0.f;
        a.beam_f[a.beam_rows + slot] = 0.f;
        a.beam_len[slot] = -1;
        a.beam_len_row[row] = -1;
      }
    }
  }
  if (!restart) return;
  // decoder input of position 0: Embed(BOS) + FixedEmbed[0], in the forms the step reads (embed_kernel)
  for (int k = tid * 4; k < a.emb; k += 512) {
    const float4 e4 = *reinterpret_cast<const float4*>(a.table + k), p4 = *reinterpret_cast<const float4*>(a.pos + k);
    put_row_piece(make_float4(e4.x + p4.x, e4.y + p4.y, e4.z + p4.z, e4.w + p4.w), a.y, a.y_ct, a.y_ss, slot, a.emb, k);
  }
  if (a.rp.q_out) put_row_projection(a.rp, slot, 0, 0, tid, 128);
}

int launch_refill(const RefillArgs& a, hipStream_t s) {
  if (!a.done || !a.slot_row || !a.slot_seg || !a.step || !a.cur_tok || !a.n_done || !a.y || !a.table || !a.pos || !a.ids ||
      !a.out_ids || !a.plan || a.rows <= 0 || a.n_new < 0 || a.n_new > a.rows || a.emb % 16 || a.ids_stride <= 0 ||
      (a.beam_f && (!a.beam_len || !a.beam_len_row)) || (a.y_ct && !a.y_ss) ||
      (a.rp.q_out && (!a.rp.ew || !a.rp.pw || a.rp.q_n % 4)))
    return mt3::fail(MT3_ERR_INVALID, "refill: bad arguments");
  if (a.n_new > 0 && (a.n_layers <= 0 || a.n_layers > kRefillMaxLayers || a.row_bytes % 16 || a.sc_bytes % 16 ||
                      a.src_batch <= 0 || a.dst_batch <= 0 || a.src_entry0 < 0 || a.src_entry0 + a.n_new > a.src_batch))
    return mt3::fail(MT3_ERR_INVALID, "refill: bad staging chunk");
  hipLaunchKernelGGL(refill_plan_kernel, dim3(1), dim3(64), 0, s, a.done, a.plan, a.n_done, a.rows, a.n_new);
  if (a.n_new > 0) {
    // a K or V row of one layer is H*T*64 elements (98 KB ... 393 KB): 8 blocks of 256 lanes per row keep >= 1000
    // workgroups in flight for a handful of segments
    const int parts = 8;
    hipLaunchKernelGGL(refill_cross_kernel, dim3(a.n_new, a.n_layers * 3, parts), dim3(256), 0, s, a, parts);
  }
  hipLaunchKernelGGL(refill_slot_kernel, dim3(a.rows), dim3(128), 0, s, a);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

__global__ void iota_kernel(int* dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = i;
}
int launch_iota(int* dst, int n, hipStream_t s) {
  hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, n);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// one float into device memory from a kernel ARGUMENT (no host buffer has to outlive the call)
__global__ void set_float_kernel(float* dst, float v) { *dst = v; }
int launch_set_float(float* dst, float v, hipStream_t s) {
  hipLaunchKernelGGL(set_float_kernel, dim3(1), dim3(1), 0, s, dst, v);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// end of a beam-1 decode: rows with a finished hypothesis return live[:len] + EOS (+ pad), the others
// keep their live sequence (t5x beam_search: "if no finished sequence, return the live one")
__global__ __launch_bounds__(256) void beam1_finalize_kernel(int* __restrict__ ids, int L,
                                                              const int* __restrict__ beam_len) {
  const int b = blockIdx.x, n = beam_len[b];
  if (n < 0) return;
  int* row = ids + static_cast<size_t>(b) * L;
  for (int i = n + threadIdx.x; i < L; i += 256) row[i] = i == n ? 1 : 0;
}

int launch_beam1_finalize(int* ids, int L, const int* beam_len, int B, hipStream_t s) {
  hipLaunchKernelGGL(beam1_finalize_kernel, dim3(B), dim3(256), 0, s, ids, L, beam_len);
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

extern "C" int mt3_op_residual_split(int32_t dtype, const float* d_x, void* d_x_ct, float* d_x_ss, int32_t rows,
                                     int32_t dim, void* stream) {
  if (dtype != MT3_BF16) return mt3::fail(MT3_ERR_INVALID, "residual_split: the split stream is a bf16-path format");
  return mt3k::launch_residual_split(d_x, d_x_ct, d_x_ss, rows, dim, static_cast<hipStream_t>(stream));
}

extern "C" int mt3_ids_to_tokens(const int32_t* d_ids, int32_t batch, int32_t length, int32_t num_regular,
                                 int32_t* d_tokens, void* stream) {
  return mt3k::launch_ids_to_tokens(d_ids, batch, length, num_regular, d_tokens, static_cast<hipStream_t>(stream));
}

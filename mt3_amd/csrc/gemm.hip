// MFMA GEMM for the T5 blocks of the MT3 path on gfx950:  out = epilogue(rs(A) * (A @ Wt^T)).
//
// One kernel template covers every dense layer of mt3/network.py / mt3/layers.py
// (`DenseGeneral`, layers.py:373-418, all bias-free):
//   * A [M,K]: either the f32 residual stream (converted to the compute type while it is staged
//     into LDS) or a compute-type activation;  Wt [N,K]: weight stored output-major so that both
//     MFMA operands are K-contiguous 16-byte chunks.
//   * NORM fuses T5 LayerNorm (layers.py:604-621 = RMSNorm, eps 1e-6) into the GEMM that consumes
//     it: the learned scale is folded into Wt's columns at load time, and the per-row
//     rsqrt(mean(x^2) + eps) is accumulated from the f32 A values as they stream through the
//     K loop (reduced across the lanes of a row on the DPP network) and applied in the epilogue --
//     no separate normalisation pass over HBM.  In the bf16 decode loop the row instead arrives as
//     a bf16 copy plus the exact f32 sums of squares of its 16-column groups (GemmArgs::a_ss), left
//     there by whichever kernel produced the row, and the GEMM only adds up those K/16 partials.
//   * epilogues: STORE (compute type), RESID (f32 out += acc, the residual add of
//     network.py:66,83,120,136,150; on request also the bf16 copy + partial sums above), GEGLU (gelu(wi_0 x) * wi_1 x of layers.py:460-473 with the
//     two weight matrices interleaved in 16-column groups so gate and linear land in the same
//     lane), POS (+ sinusoidal table row, network.py:174-180), F32 (logits, network.py:256-261),
//     HEADS (cross-attention K/V written head-major [2][B][H][T][64] for the decode kernel).
// Tiling: workgroup tile BMxBN, WMxWN waves, each wave FMxFN 16x16 MFMA fragments, K step BK;
// global -> register prefetch of tile t+1 overlaps the MFMAs of tile t; LDS rows padded by two
// 16-byte chunks (conflict-free 16-lane fragment reads); 2-byte outputs leave as dwords (lane
// pairs swap over DPP); XCD-aware block remap so that the column tiles of one row panel share an L2.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "device.h"
#include "kernels.h"

// phase marks for tools/micro/gemm_phases.hip (which defines the macro before including this file); no-ops here
#ifndef MT3_PROF_MARK
#define MT3_PROF_MARK(i)
#endif
#ifndef MT3_PROF_DECL
#define MT3_PROF_DECL
#endif
// tools/micro/glds_probe.hip builds the LDS-DMA tile with parts left out to see what bounds it (0 in the product):
// 1 = no fragment reads / MFMAs, 2 = no DMA after the prologue's DEPTH slices, 4 = no epilogue
#ifndef MT3_GLDS_PROBE
#define MT3_GLDS_PROBE 0
#endif
#ifndef MT3_GLDS_NS
#define MT3_GLDS_NS 3      // ring stages of the LDS-DMA tile (3: 48.5 KB of LDS, three workgroups per CU; 4: two)
#endif
// the tile's bf16 outputs are written once per GEMM by exactly one workgroup: streamed with non-temporal stores they do
// not push the weight tiles and the shared A panels out of the XCD's 4 MB L2 (probe: a 268 MB STORE 222 -> 183 us)
#ifndef MT3_GLDS_NO_NT
#define MT3_GLDS_ST(v, p) __builtin_nontemporal_store((v), (p))
#else
#define MT3_GLDS_ST(v, p) (*(p) = (v))
#endif
#ifndef MT3_GLDS_BK
#define MT3_GLDS_BK 32     // K slice of the LDS-DMA tile: 32 (64-byte row pieces) or 64 (whole 128-byte lines)
#endif
// round-6 A/B builds (tools/ab_r6.py compiles the sources once per value; 0 in the product): bit 1 = the decode-sized
// tile's waves do NOT raise their issue priority (the product before round 6), bit 6 = its kernel arguments are NOT
// pinned behind one scalar round trip.  Measured and removed again
// (profiles/r6_ab_decode_gemm_variants.txt): weight slices loaded non-temporally (-4 %), the f32 two-source fold launch in
// K slices of 512 -- three dependent trips instead of six (+-0: the dependent trips are not what the launch waits for);
// half the WEIGHT bytes per workgroup on twice the workgroups -- fold on 32 x 16, GEGLU on 32 x 32, out-projections on
// 32 x 16 tiles of two waves (-5 %: profiles/r6_ab_narrow_tiles.txt).
#ifndef MT3_EXP
#define MT3_EXP 0
#endif
#if MT3_EXP & 32
// bit 5 = IN-SITU phase accounting of the decode-sized tiles (tools/gemm_phases_in_situ.py): lane 0 of every workgroup
// adds its 100 MHz wall-clock spans to g_ph[epilogue]: [0] workgroups, [1] entry -> first K slice in LDS, [2] -> K loop
// done, [3] -> epilogue stores acknowledged, [4] entry -> first barrier (the first slice's
// loads requested; [1] - [4] = their arrival + the staging pass)
__device__ unsigned long long g_ph[16][8];
#undef MT3_PROF_DECL
#undef MT3_PROF_MARK
#define MT3_PROF_DECL unsigned long long ph_t0 = 0, ph_t1 = 0, ph_t2 = 0, ph_ta = 0
#define MT3_PROF_MARK(i)                                                                         \
  do {                                                                                           \
    if constexpr (BM <= 64) {                                                                    \
      if ((i) == 4) __builtin_amdgcn_s_waitcnt(0);                                               \
      if (threadIdx.x == 0) {                                                                    \
        const unsigned long long t_ = wall_clock64();                                            \
        if ((i) == 0) ph_t0 = t_;                                                                \
        else if ((i) == 1) { if (!ph_ta) ph_ta = t_; }                                           \
        else if ((i) == 2) { if (!ph_t1) ph_t1 = t_; }                                           \
        else if ((i) == 3) ph_t2 = t_;                                                           \
        else if ((i) == 4) {                                                                     \
          atomicAdd(&g_ph[EPI][0], 1ull);                                                        \
          atomicAdd(&g_ph[EPI][1], ph_t1 - ph_t0);                                               \
          atomicAdd(&g_ph[EPI][2], ph_t2 - ph_t1);                                               \
          atomicAdd(&g_ph[EPI][3], t_ - ph_t2);                                                  \
          atomicAdd(&g_ph[EPI][4], ph_ta - ph_t0);                                               \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  } while (0)
#endif

namespace mt3k {

// stage one K-slice of A (optionally f32 -> compute type, accumulating sum of squares for the fused
// RMSNorm) and of Wt from global memory into registers
// Which 16-byte chunk of a [rows][CPR] tile thread `tid` moves in pass p.  Power-of-two CPR: CPR consecutive
// lanes share a row (the fused-norm reduction relies on that); otherwise (K = 384: 48 chunks per row) chunks
// are dealt linearly, NT per pass.
template <int CPR, int NT>
__device__ __forceinline__ void tile_chunk(int tid, int p, int* row, int* chunk) {
  if constexpr ((CPR & (CPR - 1)) == 0 && NT % CPR == 0) {
    *row = tid / CPR + p * (NT / CPR);
    *chunk = tid % CPR;
  } else {
    const int c = tid + p * NT;
    *row = c / CPR;
    *chunk = c % CPR;
  }
}

template <typename CT, bool A_F32, bool NORM, int A_PASSES, int B_PASSES, int CPR, int NT>
__device__ __forceinline__ void gemm_load_tiles(u32x4 (&a_reg)[A_PASSES], u32x4 (&b_reg)[B_PASSES],
                                                float (&ss)[A_PASSES], const void* gA, const void* gW, int m0,
                                                int n0, int tid, int gM, int gLda, int gK, int k0,
                                                const void* gA2 = nullptr, int gLda2 = 0, int k1 = 0) {
  constexpr int KPL = CTraits<CT>::KPL;
  // two-source rows (kEpiResidS): weight columns [k1, K) multiply the rows of A2 (slice-uniform)
  int ka = k0;
  if (gA2 != nullptr && k0 >= k1) {
    gA = gA2;
    gLda = gLda2;
    ka = k0 - k1;
  }
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    int ld_row, ld_chunk;
    tile_chunk<CPR, NT>(tid, p, &ld_row, &ld_chunk);
    int row = m0 + ld_row;
    row = row < gM ? row : gM - 1;                       // clamp: out-of-range rows are never stored
    const size_t e = static_cast<size_t>(row) * gLda + ka + ld_chunk * KPL;
    if constexpr (A_F32) {
      const float4* src = reinterpret_cast<const float4*>(static_cast<const float*>(gA) + e);
      if constexpr (KPL == 8) {
        const float4 u = src[0], v = src[1];
        const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        if constexpr (NORM) {
          // explicit FMA chain: every row must see the SAME rounding sequence whichever unrolled pass
          // (= row position inside the tile) handles it -- results are bit-identical under a permutation
          // of the batch (left to the compiler, some passes were contracted to FMA and others not)
#pragma unroll
          for (int i = 0; i < 8; ++i) ss[p] = __builtin_fmaf(f[i], f[i], ss[p]);
        }
        a_reg[p] = pack_bf16x8(f);
      } else {
        const float4 u = src[0];
        if constexpr (NORM)
          ss[p] = __builtin_fmaf(u.w, u.w, __builtin_fmaf(u.z, u.z, __builtin_fmaf(u.y, u.y, __builtin_fmaf(u.x, u.x, ss[p]))));
        a_reg[p] = pack_f32x4(u.x, u.y, u.z, u.w);
      }
    } else {
      a_reg[p] = *reinterpret_cast<const u32x4*>(static_cast<const CT*>(gA) + e);
    }
  }
#pragma unroll
  for (int p = 0; p < B_PASSES; ++p) {
    int ld_row, ld_chunk;
    tile_chunk<CPR, NT>(tid, p, &ld_row, &ld_chunk);
    const int row = n0 + ld_row;                           // N is a multiple of BN: always in range
    const size_t e = static_cast<size_t>(row) * gK + k0 + ld_chunk * KPL;
    b_reg[p] = *reinterpret_cast<const u32x4*>(static_cast<const CT*>(gW) + e);
  }
}

// ------------------------------------------------------------------ epilogues (shared by the tile kernels below)
struct EpiCtx {
  void* out;
  const float* aux;
  int M, N, ldo, seq;       // N: width of the primary output region (= n_split for the *Q epilogues)
  void* out_ct;
  float* out_ss;
  float* out2;              // *Q epilogues: f32 [M][ld2] for tile columns >= n_split
  int n_split, ld2;
  const float* resid_src;   // RESID: old values of the f32 output region (== out unless the update is out of place)
};

// acc[i][j]: the wave's (wm, wn) sub-tile as FM x FN 16x16 C fragments of the workgroup tile at (m0, n0);
// row_rs(lrow) = 1 / rms of tile row lrow (or 1)
// PRE: pre[i][j][r] already holds the element of the f32 output region that a RESID epilogue is about to add to (the
// kernel asked for it before its K loop: on the latency-bound decode tiles the read-modify-write's read would
// otherwise be one more dependent memory round trip at the very end of the launch)
template <typename CT, int EPI_, int FM, int FN, bool PRE, typename RowRs>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[FM][FN], int wm, int wn, int lane, int m0, int n0,
                                              const EpiCtx& c, RowRs row_rs, const float (&pre)[FM][FN][4]) {
  const int frag_row = lane & 15, frag_g = lane >> 4;
  if constexpr (EPI_ == kEpiStoreQ || EPI_ == kEpiResidQ || EPI_ == kEpiResidS) {
    if (n0 >= c.n_split) {       // (tile-uniform) the second product: plain f32, no row scale
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * FM * 16 + i * 16 + frag_g * 4 + r;
          if (row >= c.M) continue;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            float* o = c.out2 + static_cast<size_t>(row) * c.ld2 + (n0 - c.n_split) + wn * FN * 16 + j * 16 + frag_row;
            if constexpr (EPI_ == kEpiResidQ) *o = (PRE ? pre[i][j][r] : *o) + acc[i][j][r];
            else *o = acc[i][j][r];
          }
        }
      return;
    }
  }
  constexpr int EPI = EPI_ == kEpiStoreQ ? MT3_EPI_STORE : (EPI_ == kEpiResidQ || EPI_ == kEpiResidS) ? MT3_EPI_RESID : EPI_;
  // ---- epilogue: C fragment (i, j): rows (lane>>4)*4 + r, col lane & 15
  // 2-byte outputs are never stored one element at a time (a sub-dword store costs a read-modify-write in the
  // cache: the bf16 STORE epilogue of a decode GEMM took 2.5 us against 0.7 us for the f32 RESID one): lanes l
  // and l^1 hold adjacent columns, so they swap over DPP and the even lane stores rows r = 0, 1, the odd lane
  // rows r = 2, 3 of the pair as whole dwords.
  constexpr bool PAIRED = sizeof(CT) == 2 && (EPI == MT3_EPI_STORE || EPI == MT3_EPI_GEGLU || EPI == MT3_EPI_HEADS);
  if constexpr (PAIRED) {
    const int odd = frag_row & 1;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int lrow0 = wm * FM * 16 + i * 16 + frag_g * 4;
#pragma unroll
      for (int j = 0; j < FN; j += (EPI == MT3_EPI_GEGLU ? 2 : 1)) {
        float mine[4], other[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float rs = row_rs(lrow0 + r);
          if constexpr (EPI == MT3_EPI_GEGLU) mine[r] = gelu_tanh_fast(acc[i][j][r] * rs) * (acc[i][j + 1][r] * rs);   // (bf16 out)
          else mine[r] = acc[i][j][r] * rs;
          other[r] = lane_xor1(mine[r]);
        }
        // logical output column of this lane's element, and of the pair's even element
        const int col = EPI == MT3_EPI_GEGLU ? ((n0 + wn * FN * 16 + j * 16) >> 1) + frag_row
                                             : n0 + wn * FN * 16 + j * 16 + frag_row;
        const int col_even = col - odd;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = odd * 2 + h;
          const int row = m0 + lrow0 + r;
          if (row >= c.M) continue;
          // (selects, not mine[r]: a runtime index would put the arrays in scratch memory)
          const float lo = odd ? other[2 + h] : mine[h], hi = odd ? mine[2 + h] : other[h];
          const unsigned pair = pack_bf16x2(lo, hi);
          size_t dst;
          if constexpr (EPI == MT3_EPI_HEADS) {   // col = kv*H*64 + h*64 + d, row = b*T + t  ->  [kv][b][h][t][d]
            const int hd = c.N >> 1;               // H * 64
            const int kv = col_even / hd, hh = (col_even % hd) >> 6, d = col_even & 63;
            const int bb = row / c.seq, t = row % c.seq, H = hd >> 6, B = c.M / c.seq;
            dst = ((((static_cast<size_t>(kv) * B + bb) * H + hh) * c.seq) + t) * 64 + d;
          } else {
            dst = static_cast<size_t>(row) * c.ldo + col_even;
          }
          *reinterpret_cast<unsigned*>(static_cast<CT*>(c.out) + dst) = pair;
        }
      }
    }
  } else if constexpr (EPI == MT3_EPI_RESID) {
    // residual update; on request also the sum of squares of each row over this fragment's 16 columns (exact f32,
    // reduced on the DPP row) and -- bf16 -- the compute-type copy of the new rows (dword stores of column pairs)
    const int odd = frag_row & 1;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int lrow0 = wm * FM * 16 + i * 16 + frag_g * 4;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * FN * 16 + j * 16 + frag_row;
        float vnew[4], other[4], part[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + lrow0 + r;
          const size_t at = static_cast<size_t>(row < c.M ? row : c.M - 1) * c.ldo + col;
          vnew[r] = (PRE ? pre[i][j][r] : c.resid_src[at]) + acc[i][j][r];
          if (row < c.M) static_cast<float*>(c.out)[at] = vnew[r];
        }
        if (c.out_ss) {
#pragma unroll
          for (int r = 0; r < 4; ++r) part[r] = row16_sum(vnew[r] * vnew[r]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + lrow0 + r;
            if (frag_row == 0 && row < c.M)
              c.out_ss[static_cast<size_t>(row) * (c.N >> 4) + ((n0 + wn * FN * 16 + j * 16) >> 4)] = part[r];
          }
        }
        if constexpr (sizeof(CT) == 2) {
          if (c.out_ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) other[r] = lane_xor1(vnew[r]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int row = m0 + lrow0 + odd * 2 + h;
              if (row >= c.M) continue;
              const float lo = odd ? other[2 + h] : vnew[h], hi = odd ? vnew[2 + h] : other[h];
              *reinterpret_cast<unsigned*>(static_cast<CT*>(c.out_ct) + static_cast<size_t>(row) * c.ldo + col - odd) =
                  pack_bf16x2(lo, hi);
            }
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = wm * FM * 16 + i * 16 + frag_g * 4 + r;
        const int row = m0 + lrow;
        if (row >= c.M) continue;
        const float rs = row_rs(lrow);
        if constexpr (EPI == MT3_EPI_GEGLU) {
          CT* out = static_cast<CT*>(c.out);
#pragma unroll
          for (int j = 0; j < FN; j += 2) {
            const int col = n0 + wn * FN * 16 + j * 16;                  // multiple of 32
            const float gate = acc[i][j][r] * rs, lin = acc[i][j + 1][r] * rs;
            out[static_cast<size_t>(row) * c.ldo + (col >> 1) + frag_row] = to_ct<CT>(gelu_tanh(gate) * lin);
          }
        } else {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int col = n0 + wn * FN * 16 + j * 16 + frag_row;
            const float v = acc[i][j][r] * rs;
            if constexpr (EPI == MT3_EPI_STORE) {
              static_cast<CT*>(c.out)[static_cast<size_t>(row) * c.ldo + col] = to_ct<CT>(v);
            } else if constexpr (EPI == MT3_EPI_RESID) {
              float* o = static_cast<float*>(c.out) + static_cast<size_t>(row) * c.ldo + col;
              *o = *o + v;
            } else if constexpr (EPI == MT3_EPI_POS) {
              static_cast<float*>(c.out)[static_cast<size_t>(row) * c.ldo + col] =
                  v + c.aux[static_cast<size_t>(row % c.seq) * c.N + col];
            } else if constexpr (EPI == MT3_EPI_F32) {
              static_cast<float*>(c.out)[static_cast<size_t>(row) * c.ldo + col] = v;
            } else {  // MT3_EPI_HEADS: col = kv*H*64 + h*64 + d, row = b*T + t  ->  [kv][b][h][t][d]
              const int hd = c.N >> 1;                       // H * 64
              const int kv = col / hd, h = (col % hd) >> 6, d = col & 63;
              const int b = row / c.seq, t = row % c.seq, H = hd >> 6, B = c.M / c.seq;
              const size_t dst = ((((static_cast<size_t>(kv) * B + b) * H + h) * c.seq) + t) * 64 + d;
              static_cast<CT*>(c.out)[dst] = to_ct<CT>(v);
            }
          }
        }
      }
    }
  }
}

// (Measured in round 3 and removed in round 4, DESIGN.md section 3: eight waves per f32 tile with the K-groups of a
// slice split two ways -- 1 % slower, the extra LDS round and two barriers cost what the halved MFMA chain saves; two K
// slices in flight on a second set of staging registers -- 4-16 % slower, the second slice queues up in the same CU's
// L1 path in front of the first slice's last lines.)
template <typename CT, int BM, int BN, int BK, int WM, int WN, bool A_F32, bool NORM, int EPI, int NPV = 8>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmArgs g) {
  constexpr int NT = WM * WN * 64;
  constexpr int KPL = CTraits<CT>::KPL;
  constexpr int KG = CTraits<CT>::KGROUP;
  constexpr int CPR = BK / KPL;          // 16-byte chunks per tile row
  // LDS row length in elements: two pad chunks make the row stride 32 (mod 64) bytes, the stride at which the
  // 16-lane groups of a ds_read_b128 fragment read (16 rows x one 16-byte chunk) touch every bank exactly once;
  // with one pad chunk (stride 16 mod 64) they are 2-way conflicted (8 instead of 4 LDS cycles per read)
  constexpr int ROWE = BK + 2 * KPL;
  constexpr int FM = BM / (WM * 16), FN = BN / (WN * 16);
  constexpr int A_PASSES = BM * CPR / NT, B_PASSES = BN * CPR / NT;
  constexpr int ROWS_PER_PASS = NT / CPR;
  static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0, "tile/threads mismatch");
  static_assert(!NORM || (NT % CPR == 0 && (CPR & (CPR - 1)) == 0 && CPR <= 64),
                "fused norm: CPR a power of two, <= one wave, dividing NT");
  static_assert(BK % KG == 0, "BK must be a multiple of the MFMA K-group");
  static_assert(!NORM || A_F32, "NORM needs the f32 A operand");
  static_assert(EPI != MT3_EPI_GEGLU || (FN % 2 == 0), "GEGLU pairs fragments");

  __shared__ __attribute__((aligned(16))) CT As[BM * ROWE];
  __shared__ __attribute__((aligned(16))) CT Bs[BN * ROWE];
  constexpr int NP = CPR > 16 ? CPR / 16 : 1;      // partial sums of squares per tile row (one per 16-lane DPP row)
  __shared__ float ss_part[NORM ? BM * NP : 1];
  __shared__ float rs_x[NORM ? 1 : BM];            // norm == 2: row scales from the producer's partial sums

  // EVERY field of the argument struct in SGPRs behind ONE scalar-memory round trip.  Left to itself the compiler sinks
  // the s_loads into the blocks that first use a field: the decode-sized tile began with four DEPENDENT batches (shape ->
  // tile origin -> operand pointers -> epilogue pointers), each a trip to the kernel-argument segment, before it
  // requested its first operand byte (tools/gemm_phases_in_situ.py: 2.3-2.9 us from entry to the first slice's loads
  // requested in situ, 1.6-2.0 alone on the chip).  The empty asm needs all of them at once; later uses are the same
  // invariant loads and fold into these registers.
#if !(MT3_EXP & 64)
  asm volatile("" ::"s"(g.A), "s"(g.Wt), "s"(g.out), "s"(g.aux), "s"(g.M), "s"(g.N), "s"(g.K), "s"(g.lda), "s"(g.ldo),
               "s"(g.seq_len), "s"(g.a_ss), "s"(g.out_ct), "s"(g.out_ss), "s"(g.out2), "s"(g.n_split), "s"(g.ld2),
               "s"(g.A2), "s"(g.lda2), "s"(g.k_split), "s"(g.resid_src), "s"(g.n_major), "s"(gridDim.x));
#endif
  // scalars out of the by-value argument struct (never take its address: that forces a private copy)
  const void* const gA = g.A;
  const void* const gW = g.Wt;
  void* const gO = g.out;
  const float* const gAux = g.aux;
  const int gM = g.M, gN = g.N, gK = g.K, gLda = g.lda, gLdo = g.ldo, gSeq = g.seq_len;
  const float* const gAss = g.a_ss;
  void* const gOutCt = g.out_ct;
  float* const gOutSs = g.out_ss;

  MT3_PROF_DECL;
  MT3_PROF_MARK(0);
  // Decode-sized tiles run beside the other row groups' attention launches, whose waves share the CU's issue arbiter
  // with this tile's four: at the default priority the latency-bound tile waits its turn behind HBM-bound waves that
  // lose nothing by waiting.  Priority 3 for the whole (short) kernel: f32 headline 466-472 -> 482-484 audio-s/s, the
  // 1024-step decode 1087-1103 -> 1061-1066 ms (profiles/r6_ab_decode_gemm_variants.txt); same instructions, same bits.
  if constexpr ((MT3_EXP & 2) == 0 && BM <= 64) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = gN / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  if (g.n_major) {
    const int tiles_m = gridDim.x / tiles_n;
    m0 = (lid % tiles_m) * BM;
    n0 = (lid / tiles_m) * BN;
  }

  const int ld_row = tid / CPR, ld_chunk = tid % CPR;
  // norm == 2: the K/16 (<= 4 * NPV) exact partial sums of squares of this thread's tile row, requested up front
  // and folded only after the operand loads have been issued (the fold must not delay them)
  // (UNCONDITIONAL loads from always-valid addresses -- the row's partials, or with no a_ss the first bytes of the weight
  // matrix -- at a clamped index, as in gemm_glds_kernel: under a per-thread, per-element "load or zero" condition the
  // compiler masked each load separately and waited for the first one before it issued the operand loads: a memory
  // round trip at the head of every norm-fused decode launch)
  float4 pv[NPV];
  const bool scale_rows = !NORM && gAss != nullptr && tid < BM;
  const int npv = gAss ? (gK >> 6) : 1;
  if constexpr (!NORM) {
    const int prow = m0 + (tid & (BM - 1)) < gM ? m0 + (tid & (BM - 1)) : gM - 1;
    const float4* p4 = gAss ? reinterpret_cast<const float4*>(gAss + static_cast<size_t>(prow) * (gK >> 4))
                            : reinterpret_cast<const float4*>(gW);
#pragma unroll
    for (int u = 0; u < NPV; ++u) pv[u] = p4[u < npv ? u : npv - 1];
  }
  // decode-sized RESID tiles (bf16): this lane's elements of the residual rows, requested before anything else
  constexpr bool kPre = (EPI == MT3_EPI_RESID || EPI == kEpiResidQ || EPI == kEpiResidS) && FM * FN <= 2;
  constexpr bool kSplitEpi = EPI == kEpiStoreQ || EPI == kEpiResidQ || EPI == kEpiResidS;
  const int ld2 = kSplitEpi ? (g.ld2 ? g.ld2 : gN - g.n_split) : 0;
  const bool second = kSplitEpi && n0 >= g.n_split;        // (tile-uniform) this tile belongs to the second product
  // kEpiResidS: second-product tiles run over the whole two-source K, the RESID tiles over its first k_split columns
  const int kend = EPI == kEpiResidS ? (second ? gK : g.k_split) : gK;
  const void* const gA2 = EPI == kEpiResidS && second ? g.A2 : nullptr;
  const int gLda2 = g.lda2, k1 = g.k_split;
  float xpre[FM][FN][4];
  const float* const gResidSrc = g.resid_src ? g.resid_src : static_cast<const float*>(gO);
  if constexpr (kPre) if (!(EPI == kEpiResidS && second)) {
    const float* src = gResidSrc;
    int ld = gLdo, c0 = n0;
    if constexpr (EPI == kEpiResidQ) {
      if (second) {                      // the second product's f32 region
        src = g.out2;
        ld = ld2;
        c0 = n0 - g.n_split;
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * FM * 16 + i * 16 + (lane >> 4) * 4 + r;
          xpre[i][j][r] = src[static_cast<size_t>(row < gM ? row : gM - 1) * ld + c0 + wn * FN * 16 + j * 16 + (lane & 15)];
        }
  }
  u32x4 a_reg[A_PASSES], b_reg[B_PASSES];
  float ss[A_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) ss[p] = 0.f;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_g = lane >> 4;
  const CT* a_base = &As[(wm * FM * 16 + frag_row) * ROWE + frag_g * KPL];
  const CT* b_base = &Bs[(wn * FN * 16 + frag_row) * ROWE + frag_g * KPL];

  gemm_load_tiles<CT, A_F32, NORM, A_PASSES, B_PASSES, CPR, NT>(a_reg, b_reg, ss, gA, gW, m0, n0, tid, gM, gLda, gK, 0,
                                                                gA2, gLda2, k1);
  if constexpr (!NORM) {
    if (scale_rows) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < NPV; ++u) {
        const float4 v = u < npv ? pv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        t = (((t + v.x) + v.y) + v.z) + v.w;                                                // fixed order per row
      }
      rs_x[tid] = rsqrtf(t / static_cast<float>(gK) + 1e-6f);                               // read after the K loop's barriers
    }
  }
  // one K slice: staging registers -> LDS, refill them with slice k0 + BK, multiply
  for (int k0 = 0; k0 < kend; k0 += BK) {
    __syncthreads();                    // every wave is done reading the previous tile
    MT3_PROF_MARK(1);
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      int r, ch;
      tile_chunk<CPR, NT>(tid, p, &r, &ch);
      *reinterpret_cast<u32x4*>(&As[r * ROWE + ch * KPL]) = a_reg[p];
    }
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) {
      int r, ch;
      tile_chunk<CPR, NT>(tid, p, &r, &ch);
      *reinterpret_cast<u32x4*>(&Bs[r * ROWE + ch * KPL]) = b_reg[p];
    }
    __syncthreads();
    MT3_PROF_MARK(2);
    if (k0 + BK < kend)                 // the next slice in flight while the MFMAs below run
      gemm_load_tiles<CT, A_F32, NORM, A_PASSES, B_PASSES, CPR, NT>(a_reg, b_reg, ss, gA, gW, m0, n0, tid, gM, gLda, gK,
                                                                    k0 + BK, gA2, gLda2, k1);
#pragma unroll
    for (int kk = 0; kk < BK / KG; ++kk) {
      u32x4 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const u32x4*>(a_base + i * 16 * ROWE + kk * KG);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(b_base + j * 16 * ROWE + kk * KG);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mfma_chunk<CT>(af[i], bf[j], acc[i][j]);
    }
  }
  MT3_PROF_MARK(3);

  if constexpr (NORM) {
    // Each tile row was streamed by CPR consecutive lanes.  Their partial sums of squares meet on the DPP
    // network inside each 16-lane row (4 v_add_f32_dpp per value; the 48 dependent ds_bpermute shuffles this
    // replaces cost 2 us of a 7 us decode GEMM), and the <= 4 row partials are added by the reader.  The
    // reduction tree is the same for every tile row, so results do not depend on where in a tile a row lands.
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      float v = ss[p];
      if constexpr (CPR >= 2) v += lane_xor1(v);
      if constexpr (CPR >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
      if constexpr (CPR >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
      if constexpr (CPR >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
      const bool writer = CPR >= 16 ? (ld_chunk & 15) == 0 : ld_chunk == 0;
      if (writer) ss_part[(ld_row + p * ROWS_PER_PASS) * NP + (CPR > 16 ? ld_chunk / 16 : 0)] = v;
    }
    __syncthreads();
  }
  // 1 / rms of tile row `lrow` (fused RMSNorm: the scale vector is folded into the weights)
  auto row_rs = [&](int lrow) -> float {
    if constexpr (!NORM) return gAss ? rs_x[lrow] : 1.f;
    float t = ss_part[lrow * NP];
#pragma unroll
    for (int q = 1; q < NP; ++q) t += ss_part[lrow * NP + q];
    return rsqrtf(t / static_cast<float>(gK) + 1e-6f);
  };
  MT3_PROF_MARK(5);
  const EpiCtx ec{gO, gAux, gM, kSplitEpi ? g.n_split : gN, gLdo, gSeq, gOutCt, gOutSs, g.out2, g.n_split, ld2, gResidSrc};
  gemm_epilogue<CT, EPI, FM, FN, kPre>(acc, wm, wn, lane, m0, n0, ec, row_rs, xpre);
  MT3_PROF_MARK(4);
}

// (Round 4 built and measured a second decode-sized tile -- 16 x 16 / 16 x 32 outputs per workgroup, the four waves
// splitting K four ways and loading their operand chunks straight from L2 in the MFMA lane layout, no LDS staging, one
// barrier for the 3 KB reduction -- on the theory that the decode GEMMs are pure latency (rocprofv3, f32, EOS schedule:
// fold launch 18.3 us, GEGLU 12.1, out-projections 7.4).  Parity-green, and SLOWER everywhere: f32 1100 -> 1362 ms per
// 1024-step decode (1269 with the fold launch on 16 x 32), EOS-schedule decode 380 -> 467 / 450 ms; bf16 590 -> 764 / 693
// and 212 -> 264 / 254 (profiles/r4_ab_split_k_tiles.txt).  Four times the workgroups re-read four times the operand
// bytes through the L1s as 64-byte row pieces, and the wider tile losing LESS says the launches are bound by L1 / L2
// request throughput, not by the dependent chain inside a workgroup.  64-row f32 tiles (one weight fetch per 64-row
// group: 64 x 16 x 256 for the fold launch, 64 x 32 x 256 for GEGLU) measured 1 % slower as well
// (profiles/r4_ab_tall_tiles_groups_wait.txt), and so did the opposite: 16 x 32 / 16 x 64 staged tiles on two waves for
// the launches of a 64-row group (all of which leave more than half of the CUs idle) -- three quarters of the bytes per
// workgroup on twice the workgroups: 1101 -> 1155 ms (profiles/r4_ab_16_row_f32_tiles.txt).  Fewer bytes per workgroup,
// fewer bytes in total, more workgroups, fewer workgroups: each is slower than the staged 32 x 32 tile, which stays.)

// ------------------------------------------------------------------ encoder-sized tile, f32 operands as three bf16 planes
// Round 4, the f32 engine's encoder.  gfx950 multiplies f32 operands at 1/16 of its bf16 rate
// (v_mfma_f32_16x16x4_f32: 32 cycles for K = 4; v_mfma_f32_16x16x32_bf16: 16 cycles for K = 32), and the f32 encoder sat at
// 0.70 of THAT peak -- five times the bf16 encoder's time.  An f32 value is EXACTLY hi + mid + lo + r with three bf16
// terms (hi = rne(x), mid = rne(x - hi), lo = rne(x - hi - mid); both subtractions are exact) and |r| <= 2^-27 |x|, so
//   a . w = ah.wh + (ah.wm + am.wh) + (ah.wl + al.wh + am.wm) + [terms <= 2^-26 |a||w|, dropped]
// -- SIX bf16 matrix instructions per 32 k instead of eight f32 ones of twice the duration: 96 against 256 cycles, with
// bf16 products exact in f32 and the instruction's adder keeping >= 23 bits (tools/micro/mfma_bf16_accuracy.hip,
// profiles/r4_mfma_bf16_accuracy.txt: one instruction 1.4e-7 of sum |p| whatever the exponent spread; a K = 512 dot
// product of unit normals: this scheme 1.3e-7 of sum |p|, the f32 instruction chain 2.1e-7, two terms per operand
// 8.6e-7).  So this is NOT a reduced-precision mode: it is at least as exact as the f32 instruction it replaces.
// Weights are split once at finalize (three [N][K] bf16 planes = 1.5x the f32 bytes); activations are split while they
// are staged (they arrive as f32 from the previous epilogue).  128 x 128 x 32 tile, 2 x 2 waves of 64 x 64, register
// staged with the next slice in flight under the 96 MFMAs of the current one, six LDS planes of [128][32 + 16] bf16
// (73.7 KB: two workgroups per CU), fused RMSNorm statistics from the f32 A stream exactly as in gemm_kernel, the same
// epilogue code (CT = float: outputs are f32).
__device__ __forceinline__ void split3_bf16(const float (&x)[8], u32x4* h, u32x4* m, u32x4* l) {
  float hf[8], mf[8], lf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hi = static_cast<__bf16>(x[i]);
    const float r1 = x[i] - static_cast<float>(hi);             // exact
    const __bf16 mi = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(mi);               // exact
    hf[i] = static_cast<float>(hi);
    mf[i] = static_cast<float>(mi);
    lf[i] = r2;
  }
  *h = pack_bf16x8(hf);       // (re-rounding an exactly representable value is the identity)
  *m = pack_bf16x8(mf);
  *l = pack_bf16x8(lf);
}

// Measured (profiles/r4_ab_f32_encoder_x6.txt): encoder + cross-K/V 28.6 -> 20.8 ms at B = 256 (7.6 -> 5.5 at B = 64): 1.37x, not
// the 2.7x of the instruction rates -- with six planes the tile is LDS-bandwidth-bound (288 KB through the LDS per CU and
// slice: ~2250 cycles against 1536 of MFMA).  A 256-row tile (BM_ = 256: 36 fragment reads per 192 MFMAs instead of 24 per
// 96, but 110 KB of LDS = ONE workgroup per CU) measured 26.1 ms -- the second co-resident workgroup is worth more than
// the read ratio, as with the bf16 LDS-DMA tile -- and is not instantiated.  Two K slices in flight (below): 20.8 -> 19.6 ms.
// W fragments loaded straight into registers in the MFMA lane layout (16 B per lane from row n: no LDS pass for W, 72
// instead of 168 KB through the LDS per workgroup and slice), with one or two sets in flight and two or three workgroups
// per CU: 21.9 - 23.3 ms -- the 64-byte row pieces cost more in the L1 path than the LDS passes they replace; removed.
template <bool NORM, int EPI, int BM_ = 128>
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(GemmArgs g, const __bf16* __restrict__ Wm,
                                                       const __bf16* __restrict__ Wl) {
  constexpr int BM = BM_, BN = 128, BK = 32, ROWE = BK + 16, FM = BM / 32, FN = 4;   // 4 chunks of 8 per tile row
  constexpr int AP = BM / 64;                          // A passes per thread: 64 tile rows per pass
  __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * ROWE];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[3][BN * ROWE];
  __shared__ float ss_part[NORM ? BM : 1];

  const float* const gA = static_cast<const float*>(g.A);
  const __bf16* const gW[3] = {static_cast<const __bf16*>(g.Wt), Wm, Wl};
  const int gM = g.M, gN = g.N, gK = g.K, gLda = g.lda;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = gN / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

  // thread -> (row, chunk) of pass p (64 tile rows per pass, 4 chunks of 8 elements per row): every 8 consecutive lanes
  // take 4 rows x 2 chunks -- at the 96-byte row stride (which makes the fragment READS conflict-free) the eight 16-byte
  // slots of such a group are distinct mod 128 bytes, so a ds_write_b128 pass is conflict-free too; with 2 rows x 4
  // chunks per 8 lanes (a quad per row) the second row started 96 bytes after the first and overlapped its first 32
  // bytes' banks: a third of the tile's LDS-active cycles were bank conflicts (counter passes of tools/gpu_pmc_enc.sh:
  // 0.05 - 0.07 per wave cycle, 0.0000 now, LDS-active cycles down by a third; the TIME did not move -- the tile is not
  // LDS-bound: matrix pipes busy 0.52 / 0.56 of the cycles in the two big launches, profiles/r4_pmc_encoder_summary.json).
  // The four lanes of a row are t, t ^ 1, t ^ 8, t ^ 9.
  const int ld_chunk = (tid & 1) | ((tid >> 2) & 2);
  const int ld_row = ((tid >> 1) & 3) | ((tid >> 4) << 2);
  // TWO K slices in flight in two sets of staging registers (40 VGPRs each): operands arrive from L2 / MALL / HBM in
  // 2-3 us, a slice's 96 MFMAs take 0.64 us -- with one slice of look-ahead a workgroup spent most of every slice
  // waiting for its loads (3.5 us per slice, matrix pipes 36 % busy with two workgroups per CU)
  float4 a_st[2][AP][2];
  u32x4 w_st[2][3][2];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 ss2[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) ss2[p] = f32x2{0.f, 0.f};
  // uniform base (SGPRs, advanced with k0) + one constant 32-bit byte offset per thread and pass: four address VGPRs for
  // the twenty loads of the two sets
  unsigned a_offs[AP], w_offs[2];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    int row = m0 + ld_row + 64 * p;
    row = row < gM ? row : gM - 1;                         // clamp: such rows are never stored
    a_offs[p] = (static_cast<unsigned>(row) * static_cast<unsigned>(gLda) + ld_chunk * 8) * 4u;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) w_offs[p] = (static_cast<unsigned>(n0 + ld_row + 64 * p) * static_cast<unsigned>(gK) + ld_chunk * 8) * 2u;
  auto load = [&](int k0, auto set) {
    constexpr int S = decltype(set)::value;
    const char* const ab = reinterpret_cast<const char*>(gA) + static_cast<size_t>(k0) * 4;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const float4* src = reinterpret_cast<const float4*>(ab + a_offs[p]);
      a_st[S][p][0] = src[0];
      a_st[S][p][1] = src[1];
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const char* const wb = reinterpret_cast<const char*>(gW[pl]) + static_cast<size_t>(k0) * 2;
#pragma unroll
      for (int p = 0; p < 2; ++p) w_st[S][pl][p] = *reinterpret_cast<const u32x4*>(wb + w_offs[p]);
    }
  };
  // registers of one set -> the six LDS planes (A split into its three bf16 terms on the way)
  auto stage = [&](auto set) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const float f[8] = {a_st[S][p][0].x, a_st[S][p][0].y, a_st[S][p][0].z, a_st[S][p][0].w,
                          a_st[S][p][1].x, a_st[S][p][1].y, a_st[S][p][1].z, a_st[S][p][1].w};
      if constexpr (NORM) {
        // explicit FMA chains in a fixed order (even and odd elements, as register pairs: v_pk_fma_f32 on the pairs the
        // loads deliver -- a scalar chain per pass gets paired ACROSS the passes and shuffles freshly loaded registers):
        // a row's statistics do not depend on where in a tile the row sits
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const f32x2 v = {f[i], f[i + 1]};
          ss2[p] = __builtin_elementwise_fma(v, v, ss2[p]);
        }
      }
      u32x4 h, m, l;
      split3_bf16(f, &h, &m, &l);
      const int at = (ld_row + 64 * p) * ROWE + ld_chunk * 8;
      *reinterpret_cast<u32x4*>(&As[0][at]) = h;
      *reinterpret_cast<u32x4*>(&As[1][at]) = m;
      *reinterpret_cast<u32x4*>(&As[2][at]) = l;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int at = (ld_row + 64 * p) * ROWE + ld_chunk * 8;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(&Bs[pl][at]) = w_st[S][pl][p];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag_row = lane & 15, frag_g = lane >> 4;
  const int a_off = (wm * (BM / 2) + frag_row) * ROWE + frag_g * 8, b_off = (wn * 64 + frag_row) * ROWE + frag_g * 8;
  // the 96 MFMAs of the slice in LDS.  The wave's four column fragments go in two halves so that the second set of
  // staging registers fits (B fragments of two columns live at a time; the A fragments are read once per half)
  auto mfma_slice = [&]() {
#pragma unroll
    for (int jh = 0; jh < FN; jh += 2) {
      u32x4 bf[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[pl][j] = *reinterpret_cast<const u32x4*>(&Bs[pl][b_off + (jh + j) * 16 * ROWE]);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        u32x4 af[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const u32x4*>(&As[pl][a_off + i * 16 * ROWE]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {                      // smallest terms first
          mfma_chunk<__bf16>(af[1], bf[1][j], acc[i][jh + j]);    // mid . mid
          mfma_chunk<__bf16>(af[0], bf[2][j], acc[i][jh + j]);    // hi  . lo
          mfma_chunk<__bf16>(af[2], bf[0][j], acc[i][jh + j]);    // lo  . hi
          mfma_chunk<__bf16>(af[0], bf[1][j], acc[i][jh + j]);    // hi  . mid
          mfma_chunk<__bf16>(af[1], bf[0][j], acc[i][jh + j]);    // mid . hi
          mfma_chunk<__bf16>(af[0], bf[0][j], acc[i][jh + j]);    // hi  . hi
        }
      }
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  // K is a multiple of 2 BK (launch_gemm_x6): slices go in pairs, and the steady-state loop reloads BOTH sets
  // unconditionally -- with the reload under an `if` the compiler has to assume the path without newer loads and
  // waits for every outstanding load (vmcnt 0) before it touches a set, which is one slice of look-ahead again
  load(0, Set0{});
  __builtin_amdgcn_sched_barrier(0);                       // set 0's loads are all older than set 1's (counted vmcnt waits)
  load(BK, Set1{});
  __builtin_amdgcn_sched_barrier(0);
  int k0 = 0;
  // (a fence behind each reload that vector-ALU and vector-memory instructions may not cross: the split of the OTHER set,
  // hoisted in front of part of this reload, would wait for that part -- a fresh memory round trip in the middle of the
  // slice; SALU, MFMA and LDS instructions move freely)
  constexpr int kNoValuNoVmem = 0x4 | 0x8 | 0x80 | 0x100 | 0x200;
  for (; k0 + 2 * BK < gK; k0 += 2 * BK) {
    __syncthreads();                                       // every wave is done reading the previous slice
    stage(Set0{});
    __syncthreads();
    load(k0 + 2 * BK, Set0{});                             // two slices ahead, under this slice's and the next one's MFMAs
    __builtin_amdgcn_sched_barrier(kNoValuNoVmem);
    mfma_slice();
    __syncthreads();
    stage(Set1{});
    __syncthreads();
    load(k0 + 3 * BK, Set1{});
    __builtin_amdgcn_sched_barrier(kNoValuNoVmem);
    mfma_slice();
  }
  __syncthreads();
  stage(Set0{});
  __syncthreads();
  mfma_slice();
  __syncthreads();
  stage(Set1{});
  __syncthreads();
  mfma_slice();
  if constexpr (NORM) {
    // the four lanes that streamed one tile row: their partial sums of squares meet on the DPP network (xor 1, then xor 8)
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      float v = ss2[p][0] + ss2[p][1];                     // lanes t, t ^ 1, t ^ 8, t ^ 9 streamed one tile row
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
      if (ld_chunk == 0) ss_part[ld_row + 64 * p] = v;
    }
    __syncthreads();
  }
  auto row_rs = [&](int lrow) -> float {
    if constexpr (!NORM) return 1.f;
    return rsqrtf(ss_part[lrow] / static_cast<float>(gK) + 1e-6f);
  };
  const EpiCtx ec{g.out, g.aux, gM, gN, g.ldo, g.seq_len, nullptr, nullptr, nullptr, 0, 0, static_cast<const float*>(g.out)};
  const float nopre[FM][FN][4] = {};
  gemm_epilogue<float, EPI, FM, FN, false>(acc, wm, wn, lane, m0, n0, ec, row_rs, nopre);
}

int launch_gemm_x6(const GemmArgs& g, const void* Wm, const void* Wl, bool norm, int epi, hipStream_t s) {
  if (g.M <= 0 || g.N % 128 || g.K % 64 || g.K < 64 || !g.A || !g.Wt || !Wm || !Wl || !g.out)
    return mt3::fail(MT3_ERR_INVALID, "gemm_x6: bad shape or null pointer");
  // 32-bit byte offsets from the operand bases (kernel: a_offs / w_offs)
  if (static_cast<size_t>(g.M) * g.lda * 4 >= (1ull << 32) || static_cast<size_t>(g.N) * g.K * 2 >= (1ull << 32))
    return mt3::fail(MT3_ERR_INVALID, "gemm_x6: operand larger than 4 GB");
  if (epi == MT3_EPI_POS && (!g.aux || g.seq_len <= 0)) return mt3::fail(MT3_ERR_INVALID, "gemm_x6: POS needs aux / seq_len");
  if (epi == MT3_EPI_HEADS && (g.seq_len <= 0 || g.M % g.seq_len)) return mt3::fail(MT3_ERR_INVALID, "gemm_x6: HEADS needs M = B*T");
  const dim3 grid(((g.M + 127) / 128) * (g.N / 128)), block(256);
  const __bf16 *m = static_cast<const __bf16*>(Wm), *l = static_cast<const __bf16*>(Wl);
#define MT3_X6(NORM, EPI) hipLaunchKernelGGL((gemm_x6_kernel<NORM, EPI, 128>), grid, block, 0, s, g, m, l)
  if (norm && epi == MT3_EPI_STORE) MT3_X6(true, MT3_EPI_STORE);
  else if (norm && epi == MT3_EPI_GEGLU) MT3_X6(true, MT3_EPI_GEGLU);
  else if (!norm && epi == MT3_EPI_RESID) MT3_X6(false, MT3_EPI_RESID);
  else if (!norm && epi == MT3_EPI_POS) MT3_X6(false, MT3_EPI_POS);
  else if (!norm && epi == MT3_EPI_HEADS) MT3_X6(false, MT3_EPI_HEADS);
  else return mt3::fail(MT3_ERR_INVALID, "gemm_x6: unsupported (norm, epilogue) combination");
#undef MT3_X6
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// ------------------------------------------------------------------ encoder-sized tile, LDS-DMA staged (bf16)
// 128x128x32 tile, 2x2 waves of 64x64 (4x4 MFMA fragments), both operands bf16 in memory.  The encoder GEMMs have
// SHORT K (384 .. 1024) and operands that come from L2 / MALL at 1-2 us: with one K slice of look-ahead (the
// register-staged tile, and this kernel's first form) every K step costs a full memory latency and the matrix pipe
// idles 80 % of the time (measured: 27 us per 128x128x512 tile against 1.7 us of MFMA work).  So the staging is an
// LDS-DMA RING: `global_load_lds_dwordx4` moves 64 lanes x 16 B = 16 tile rows x 64 B straight into LDS (no staging
// VGPRs, no ds_write pass), 4 stages of 16 KB, THREE K slices in flight per workgroup, counted `s_waitcnt vmcnt(N)`
// (never 0 in steady state) and a raw `s_barrier` so that the DMAs stay in flight across the barrier; two
// workgroups per CU overlap one's prologue / epilogue with the other's K loop.
// A DMA's LDS image is lane-linear, so rows cannot be padded; bank conflicts are removed by an XOR swizzle applied on
// the SOURCE side: lane (row r, slot c) of a DMA fetches global chunk c ^ ((r >> 2) & 3) of its row, and a fragment
// read of logical chunk q of row r looks at slot q ^ ((r >> 2) & 3) -- the 16 rows of a ds_read_b128 lane group
// then cover all 16 sixteen-byte bank slots.  The fused RMSNorm comes from the producer's partial sums of squares
// (norm 2), folded after the K loop.
// BM_ = 256 (round 3; STORE / GEGLU / HEADS epilogues): a wave owns 128 x 64 outputs (8 x 4 fragments), so a K slice costs
// 12 fragment reads for 32 MFMAs instead of 8 for 16 and the A panel is shared by twice the rows: the 128 x 128 tile's
// K loop alone ran at 55 % of the matrix peak with LDS reads and MFMAs both at ~100 % of their own pipes.
// (Round 3 also built fragment double-buffering for the 128-row tile -- four ring stages, the fragment reads of slice
// t + 1 issued before the MFMAs of slice t: 64 KB of ring and 180 VGPRs leave two workgroups per CU instead of three and
// the encoder ran 6.17 against 5.34 ms; removed in round 4)
template <int EPI, int NPV, int BM_ = 128>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs g) {
  using CT = __bf16;
  constexpr int BM = BM_, BN = 128, BK = MT3_GLDS_BK, FM = BM / 32, FN = 4;
  static_assert(BM == 128 || (BM == 256 && EPI != MT3_EPI_RESID && MT3_GLDS_BK == 32), "tile height");
  constexpr int ROWB = BK * 2;                       // bytes per tile row: 64 (4 chunks) or 128 (8 chunks)
  constexpr int CPROW = ROWB / 16;                   // 16-byte chunks per row
  constexpr int RPP = 64 / CPROW;                    // rows per 1 KB DMA piece: 16 or 8
  constexpr int KSH = BK == 32 ? 2 : 1;              // swizzle key of row r: (r >> KSH) & (CPROW - 1)
  constexpr int STAGE_B = (BM + BN) * ROWB;          // 16 / 32 KB per stage: A rows, then W rows
  constexpr int NS = MT3_GLDS_NS, DEPTH = NS - 1;    // ring stages / K slices in flight
  constexpr int PPW = (BM + BN) / RPP / 4;           // 1 KB pieces per wave per stage: 4 (8 with 128-byte rows), 6 at BM = 256
  static_assert(BK == 32 || BK == 64, "K slice");
  // ONE shared object (a second one makes hipcc drain the DMA queue before every k-step's first ds_read)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE_B + BM * 4];
  float* const rs_x = reinterpret_cast<float*>(smem + NS * STAGE_B);

  const void* const gA = g.A;
  const void* const gW = g.Wt;
  const int gM = g.M, gN = g.N, gK = g.K, gLda = g.lda;
  const float* const gAss = g.a_ss;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = gN / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

  // norm 2: request this thread's tile row's partial sums now, fold them after the K loop
  // (unconditional loads from always-valid addresses: a per-element "load or zero" select makes hipcc branch around
  // every load and wait for each one)
  // (the RESID epilogue never carries a row scale: no partial sums, their registers go to its residual prefetch)
  constexpr bool kMayScale = EPI != MT3_EPI_RESID;
  float4 pv[NPV];
  const bool scale_rows = kMayScale && gAss != nullptr && tid < BM;
  const int npv = gAss ? (gK >> 6) : 1;
  if constexpr (kMayScale) {
    const int prow = m0 + (tid & (BM - 1)) < gM ? m0 + (tid & (BM - 1)) : gM - 1;
    const float4* p4 = gAss ? reinterpret_cast<const float4*>(gAss + static_cast<size_t>(prow) * (gK >> 4))
                            : reinterpret_cast<const float4*>(gW);
#pragma unroll
    for (int u = 0; u < NPV; ++u) pv[u] = p4[u < npv ? u : npv - 1];
  }

  // ---- DMA plan: a stage is (BM + BN) / RPP one-KB pieces: the A rows, then the W rows; wave w brings pieces
  // w * PPW .. w * PPW + PPW - 1; lane i of a piece: row = RPP * piece + i / CPROW (of the A | W row stack), slot = i % CPROW.
  const unsigned char* src[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int rr = (wave * PPW + j) * RPP + lane / CPROW;                // row inside the A | W stack
    const bool is_a = rr < BM;                                           // (piece-uniform: BM is a multiple of RPP)
    const int r = is_a ? rr : rr - BM;                                   // row inside the operand tile
    const int chunk = (lane % CPROW) ^ ((r >> KSH) & (CPROW - 1));
    if (is_a) {
      int row = m0 + r;
      row = row < gM ? row : gM - 1;                                     // clamp: such rows are never stored
      src[j] = static_cast<const unsigned char*>(gA) + (static_cast<size_t>(row) * gLda + chunk * 8) * 2;
    } else {
      src[j] = static_cast<const unsigned char*>(gW) + (static_cast<size_t>(n0 + r) * gK + chunk * 8) * 2;
    }
  }
  const int piece0 = wave * PPW;
  auto issue = [&](int kt, int stage) {
    const int dst0 = __builtin_amdgcn_readfirstlane(stage * STAGE_B + piece0 * 1024);
#pragma unroll
    for (int j = 0; j < PPW; ++j)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src[j] + static_cast<size_t>(kt) * ROWB),   // C-style: the
          (__attribute__((address_space(3))) void*)(smem + dst0 + j * 1024), 16, 0, 0);               // only legal cast
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_g = lane >> 4;
  // byte offsets of this lane's fragment pieces inside a stage: logical chunk 4 kk + frag_g of row frag_row (+ 16 i)
  const int key = (frag_row >> KSH) & (CPROW - 1);
  const int a_off = (wm * (BM / 2) + frag_row) * ROWB, b_off = BM * ROWB + (wn * 64 + frag_row) * ROWB;

  const int KT = gK / BK;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < KT) issue(d, d);
  for (int t = 0; t < KT; ++t) {
    // slice t has landed once at most the (<= DEPTH - 1) younger slices' DMAs of this wave are still outstanding
    const int ahead = KT - 1 - t < DEPTH - 1 ? KT - 1 - t : DEPTH - 1;
    if (DEPTH > 2 && ahead >= 2) {
      if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (PPW == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (DEPTH > 1 && ahead == 1) {
      if constexpr (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // raw barrier (no fence: a __syncthreads() here would drain the DMA queue): after it every wave's pieces of
    // slice t are in LDS, and nobody reads stage (t - 1) % NS any more -- the stage slice t + DEPTH goes to
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!(MT3_GLDS_PROBE & 2) && t + DEPTH < KT) issue(t + DEPTH, (t + DEPTH) % NS);
    if (MT3_GLDS_PROBE & 1) continue;
    const unsigned char* st = smem + (t % NS) * STAGE_B;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int so = ((4 * kk + frag_g) ^ key) * 16;
      u32x4 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const u32x4*>(st + a_off + i * 16 * ROWB + so);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(st + b_off + j * 16 * ROWB + so);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mfma_chunk<CT>(af[i], bf[j], acc[i][j]);
    }
  }
  if constexpr (kMayScale) {
    if (scale_rows) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < NPV; ++u) {
        const float4 v = u < npv ? pv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        t = (((t + v.x) + v.y) + v.z) + v.w;                                                // fixed order per row
      }
      rs_x[tid] = rsqrtf(t / static_cast<float>(gK) + 1e-6f);
    }
  }
  if ((MT3_GLDS_PROBE & 4) && gM > 0) return;
  // ---- epilogue through LDS.  The C fragments (lane = 4 rows x 1 column) would reach memory as 32/64-byte pieces
  // of many rows per instruction; the short-K encoder GEMMs move about as many output as operand bytes, so the tile
  // is first transposed through the (now idle) ring, 64 rows at a time (a [64][128] f32 image = 32 KB, the rows of
  // the wave pair wm = 0, then wm = 1): 16-float blocks of a row are XOR-permuted by (row >> 2) & 3 so that the four
  // lane groups of a fragment write hit disjoint banks; then every thread walks rows with float4s -- a wave
  // instruction covers two whole 512-byte tile rows.
  float* const tile = reinterpret_cast<float*>(smem);
  const bool has_rs = kMayScale && gAss != nullptr;
  auto tile4 = [&](int row, int col) -> float4 {                   // logical (half-local row, col .. col + 3), col % 4 == 0
    return *reinterpret_cast<const float4*>(&tile[row * BN + (col ^ (((row >> 2) & 3) << 4))]);
  };
  // RESID: the residual rows this thread will update (both halves) are requested NOW, into the registers the operand
  // fragments just left, so that their HBM latency runs under the two LDS transpositions instead of after each
  f32x4 xpre[2][8];
  if constexpr (EPI == MT3_EPI_RESID) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int grow = m0 + h * 64 + (tid >> 5) + 8 * p;
        xpre[h][p] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.out) +
                                                     static_cast<size_t>(grow < gM ? grow : gM - 1) * g.ldo + n0 + (tid & 31) * 4);
      }
  }
  // passes of 64 rows: pass hm = rows [64 hm, 64 hm + 64) of the tile = fragments (hm % PW) * 4 .. + 3 of the waves with
  // wm == hm / PW (PW = passes per wave row: 1 at BM = 128, 2 at BM = 256); the fragment index must be static
  constexpr int PW = FM / 4;
#pragma unroll 1
  for (int hw = 0; hw < 2; ++hw)
#pragma unroll
  for (int hp = 0; hp < PW; ++hp) {
    const int hm = hw * PW + hp;
    __syncthreads();                     // the ring / the previous pass's image is no longer read (rs_x is visible)
    if (wm == hw) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + frag_g * 4 + r;                // (row >> 2) & 3 == frag_g
            const int col = (wn * 64 + j * 16 + frag_row) ^ (frag_g << 4);
            tile[row * BN + col] = acc[hp * 4 + i][j][r];
          }
    }
    __syncthreads();
    const int mh = m0 + hm * 64;                                     // first global row of this pass
    if constexpr (EPI == MT3_EPI_GEGLU) {
      // tile columns [32q, 32q + 16) = gate, [32q + 16, 32q + 32) = linear of hidden units (n0 >> 1) + 16q + 0..15
      CT* const out = static_cast<CT*>(g.out);
      const int ldo = g.ldo, u4 = (tid & 15) * 4, q = u4 >> 4, s4 = u4 & 15;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = (tid >> 4) + 16 * p;
        if (mh + row >= gM) continue;
        const float rs = has_rs ? rs_x[hm * 64 + row] : 1.f;
        const float4 ga = tile4(row, 32 * q + s4), li = tile4(row, 32 * q + 16 + s4);
        u32x2 pk;
        pk.x = pack_bf16x2(gelu_tanh_fast(ga.x * rs) * (li.x * rs), gelu_tanh_fast(ga.y * rs) * (li.y * rs));
        pk.y = pack_bf16x2(gelu_tanh_fast(ga.z * rs) * (li.z * rs), gelu_tanh_fast(ga.w * rs) * (li.w * rs));
        MT3_GLDS_ST(pk, reinterpret_cast<u32x2*>(out + static_cast<size_t>(mh + row) * ldo + (n0 >> 1) + u4));
      }
    } else {
      const int c4 = (tid & 31) * 4;
      constexpr int kRowUnroll = EPI == MT3_EPI_RESID ? 8 : 4;      // RESID: p must be static (xpre lives in registers)
#pragma unroll kRowUnroll
      for (int p = 0; p < 8; ++p) {
        const int row = (tid >> 5) + 8 * p;
        const int grow = mh + row;
        if (grow >= gM) continue;                                   // (whole half-waves: the quad reductions below stay intact)
        const float rs = has_rs ? rs_x[hm * 64 + row] : 1.f;
        float4 v = tile4(row, c4);
        v.x *= rs, v.y *= rs, v.z *= rs, v.w *= rs;
        const int col = n0 + c4;
        if constexpr (EPI == MT3_EPI_RESID) {
          f32x4* xp = reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + static_cast<size_t>(grow) * g.ldo + col);
          const f32x4 x = hw ? xpre[1][p] : xpre[0][p];   // (plain accesses: non-temporal ones made the read-modify-write 10-20 % slower)
          v.x += x.x, v.y += x.y, v.z += x.z, v.w += x.w;
          *xp = f32x4{v.x, v.y, v.z, v.w};
          if (g.out_ct) {
            float t = __builtin_fmaf(v.w, v.w, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
            t = quad_sum(t);                                        // the 4 lanes of a quad hold one 16-column group
            if ((tid & 3) == 0) g.out_ss[static_cast<size_t>(grow) * (gN >> 4) + (col >> 4)] = t;
            u32x2 pk;
            pk.x = pack_bf16x2(v.x, v.y);
            pk.y = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<u32x2*>(static_cast<CT*>(g.out_ct) + static_cast<size_t>(grow) * g.ldo + col) = pk;
          }
        } else if constexpr (EPI == MT3_EPI_F32) {
          MT3_GLDS_ST((f32x4{v.x, v.y, v.z, v.w}),
                      reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + static_cast<size_t>(grow) * g.ldo + col));
        } else {
          u32x2 pk;
          pk.x = pack_bf16x2(v.x, v.y);
          pk.y = pack_bf16x2(v.z, v.w);
          size_t dst;
          if constexpr (EPI == MT3_EPI_HEADS) {   // col = kv*H*64 + h*64 + d, row = b*T + t  ->  [kv][b][h][t][d]
            const int hd = gN >> 1, seq = g.seq_len;
            const int kv = col / hd, hh = (col % hd) >> 6, d = col & 63;
            const int bb = grow / seq, tt = grow % seq, H = hd >> 6, B = gM / seq;
            dst = ((((static_cast<size_t>(kv) * B + bb) * H + hh) * seq) + tt) * 64 + d;
          } else {
            dst = static_cast<size_t>(grow) * g.ldo + col;
          }
          MT3_GLDS_ST(pk, reinterpret_cast<u32x2*>(static_cast<CT*>(g.out) + dst));
        }
      }
    }
  }
}

template <int EPI>
static int launch_glds(const GemmArgs& g, hipStream_t s) {
  if constexpr (EPI != MT3_EPI_RESID && MT3_GLDS_BK == 32) {
    // 256-row tiles when the launch still fills the chip (two workgroups per CU) twice over with them: measured at
    // B = 256 the tall tile is 5-9 % faster per launch (GEGLU 243 against 260 us, QKV 139 against 153), at B = 64
    // (1.1 rounds of tall tiles) 1 % slower
    const int grid256 = ((g.M + 255) / 256) * (g.N / 128);
    if (grid256 >= 1024) {
      if (g.a_ss && g.K > 512)
        hipLaunchKernelGGL((gemm_glds_kernel<EPI, 16, 256>), dim3(grid256), dim3(256), 0, s, g);
      else
        hipLaunchKernelGGL((gemm_glds_kernel<EPI, 8, 256>), dim3(grid256), dim3(256), 0, s, g);
      MT3_HIP_CHECK(hipGetLastError());
      return MT3_OK;
    }
  }
  const int grid = ((g.M + 127) / 128) * (g.N / 128);
  if (g.a_ss && g.K > 512)
    hipLaunchKernelGGL((gemm_glds_kernel<EPI, 16>), dim3(grid), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_glds_kernel<EPI, 8>), dim3(grid), dim3(256), 0, s, g);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}
// big tile, bf16, both operands bf16 in memory, K a multiple of 64 (<= 1024 with norm 2), N of 128
static bool glds_eligible(const GemmArgs& g, bool a_f32, int norm, int epi) {
  if (a_f32 || norm == 1 || g.K % 64 || g.N % 128 || g.lda % 8) return false;   // (K % 64: norm-2 partials)
  if (norm == 2 && (!g.a_ss || g.K > 1024)) return false;
  return epi == MT3_EPI_STORE || epi == MT3_EPI_RESID || epi == MT3_EPI_GEGLU || epi == MT3_EPI_HEADS ||
         epi == MT3_EPI_F32;
}

// ------------------------------------------------------------------ dispatch
template <typename CT, int BM, int BN, int BK, int WM, int WN, bool A_F32, bool NORM, int EPI, int NPV = 8>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
  if (g.N % BN != 0 || g.K % BK != 0) return mt3::fail(MT3_ERR_INVALID, "gemm: N/K not a multiple of the tile");
  if (g.a_ss && g.K > 64 * NPV) return mt3::fail(MT3_ERR_INVALID, "gemm: K too large for this tile's partial-sum registers");
  const int grid = ((g.M + BM - 1) / BM) * (g.N / BN);
  hipLaunchKernelGGL((gemm_kernel<CT, BM, BN, BK, WM, WN, A_F32, NORM, EPI, NPV>), dim3(grid), dim3(WM * WN * 64), 0, s, g);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

// Tile choice.
//   big   (encoder, M = B*T >= 2048): 128x128 tile, K step = one MFMA K-group, 2x2 waves of 64x64.
//   small (decode, M = B <= a few hundred): the GEMM is latency-bound (weights are L2/MALL resident,
//         ~1 MB), so the tile is 32x32 (32x64 for GEGLU, which pairs fragments inside a wave) to put
//         >= 100 workgroups on the chip, and the K step is as deep as LDS allows (16 K-groups = 512 bf16
//         elements: K = 512 in ONE slice) so that every global load of the block is in flight at once
//         instead of 8-16 dependent load->barrier->MFMA rounds.
// decode-sized tile BMxBN with K slice BK
template <typename CT, int BM, int BN, int BK, bool A_F32, bool NORM, int EPI, int NPV = 8>
static int launch_small(const GemmArgs& g, hipStream_t s) {
  return launch_cfg<CT, BM, BN, BK, 2, 2, A_F32, NORM, EPI, NPV>(g, s);
}

template <typename CT, bool A_F32, bool NORM, int EPI>
static int launch_tile(const GemmArgs& g, bool small, hipStream_t s) {
  constexpr int KG = CTraits<CT>::KGROUP;
  if (small) {
    const bool deep = g.K % (16 * KG) == 0;
    // (measured and removed, DESIGN.md section 3: the two-source fold launch on 32 x 64 tiles -- 25 % fewer operand
    // bytes through a CU's L1 but one workgroup per CU, slower; eight-wave split-K f32 tiles, slower)
    if constexpr (!NORM && !A_F32 && KG == 32 && EPI != MT3_EPI_HEADS && EPI != MT3_EPI_POS) {
      // ismir2022/base.gin shape (emb = heads * 64 = 768): K = 768 as ONE slice too, with room for its 48 partial
      // sums of squares when the rows arrive as the bf16 residual copy (norm 2)
      if (g.K == 24 * KG) {
        // (the one-slice tiles hold 100 / 150 KB of LDS = ONE workgroup per CU: a launch with more workgroups than CUs
        // -- QKV N = 3072: 768, GEGLU N = 4096: 512 -- would run in rounds; those take K in two slices of 384 instead,
        // 51 / 75 KB, three / two workgroups per CU, one round)
        if constexpr (EPI == MT3_EPI_GEGLU) {
          if (((g.M + 31) / 32) * (g.N / 64) > 256)
            return launch_cfg<CT, 32, 64, 12 * KG, 2, 2, A_F32, NORM, EPI, 16>(g, s);
          return launch_cfg<CT, 32, 64, 24 * KG, 2, 2, A_F32, NORM, EPI, 16>(g, s);
        } else {
          if (((g.M + 31) / 32) * (g.N / 32) > 256)
            return launch_cfg<CT, 32, 32, 12 * KG, 2, 2, A_F32, NORM, EPI, 16>(g, s);
          return launch_cfg<CT, 32, 32, 24 * KG, 2, 2, A_F32, NORM, EPI, 16>(g, s);
        }
      }
    }
    if constexpr (KG == 16 && !NORM && !A_F32 &&
                  (EPI == kEpiResidS || EPI == MT3_EPI_GEGLU || EPI == kEpiResidQ || EPI == MT3_EPI_RESID)) {
      // f32 operands, the four dense launches of a decoder layer for a LARGE row group (>= 256 rows per group: engines of
      // 1024+ slots) that runs BESIDE other groups (GemmArgs::concurrent): 64 x 32 tiles with K slices of 128 -- half the
      // workgroups of the 32-row tiles, a 64-row block fetches each weight byte once, and 51 KB of LDS let three
      // workgroups share a CU (four waves stacked along M, 16 x 32 outputs each).  Outputs are BIT-IDENTICAL to the
      // 32-row tiles' (the K order of every output element is the tile-independent MFMA chain;
      // tests/test_gpu_parity_r5.py holds 256-row groups against one stream).  Measured (round 5,
      // profiles/r5_ab_refill_schedule.txt part 9; refilled ragged corpus, f32): 312 rows per group 2,419 -> 2,603
      // audio-s/s (the fold launch alone: 2,532; with slices of 256: 2,471), canonical 1250-row decode 4,759 -> 4,575 ms;
      // 256 rows per group 2,382 -> 2,541; 128 rows per group 2,298 -> 2,225 (slower: hence the threshold); ONE stream of
      // 256 rows alone on the chip 1,178 -> 1,214 ms per decode (slower: hence `concurrent` -- under contention the
      // bytes through the L1s decide, alone the latency of a workgroup does)
      if (g.concurrent && g.M >= 256 && g.K % (8 * KG) == 0) return launch_cfg<CT, 64, 32, 8 * KG, 4, 1, A_F32, NORM, EPI>(g, s);
    }
    if constexpr (EPI == MT3_EPI_GEGLU) {
      if (deep) return launch_small<CT, 32, 64, 16 * KG, A_F32, NORM, EPI>(g, s);
      return launch_cfg<CT, 32, 64, 4 * KG, 2, 2, A_F32, NORM, EPI>(g, s);
    } else {

      if constexpr (!NORM && !A_F32 && KG == 32) {
        // the attention out-projections (K = 384 = 12 K-groups) as ONE slice as well (-1 % of the decode; the
        // same for wo, K = 1024 in 133 KB of LDS, measured slower than its two 512-slices)
        if (g.K == 12 * KG) return launch_cfg<CT, 32, 32, 12 * KG, 2, 2, A_F32, NORM, EPI>(g, s);
      }
      if constexpr (!NORM && !A_F32 && KG == 16) {
        // f32 operands: the attention out-projections (K = 384) as ONE slice (100 KB of LDS: one workgroup per CU,
        // these launches have at most 224) -- every slice of an f32 tile is a dependent 64 KB trip through the CU's L1
        if (g.K == 24 * KG) {
          const int wgs = ((g.M + 31) / 32) * (g.N / 32);
          if (wgs <= 256) return launch_cfg<CT, 32, 32, 24 * KG, 2, 2, A_F32, NORM, EPI>(g, s);
          return launch_small<CT, 32, 32, 12 * KG, A_F32, NORM, EPI>(g, s);
        }
      }
      if (deep) return launch_small<CT, 32, 32, 16 * KG, A_F32, NORM, EPI>(g, s);
      return launch_cfg<CT, 32, 32, 4 * KG, 2, 2, A_F32, NORM, EPI>(g, s);
    }
  }
  if constexpr (!NORM && !A_F32)
    if (g.a_ss && g.K > 512) return launch_cfg<CT, 128, 128, KG, 2, 2, A_F32, NORM, EPI, 16>(g, s);
  return launch_cfg<CT, 128, 128, KG, 2, 2, A_F32, NORM, EPI>(g, s);
}

template <typename CT>
static int launch_typed(const GemmArgs& g, bool a_f32, int norm, int epi, bool small, hipStream_t s) {
  // Only the combinations the engine uses are instantiated.
  if (norm == 2) {
    if (a_f32 || !g.a_ss || g.K % 64 || (small ? (g.K > 512 && g.K != 768) : g.K > 1024))
      return mt3::fail(MT3_ERR_INVALID, "gemm: norm 2 needs a compute-type A, a_ss and K = 64n <= 512 or 768 (decode-sized "
                                        "tile) / <= 1024 (encoder-sized tiles)");
    switch (epi) {
      case MT3_EPI_STORE: return launch_tile<CT, false, false, MT3_EPI_STORE>(g, small, s);
      case MT3_EPI_GEGLU: return launch_tile<CT, false, false, MT3_EPI_GEGLU>(g, small, s);
      case MT3_EPI_F32: return launch_tile<CT, false, false, MT3_EPI_F32>(g, small, s);
      case kEpiStoreQ:
        if (small) return launch_tile<CT, false, false, kEpiStoreQ>(g, small, s);
        break;
      default: break;
    }
    return mt3::fail(MT3_ERR_INVALID, "gemm: unsupported epilogue for norm 2");
  }
  if (g.a_ss) return mt3::fail(MT3_ERR_INVALID, "gemm: a_ss without norm 2");
  if (norm) {
    if (!a_f32) return mt3::fail(MT3_ERR_INVALID, "gemm: norm requires an f32 A operand");
    switch (epi) {
      case MT3_EPI_STORE: return launch_tile<CT, true, true, MT3_EPI_STORE>(g, small, s);
      case MT3_EPI_GEGLU: return launch_tile<CT, true, true, MT3_EPI_GEGLU>(g, small, s);
      case MT3_EPI_F32: return launch_tile<CT, true, true, MT3_EPI_F32>(g, small, s);
      default: break;
    }
  } else if (a_f32) {
    switch (epi) {
      case MT3_EPI_POS: return launch_tile<CT, true, false, MT3_EPI_POS>(g, small, s);
      case MT3_EPI_F32: return launch_tile<CT, true, false, MT3_EPI_F32>(g, small, s);
      case MT3_EPI_STORE: return launch_tile<CT, true, false, MT3_EPI_STORE>(g, small, s);
      default: break;
    }
  } else {
    switch (epi) {
      case MT3_EPI_RESID: return launch_tile<CT, false, false, MT3_EPI_RESID>(g, small, s);
      case kEpiResidQ:
        if (small) return launch_tile<CT, false, false, kEpiResidQ>(g, small, s);
        break;
      case kEpiResidS:
        if (small) return launch_tile<CT, false, false, kEpiResidS>(g, small, s);
        break;
      case MT3_EPI_HEADS: return launch_tile<CT, false, false, MT3_EPI_HEADS>(g, small, s);
      case MT3_EPI_STORE: return launch_tile<CT, false, false, MT3_EPI_STORE>(g, small, s);
      case MT3_EPI_F32: return launch_tile<CT, false, false, MT3_EPI_F32>(g, small, s);
      default: break;
    }
  }
  return mt3::fail(MT3_ERR_INVALID, "gemm: unsupported (a_is_f32, norm, epilogue) combination");
}

int launch_gemm(int dtype, const GemmArgs& g, bool a_f32, int norm, int epi, bool small, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.Wt || !g.out)
    return mt3::fail(MT3_ERR_INVALID, "gemm: bad shape or null pointer");
  if (epi == MT3_EPI_POS && (!g.aux || g.seq_len <= 0)) return mt3::fail(MT3_ERR_INVALID, "gemm: POS needs aux/seq_len");
  if ((epi == kEpiStoreQ || epi == kEpiResidQ || epi == kEpiResidS) &&
      (!g.out2 || g.n_split <= 0 || g.n_split >= g.N || g.n_split % 64))
    return mt3::fail(MT3_ERR_INVALID, "gemm: split epilogue needs out2 and 0 < n_split < N, n_split a multiple of 64");
  if (epi == kEpiResidS && (!g.A2 || g.k_split <= 0 || g.k_split >= g.K || g.k_split % 512 || (g.K - g.k_split) % 512))
    return mt3::fail(MT3_ERR_INVALID, "gemm: the two-source epilogue needs A2 and slice-aligned 0 < k_split < K");
  if (epi == MT3_EPI_HEADS && (g.seq_len <= 0 || g.M % g.seq_len != 0 || g.N % 128 != 0))
    return mt3::fail(MT3_ERR_INVALID, "gemm: HEADS needs M = B*T and N = 2*H*64");
  if (dtype == MT3_BF16 && !small && glds_eligible(g, a_f32, norm, epi)) {
    switch (epi) {
      case MT3_EPI_STORE: return launch_glds<MT3_EPI_STORE>(g, s);
      case MT3_EPI_RESID: return launch_glds<MT3_EPI_RESID>(g, s);
      case MT3_EPI_GEGLU: return launch_glds<MT3_EPI_GEGLU>(g, s);
      case MT3_EPI_HEADS: return launch_glds<MT3_EPI_HEADS>(g, s);
      default: return launch_glds<MT3_EPI_F32>(g, s);
    }
  }
  GemmArgs gg = g;
  // an XCD's 4 MB L2 sees every weight column when tiles are dealt by row block; above ~3 MB per matrix (base.gin
  // shape, f32 operands) dealing by column slice keeps an eighth of it there instead (measured: base.gin decode
  // without attention 644 -> 603 us per step; MT3 shape bf16, weights <= 2 MB: 198 -> 203, hence the threshold)
  const size_t w_bytes = static_cast<size_t>(g.N) * g.K * (dtype == MT3_BF16 ? 2 : 4);
  gg.n_major = small && w_bytes > (3u << 20) ? 1 : 0;
  if (dtype == MT3_BF16) return launch_typed<__bf16>(gg, a_f32, norm, epi, small, s);
  if (dtype == MT3_F32) return launch_typed<float>(gg, a_f32, norm, epi, small, s);
  return mt3::fail(MT3_ERR_INVALID, "gemm: unknown dtype");
}

}  // namespace mt3k

extern "C" int mt3_op_gemm_ex(int32_t dtype, const void* d_A, int32_t a_is_f32, int32_t norm, const void* d_Wt,
                              void* d_out, int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* d_aux,
                              int32_t seq_len, int32_t small, const float* d_a_ss, void* d_out_ct, float* d_out_ss,
                              void* stream) {
  if (norm < 0 || norm > 2) return mt3::fail(MT3_ERR_INVALID, "gemm: norm must be 0, 1 or 2");
  mt3k::GemmArgs g{};
  g.A = d_A;
  g.Wt = d_Wt;
  g.out = d_out;
  g.aux = d_aux;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldo = epilogue == MT3_EPI_GEGLU ? N / 2 : N;
  g.seq_len = seq_len;
  g.a_ss = norm == 2 ? d_a_ss : nullptr;
  g.out_ct = d_out_ct;
  g.out_ss = d_out_ss;
  if (dtype == MT3_BF16 ? (d_out_ct != nullptr) != (d_out_ss != nullptr) : d_out_ct != nullptr)
    return mt3::fail(MT3_ERR_INVALID, "gemm: bf16: out_ct and out_ss come together; f32: out_ss alone (the rows are their "
                                      "own compute-type copy)");
  return mt3k::launch_gemm(dtype, g, a_is_f32 != 0, norm, epilogue, small != 0, static_cast<hipStream_t>(stream));
}

extern "C" int mt3_op_gemm(int32_t dtype, const void* d_A, int32_t a_is_f32, int32_t norm, const void* d_Wt,
                           void* d_out, int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* d_aux,
                           int32_t seq_len, int32_t small, void* stream) {
  mt3k::GemmArgs g{};
  g.A = d_A;
  g.Wt = d_Wt;
  g.out = d_out;
  g.aux = d_aux;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldo = epilogue == MT3_EPI_GEGLU ? N / 2 : N;
  g.seq_len = seq_len;
  return mt3k::launch_gemm(dtype, g, a_is_f32 != 0, norm != 0 ? 1 : 0, epilogue, small != 0, static_cast<hipStream_t>(stream));
}

#if MT3_EXP & 32
// experiment build only: copy the phase accumulators out (16 x 8 uint64) and clear them
extern "C" int mt3_exp_gemm_phases(unsigned long long* h_out) {
  if (hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_ph), sizeof(unsigned long long) * 16 * 8) != hipSuccess) return 1;
  static const unsigned long long zeros[16 * 8] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ph), zeros, sizeof(zeros)) != hipSuccess;
}
#endif

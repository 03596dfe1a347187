// MXFP8 dense path of the encoder (BASELINE configs[4]: "fp8 MFMA path"): OCP e4m3fn elements with one E8M0
// power-of-two scale per 32 consecutive K elements (the OCP microscaling layout), multiplied by gfx950's block-scaled
// matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 per instruction, twice the bf16 rate), f32
// accumulation.  There is no counterpart in the reference (its Dense layers are f32, layers.py:311-360): parity is a
// stated bound against the f32 oracle plus exactness against a numpy emulation of this quantisation (tests).
//
// Operand map of the scaled MFMA (A 16 x 128, B 128 x 16), measured with tools/micro/mfma_scale_check.hip because the
// guides do not give it: lane l feeds row / column l & 15; with g = l >> 4 its first four VGPRs are k = 16 g .. 16 g + 15
// and its last four k = 64 + 16 g .. 64 + 16 g + 15 (NOT 32 consecutive k); its scale VGPR (byte op_sel) is the E8M0
// of the CONSECUTIVE block k = 32 g .. 32 g + 31 of that row -- so a lane's scale belongs to other lanes' data, and
// the memory format (one scale per 32 consecutive K) needs no re-blocking.  C/D as every other 16x16 MFMA: row
// (l >> 4) * 4 + r, column l & 15.
//
// Tile: 128 x 128 outputs, K step 128 (one 128-byte line per operand row and step -- whole-line DMAs), 2 x 2 waves of
// 64 x 64, LDS-DMA ring as in gemm.hip's bf16 tile (global_load_lds, counted vmcnt, raw s_barrier): per stage 16 KB of
// A rows, 16 KB of W rows and 2 x 512 B of scale dwords (a dword = the four block scales of one K step of one row).
// Bank conflicts: the DMA image is lane-linear (rows of 128 B cannot be padded), so the 16-byte slots of a row are
// XOR-permuted on the SOURCE side with key f(r) = r & 6; a lane's fragment is logical chunks g and g + 4, and with
// this key the 16 lanes of every ds_read_b128 service group touch 16 different 16-byte bank slots.
#include "common.h"
#include "device.h"
#include "kernels.h"
#include "mx8.h"

#ifndef MT3_MX8_NS
#define MT3_MX8_NS 2          // ring stages: 2 (66 KB, two workgroups per CU) or 3 (99 KB, one)
#endif
#ifndef MT3_MX8_PROBE
#define MT3_MX8_PROBE 0       // tools/micro/mx8_probe.hip: 1 no fragment reads / MFMAs, 2 no DMA after the prologue, 4 no epilogue
#endif

namespace mt3k {
namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)));
}
// max over the 8 lanes of an aligned 8-lane group (every lane gets it): xor 1, xor 2, mirrored half-row
__device__ __forceinline__ float oct_max(float v) {
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  return dpp_max<0x141>(v);
}
// E8M0 byte and reciprocal of the block scale 2^(floor(log2 amax) - 7): amax / scale in [128, 256), e4m3fn tops out
// at 448, nothing saturates.  amax = 0 (or below 2^-120): byte 0 = 2^-127.
__device__ __forceinline__ void mx8_block_scale(float amax, unsigned* byte, float* inv) {
  const unsigned e = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
  *byte = e > 7u ? e - 7u : 0u;
  const unsigned ie = 261u - e;                          // exponent field of 2^(7 - (e - 127))
  *inv = __builtin_bit_cast(float, (ie > 254u ? 254u : ie) << 23);
}
__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return static_cast<unsigned>(w);
}
// the four values of a lane -> its dword of e4m3 and (lanes with (lane & 15) == 0 of every 16-lane row) the two block
// scales of that row as one ushort; 8 consecutive lanes = one 32-element block
__device__ __forceinline__ void mx8_quantize4(float a, float b, float c, float d, unsigned* q, unsigned* sc_pair) {
  const float am = oct_max(fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
  unsigned byte;
  float inv;
  mx8_block_scale(am, &byte, &inv);
  *q = pack_e4m3x4(a * inv, b * inv, c * inv, d * inv);
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(byte), 0x108, 0xF, 0xF, true));  // row_shl:8
  *sc_pair = byte | (hi << 8);
}

// ------------------------------------------------------------------ rows -> MXFP8 (+ the partial sums of squares)
template <typename IN, bool WITH_SS>
__global__ __launch_bounds__(256) void mx8_quantize_kernel(const IN* __restrict__ x, uint8_t* __restrict__ q,
                                                           uint8_t* __restrict__ sc, float* __restrict__ ss, long n4) {
  // lane = 4 consecutive elements; K % 64 == 0, so a 16-lane DPP row never straddles two matrix rows
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  const long at = (i < n4 ? i : n4 - 1) * 4;
  float v[4];
  if constexpr (sizeof(IN) == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(x + at);
    v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
  } else {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(x + at);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = static_cast<float>(t[j]);
  }
  unsigned qd, pair;
  mx8_quantize4(v[0], v[1], v[2], v[3], &qd, &pair);
  if constexpr (WITH_SS) {
    const float t = quad_sum(__builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0]))));
    if (i < n4 && (threadIdx.x & 3) == 0) ss[at >> 4] = t;
  }
  if (i >= n4) return;
  *reinterpret_cast<unsigned*>(q + at) = qd;
  if ((threadIdx.x & 15) == 0) *reinterpret_cast<unsigned short*>(sc + (at >> 5)) = static_cast<unsigned short>(pair);
}

// ------------------------------------------------------------------ the GEMM
template <int EPI, int NPV>
__global__ __launch_bounds__(256) void gemm_mx8_kernel(Mx8Args g) {
  constexpr int BM = 128, BN = 128, FM = 4, FN = 4;
  constexpr int ROWB = 128;                           // bytes (= K elements) per tile row and stage
  constexpr int TILE_B = BM * ROWB;                   // 16 KB per operand
  constexpr int SC_OFF = 2 * TILE_B;                  // scale dwords: A rows 0..127, then W rows 0..127
  constexpr int STAGE_B = SC_OFF + 1024;              // 33 KB
  constexpr int NS = MT3_MX8_NS, DEPTH = NS - 1;
  constexpr int LPS = 9;                              // DMA instructions per wave and stage: 8 x 1 KB of rows + 256 B of scales
  static_assert(NS == 2 || NS == 3, "ring stages");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * STAGE_B + BM * 4];
  float* const rs_x = reinterpret_cast<float*>(smem + NS * STAGE_B);

  const int gM = g.M, gN = g.N, gK = g.K;
  const float* const gAss = g.a_ss;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = gN / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

  // fused RMSNorm (norm 2): this thread's tile row's partial sums of squares, requested now, folded after the K loop
  constexpr bool kMayScale = EPI != MT3_EPI_RESID;          // (RESID never carries a row scale)
  float4 pv[NPV];
  const bool scale_rows = kMayScale && gAss != nullptr && tid < BM;
  const int npv = gAss ? (gK >> 6) : 1;
  if constexpr (kMayScale) {
    const int prow = m0 + (tid & (BM - 1)) < gM ? m0 + (tid & (BM - 1)) : gM - 1;
    const float4* p4 = gAss ? reinterpret_cast<const float4*>(gAss + static_cast<size_t>(prow) * (gK >> 4))
                            : reinterpret_cast<const float4*>(g.W);
#pragma unroll
    for (int u = 0; u < NPV; ++u) pv[u] = p4[u < npv ? u : npv - 1];
  }

  // RESID: the residual rows this thread will update in the epilogue (both 64-row halves) are requested FIRST, so
  // that the f32 read of the read-modify-write -- 40 % of this launch's HBM bytes -- runs under the K loop
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

  // ---- DMA plan.  Waves 0, 1 bring A rows 0-63 / 64-127, waves 2, 3 W rows; piece j = 8 rows x 128 B; lane i of a
  // piece: row 8 j + i / 8, slot i % 8 <- global chunk slot ^ (row & 6); then one 256-byte piece of scale dwords
  const bool is_a = wave < 2;
  const int half = wave & 1;
  const int ldsc = gK >> 5;                                                // scale bytes per row
  const unsigned char* src[8];
  const unsigned char* src_sc;
  {
    const unsigned char* base = is_a ? g.A : g.W;
    const unsigned char* sbase = is_a ? g.a_sc : g.w_sc;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = half * 64 + j * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (r & 6);
      int row = (is_a ? m0 : n0) + r;
      if (is_a) row = row < gM ? row : gM - 1;                             // clamp: such rows are never stored
      src[j] = base + static_cast<size_t>(row) * gK + chunk * 16;
    }
    int row = (is_a ? m0 : n0) + half * 64 + lane;
    if (is_a) row = row < gM ? row : gM - 1;
    src_sc = sbase + static_cast<size_t>(row) * ldsc;
  }
  const int dst_rows = (is_a ? 0 : TILE_B) + half * 8192;
  const int dst_sc = SC_OFF + (is_a ? 0 : 512) + half * 256;
  auto issue = [&](int kt, int stage) {
    const int s0 = __builtin_amdgcn_readfirstlane(stage * STAGE_B);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + static_cast<size_t>(kt) * ROWB),
                                       (__attribute__((address_space(3))) void*)(smem + s0 + dst_rows + j * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_sc + kt * 4),
                                     (__attribute__((address_space(3))) void*)(smem + s0 + dst_sc), 4, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_g = lane >> 4;
  // logical chunks g (k = 16 g ..) and g + 4 (k = 64 + 16 g ..) of the row; rows i * 16 + frag_row share the key
  const int so0 = (frag_g ^ (frag_row & 6)) * 16, so1 = so0 ^ 64;
  const int a_off = (wm * 64 + frag_row) * ROWB, b_off = TILE_B + (wn * 64 + frag_row) * ROWB;
  const int sa_off = SC_OFF + (wm * 64 + frag_row) * 4, sb_off = SC_OFF + 512 + (wn * 64 + frag_row) * 4;
  const int sh = frag_g * 8;                                               // this lane SUPPLIES block g's scale: byte g of the dword

  const int KT = gK / ROWB;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < KT) issue(d, d);
  for (int t = 0; t < KT; ++t) {
    const int ahead = KT - 1 - t < DEPTH - 1 ? KT - 1 - t : DEPTH - 1;     // younger slices still allowed in flight
    if (DEPTH > 1 && ahead == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(LPS == 9, "the vmcnt immediates above count 9 DMAs per stage");
    // raw barrier (a __syncthreads() would drain the DMA queue): slice t is in LDS for every wave, and stage
    // (t - 1) % NS is free -- slice t + DEPTH goes there
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!(MT3_MX8_PROBE & 2) && t + DEPTH < KT) issue(t + DEPTH, (t + DEPTH) % NS);
    if (MT3_MX8_PROBE & 1) continue;
    const unsigned char* st = smem + (t % NS) * STAGE_B;
    i32x8 af[FM], bf[FN];
    int sa[FM], sb[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(st + a_off + i * 16 * ROWB + so0);
      const u32x4 hi = *reinterpret_cast<const u32x4*>(st + a_off + i * 16 * ROWB + so1);
      af[i] = i32x8{int(lo[0]), int(lo[1]), int(lo[2]), int(lo[3]), int(hi[0]), int(hi[1]), int(hi[2]), int(hi[3])};
      sa[i] = static_cast<int>(*reinterpret_cast<const unsigned*>(st + sa_off + i * 64) >> sh);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const u32x4 lo = *reinterpret_cast<const u32x4*>(st + b_off + j * 16 * ROWB + so0);
      const u32x4 hi = *reinterpret_cast<const u32x4*>(st + b_off + j * 16 * ROWB + so1);
      bf[j] = i32x8{int(lo[0]), int(lo[1]), int(lo[2]), int(lo[3]), int(hi[0]), int(hi[1]), int(hi[2]), int(hi[3])};
      sb[j] = static_cast<int>(*reinterpret_cast<const unsigned*>(st + sb_off + j * 64) >> sh);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, sa[i], 0, sb[j]);
  }
  if constexpr (kMayScale) {
    if (scale_rows) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < NPV; ++u) {
        const float4 v = u < npv ? pv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        t = (((t + v.x) + v.y) + v.z) + v.w;                               // fixed order per row
      }
      rs_x[tid] = rsqrtf(t / static_cast<float>(gK) + 1e-6f);
    }
  }
  if ((MT3_MX8_PROBE & 4) && gM > 0) return;

  // ---- epilogue through LDS, 64 rows at a time (as gemm.hip's bf16 tile): the C fragments are transposed through the
  // idle ring so that every thread then walks rows with float4s and stores whole lines
  float* const tile = reinterpret_cast<float*>(smem);
  const bool has_rs = kMayScale && gAss != nullptr;
  auto tile4 = [&](int row, int col) -> float4 {
    return *reinterpret_cast<const float4*>(&tile[row * BN + (col ^ (((row >> 2) & 3) << 4))]);
  };
#pragma unroll 1
  for (int hm = 0; hm < 2; ++hm) {
    __syncthreads();
    if (wm == hm) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + frag_g * 4 + r;
            const int col = (wn * 64 + j * 16 + frag_row) ^ (frag_g << 4);
            tile[row * BN + col] = acc[i][j][r];
          }
    }
    __syncthreads();
    const int mh = m0 + hm * 64;
    if constexpr (EPI == MT3_EPI_GEGLU) {
      // tile columns [32q, 32q + 16) = gate, [32q + 16, 32q + 32) = linear of hidden units (n0 >> 1) + 16q + 0..15;
      // the output IS the next GEMM's MXFP8 operand: 16 lanes = one tile row's 64 hidden units = two 32-blocks
      const int ldo = g.ldo, u4 = (tid & 15) * 4, q = u4 >> 4, s4 = u4 & 15;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = (tid >> 4) + 16 * p;
        const bool ok = mh + row < gM;
        const int grow = ok ? mh + row : gM - 1;
        const float rs = has_rs ? rs_x[hm * 64 + row] : 1.f;
        const float4 ga = tile4(row, 32 * q + s4), li = tile4(row, 32 * q + 16 + s4);
        const float h0 = gelu_tanh_fast(ga.x * rs) * (li.x * rs), h1 = gelu_tanh_fast(ga.y * rs) * (li.y * rs);
        const float h2 = gelu_tanh_fast(ga.z * rs) * (li.z * rs), h3 = gelu_tanh_fast(ga.w * rs) * (li.w * rs);
        unsigned qd, pair;
        mx8_quantize4(h0, h1, h2, h3, &qd, &pair);
        if (!ok) continue;
        const int col = (n0 >> 1) + u4;
        __builtin_nontemporal_store(qd, reinterpret_cast<unsigned*>(g.out_q + static_cast<size_t>(grow) * ldo + col));
        if ((tid & 15) == 0)
          *reinterpret_cast<unsigned short*>(g.out_sc + static_cast<size_t>(grow) * (ldo >> 5) + (col >> 5)) =
              static_cast<unsigned short>(pair);
      }
    } else {
      const int c4 = (tid & 31) * 4;
      constexpr int kRowUnroll = EPI == MT3_EPI_RESID ? 8 : 4;      // RESID: p must be static (xpre lives in registers)
#pragma unroll kRowUnroll
      for (int p = 0; p < 8; ++p) {
        const int row = (tid >> 5) + 8 * p;
        const bool ok = mh + row < gM;
        const int grow = ok ? mh + row : gM - 1;
        const float rs = has_rs ? rs_x[hm * 64 + row] : 1.f;
        float4 v = tile4(row, c4);
        v.x *= rs, v.y *= rs, v.z *= rs, v.w *= rs;
        const int col = n0 + c4;
        if constexpr (EPI == MT3_EPI_RESID) {
          // x += product; the new rows also leave as the split residual form of this path: per-16-column sums of
          // squares (exact f32, for the next fused RMSNorm) and the MXFP8 copy the next GEMM reads
          f32x4* xp = reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + static_cast<size_t>(grow) * g.ldo + col);
          const f32x4 x = hm ? xpre[1][p] : xpre[0][p];
          v.x += x.x, v.y += x.y, v.z += x.z, v.w += x.w;
          float t = __builtin_fmaf(v.w, v.w, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
          t = quad_sum(t);
          unsigned qd, pair;
          mx8_quantize4(v.x, v.y, v.z, v.w, &qd, &pair);
          if (!ok) continue;
          *xp = f32x4{v.x, v.y, v.z, v.w};
          if ((tid & 3) == 0) g.out_ss[static_cast<size_t>(grow) * (gN >> 4) + (col >> 4)] = t;
          *reinterpret_cast<unsigned*>(g.out_q + static_cast<size_t>(grow) * gN + col) = qd;
          if ((tid & 15) == 0)
            *reinterpret_cast<unsigned short*>(g.out_sc + static_cast<size_t>(grow) * (gN >> 5) + (col >> 5)) =
                static_cast<unsigned short>(pair);
        } else {
          if (!ok) continue;
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
          __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(static_cast<__bf16*>(g.out) + dst));
        }
      }
    }
  }
}

template <int EPI>
int launch_mx8(const Mx8Args& g, hipStream_t s) {
  const int grid = ((g.M + 127) / 128) * (g.N / 128);
  if (g.a_ss && g.K > 512)
    hipLaunchKernelGGL((gemm_mx8_kernel<EPI, 16>), dim3(grid), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_mx8_kernel<EPI, 8>), dim3(grid), dim3(256), 0, s, g);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

}  // namespace

int launch_gemm_mx8(const Mx8Args& g, int epi, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.a_sc || !g.W || !g.w_sc)
    return mt3::fail(MT3_ERR_INVALID, "gemm_mx8: bad shape or null pointer");
  if (g.K % 128 || g.N % 128) return mt3::fail(MT3_ERR_INVALID, "gemm_mx8: K and N must be multiples of 128");
  if (g.a_ss && g.K > 1024) return mt3::fail(MT3_ERR_INVALID, "gemm_mx8: fused RMSNorm needs K <= 1024");
  switch (epi) {
    case MT3_EPI_STORE:
      if (!g.out || g.ldo < g.N) break;
      return launch_mx8<MT3_EPI_STORE>(g, s);
    case MT3_EPI_HEADS:
      if (!g.out || g.seq_len <= 0 || g.M % g.seq_len) break;
      return launch_mx8<MT3_EPI_HEADS>(g, s);
    case MT3_EPI_RESID:
      if (!g.out || !g.out_q || !g.out_sc || !g.out_ss || g.ldo < g.N || g.a_ss) break;
      return launch_mx8<MT3_EPI_RESID>(g, s);
    case MT3_EPI_GEGLU:
      if (!g.out_q || !g.out_sc || g.N % 256 || g.ldo != g.N / 2) break;   // (N / 2 hidden units: whole 128-blocks per row)
      return launch_mx8<MT3_EPI_GEGLU>(g, s);
    default:
      return mt3::fail(MT3_ERR_INVALID, "gemm_mx8: unsupported epilogue");
  }
  return mt3::fail(MT3_ERR_INVALID, "gemm_mx8: outputs missing or inconsistent for this epilogue");
}

int launch_mx8_quantize(const void* in, bool in_f32, int M, int K, uint8_t* q, uint8_t* sc, float* ss, hipStream_t s) {
  if (!in || !q || !sc || M <= 0 || K <= 0 || K % 64) return mt3::fail(MT3_ERR_INVALID, "mx8_quantize: bad arguments (K = 64n)");
  if (ss && !in_f32) return mt3::fail(MT3_ERR_INVALID, "mx8_quantize: sums of squares come with the f32 input only");
  const long n4 = static_cast<long>(M) * K / 4;
  const dim3 grid(static_cast<unsigned>((n4 + 255) / 256));
  if (in_f32 && ss)
    hipLaunchKernelGGL((mx8_quantize_kernel<float, true>), grid, dim3(256), 0, s, static_cast<const float*>(in), q, sc, ss, n4);
  else if (in_f32)
    hipLaunchKernelGGL((mx8_quantize_kernel<float, false>), grid, dim3(256), 0, s, static_cast<const float*>(in), q, sc, ss, n4);
  else
    hipLaunchKernelGGL((mx8_quantize_kernel<__bf16, false>), grid, dim3(256), 0, s, static_cast<const __bf16*>(in), q, sc, ss, n4);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

}  // namespace mt3k

extern "C" int mt3_op_mx8_quantize(const void* d_in, int32_t in_is_f32, int32_t M, int32_t K, uint8_t* d_q,
                                   uint8_t* d_sc, float* d_ss, void* stream) {
  return mt3k::launch_mx8_quantize(d_in, in_is_f32 != 0, M, K, d_q, d_sc, d_ss, static_cast<hipStream_t>(stream));
}

extern "C" int mt3_op_gemm_mx8(const uint8_t* d_A, const uint8_t* d_a_sc, const uint8_t* d_W, const uint8_t* d_w_sc,
                               void* d_out, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t seq_len,
                               const float* d_a_ss, uint8_t* d_out_q, uint8_t* d_out_sc, float* d_out_ss, void* stream) {
  mt3k::Mx8Args g{};
  g.A = d_A;
  g.a_sc = d_a_sc;
  g.W = d_W;
  g.w_sc = d_w_sc;
  g.out = d_out;
  g.M = M;
  g.N = N;
  g.K = K;
  g.ldo = epilogue == MT3_EPI_GEGLU ? N / 2 : N;
  g.seq_len = seq_len;
  g.a_ss = d_a_ss;
  g.out_q = d_out_q;
  g.out_sc = d_out_sc;
  g.out_ss = d_out_ss;
  return mt3k::launch_gemm_mx8(g, epilogue, static_cast<hipStream_t>(stream));
}

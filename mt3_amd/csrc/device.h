// Device-side vocabulary shared by the gfx950 kernels: 16-byte lane chunks and the
// MFMA wrappers for the two compute types.
//
// "Chunk" = the 16 bytes one lane feeds to the matrix pipe for one K-group:
//   bf16: 8 elements -> ONE v_mfma_f32_16x16x32_bf16 (K = 4 lane-groups x 8 = 32)
//   f32 : 4 elements -> FOUR v_mfma_f32_16x16x4_f32  (K = 4 lane-groups x 1, element j each)
// Operand maps (guide: cdna_hip_programming.md section 3):
//   A[i][k]: lane l feeds row i = l & 15, K-group g = l >> 4
//   B[k][n]: lane l feeds col n = l & 15, K-group g = l >> 4
//   C[i][n]: lane l, reg r holds row i = (l >> 4) * 4 + r, col n = l & 15
// Which k a (group, element) pair denotes is irrelevant as long as A and B are
// loaded with the same rule -- every kernel here loads both as "element e of
// lane-group g = k-index g * KPL + e of the current K-group".
#ifndef MT3_DEVICE_H_
#define MT3_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mt3k {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // one 16-byte lane chunk
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename CT>
struct CTraits;
template <>
struct CTraits<__bf16> {
  static constexpr int KPL = 8;       // elements per 16-byte chunk
  static constexpr int KGROUP = 32;   // K covered by one chunk-MFMA
};
template <>
struct CTraits<float> {
  static constexpr int KPL = 4;
  static constexpr int KGROUP = 16;
};

template <typename CT>
__device__ __forceinline__ void mfma_chunk(const u32x4& a, const u32x4& b, f32x4& acc);

template <>
__device__ __forceinline__ void mfma_chunk<__bf16>(const u32x4& a, const u32x4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                0, 0);
}
template <>
__device__ __forceinline__ void mfma_chunk<float>(const u32x4& a, const u32x4& b, f32x4& acc) {
  const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
}

// f32 -> CT with round-to-nearest-even
template <typename CT>
__device__ __forceinline__ CT to_ct(float v) {
  return static_cast<CT>(v);
}
template <typename CT>
__device__ __forceinline__ float to_f32(CT v) {
  return static_cast<float>(v);
}

// value of the neighbouring lane (lane ^ 1) on the DPP network (quad_perm [1,0,3,2])
__device__ __forceinline__ float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (every lane gets it): xor 1, xor 2 in the quad, then the mirrored half-row / row
__device__ __forceinline__ float row16_sum(float v) {
  v += lane_xor1(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
// sum over the 4 lanes of a quad (every lane gets it)
__device__ __forceinline__ float quad_sum(float v) {
  v += lane_xor1(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}
// two floats -> one dword of two bf16 (lo at the lower address), round-to-nearest-even
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
  bf16x2_v h;
  h[0] = static_cast<__bf16>(lo);
  h[1] = static_cast<__bf16>(hi);
  return __builtin_bit_cast(unsigned, h);
}

// pack 8 floats into one bf16 chunk / 4 floats into one f32 chunk
__device__ __forceinline__ u32x4 pack_bf16x8(const float (&v)[8]) {
  bf16x8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = static_cast<__bf16>(v[i]);
  return __builtin_bit_cast(u32x4, h);
}
__device__ __forceinline__ u32x4 pack_f32x4(float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  return __builtin_bit_cast(u32x4, v);
}

// unpack a chunk into KPL floats
template <typename CT>
__device__ __forceinline__ void unpack_chunk(const u32x4& c, float* out);
template <>
__device__ __forceinline__ void unpack_chunk<__bf16>(const u32x4& c, float* out) {
  const bf16x8 h = __builtin_bit_cast(bf16x8, c);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = static_cast<float>(h[i]);
}
template <>
__device__ __forceinline__ void unpack_chunk<float>(const u32x4& c, float* out) {
  const f32x4 v = __builtin_bit_cast(f32x4, c);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = v[i];
}

// bijective XCD-aware block remap (8 XCDs; block b is dispatched to XCD b % 8): gives every
// XCD a contiguous run of logical ids so that neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// the same function for outputs that are rounded to bf16 anyway: 0.5 x (1 + tanh u) = x * sigmoid(2u) = x / (1 + e^(-2u)),
// one v_exp_f32 and one v_rcp_f32 instead of the ~40-instruction tanhf expansion (relative error ~1e-6, bf16 keeps
// 2^-9); saturates correctly: e^(-2u) -> inf gives x * 0, -> 0 gives x
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);                 // 2u
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u2));
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // flax.linen.gelu(approximate=True): 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ))
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}

}  // namespace mt3k
#endif  // MT3_DEVICE_H_

// Launcher interface of the MXFP8 dense path (gemm_mx8.hip) towards the engine.
#ifndef MT3_MX8_H_
#define MT3_MX8_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mt3k {

// MXFP8 operands (gemm_mx8.hip): e4m3fn bytes [rows][K] + E8M0 block scales [rows][K / 32] (one per 32 consecutive K)
struct Mx8Args {
  const uint8_t* A;      // [M][K]
  const uint8_t* a_sc;   // [M][K/32]
  const uint8_t* W;      // [N][K]
  const uint8_t* w_sc;   // [N][K/32]
  void* out;             // STORE: bf16 [M][ldo]; HEADS: bf16 [2][B][H][seq][64]; RESID: f32 [M][ldo] (+=); GEGLU: unused
  int M, N, K, ldo, seq_len;
  const float* a_ss;     // non-null: fused RMSNorm, [M][K/16] partial sums of squares of the rows A was quantised from
  // RESID: the updated rows again as MXFP8 [M][N] / [M][N/32] and their per-16-column sums of squares [M][N/16];
  // GEGLU: THE output, MXFP8 [M][N/2] / [M][N/64] (ldo = N / 2)
  uint8_t* out_q;
  uint8_t* out_sc;
  float* out_ss;
};
int launch_gemm_mx8(const Mx8Args& g, int epi, hipStream_t s);
// rows (f32 or bf16) -> MXFP8; ss (f32 input only, may be null): per-16-column sums of squares
int launch_mx8_quantize(const void* in, bool in_f32, int M, int K, uint8_t* q, uint8_t* sc, float* ss, hipStream_t s);

}  // namespace mt3k
#endif  // MT3_MX8_H_

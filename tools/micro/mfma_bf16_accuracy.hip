// How exact is the bf16 matrix instruction's accumulation, and what does a split-bf16 f32 product keep?  (round 4)
//   (1) ONE v_mfma_f32_16x16x32_bf16 on bf16 operands with a wide exponent spread against the exact (double) sum of its 32
//       products: the error in units of sum |a_k b_k| says how many bits the instruction's internal adder keeps;
//   (2) an f32 x f32 dot product of K = 512 (unit normal operands) three ways against double: the f32 instruction
//       (128 x v_mfma_f32_16x16x4_f32), bf16 x 3 (operands split hi + lo, products hh + hl + lh) and bf16 x 6 (hi + mid + lo,
//       products hh + hm + mh + hl + lh + mm).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_bf16_accuracy.hip -o build/micro/mfma_bf16_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// A [16][K] / B [16][K] (row i of A, column n of B as a row), K a multiple of 32, as planes of bf16; mode: number of products
__global__ void dot_bf16(const uint16_t* ah, const uint16_t* am, const uint16_t* al, const uint16_t* bh, const uint16_t* bm,
                         const uint16_t* bl, int K, int stride, int mode, f32x4* c) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  auto ld = [&](const uint16_t* p, int k0) { return *reinterpret_cast<const bf16x8*>(p + r * stride + k0 + g * 8); };
  for (int k0 = 0; k0 < K; k0 += 32) {
    const bf16x8 Ah = ld(ah, k0), Bh = ld(bh, k0);
    if (mode >= 6) {        // small terms first
      const bf16x8 Am = ld(am, k0), Al = ld(al, k0), Bm = ld(bm, k0), Bl = ld(bl, k0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, acc, 0, 0, 0);
    } else if (mode >= 3) {
      const bf16x8 Am = ld(am, k0), Bm = ld(bm, k0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc, 0, 0, 0);
  }
  c[l] = acc;
}

__global__ void dot_f32(const float* a, const float* b, int K, f32x4* c) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r * K + k0 + g], b[r * K + k0 + g], acc, 0, 0, 0);
  c[l] = acc;
}

static uint16_t bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
static float bf16_val(uint16_t h) {
  uint32_t u = static_cast<uint32_t>(h) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

int main() {
  std::mt19937 rng(3);
  const int K = 512;
  std::vector<float> A(16 * K), B(16 * K);
  std::vector<uint16_t> P[6];
  for (auto& p : P) p.assign(16 * K, 0);
  uint16_t* dP[6];
  float *dA, *dB;
  f32x4* dC;
  for (int i = 0; i < 6; ++i) hipMalloc(&dP[i], 16 * K * 2);
  hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, 16 * K * 4); hipMalloc(&dC, 64 * 16);
  std::vector<float> C(256);
  auto upload = [&]() {
    for (int i = 0; i < 6; ++i) hipMemcpy(dP[i], P[i].data(), 16 * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(dA, A.data(), 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 16 * K * 4, hipMemcpyHostToDevice);
  };
  auto split = [&]() {       // A -> P[0..2], B -> P[3..5]: hi, mid, lo (exact residuals, RNE)
    for (int s = 0; s < 2; ++s) {
      const std::vector<float>& X = s ? B : A;
      for (int i = 0; i < 16 * K; ++i) {
        const uint16_t h = bf16_rne(X[i]);
        const float r1 = X[i] - bf16_val(h);
        const uint16_t m = bf16_rne(r1);
        const float r2 = r1 - bf16_val(m);
        P[3 * s][i] = h; P[3 * s + 1][i] = m; P[3 * s + 2][i] = bf16_rne(r2);
      }
    }
  };
  auto report = [&](const char* what, int kk, bool planes_exact) {
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    double worst_abs = 0, worst_rel = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int i = (l >> 4) * 4 + r, n = l & 15;
        double exact = 0, sabs = 0;
        for (int k = 0; k < kk; ++k) {
          const double a = planes_exact ? bf16_val(P[0][i * K + k]) : A[i * K + k];
          const double b = planes_exact ? bf16_val(P[3][n * K + k]) : B[n * K + k];
          exact += a * b;
          sabs += std::fabs(a * b);
        }
        const double err = std::fabs(C[l * 4 + r] - exact);
        worst_abs = std::fmax(worst_abs, err / sabs);
        worst_rel = std::fmax(worst_rel, err / std::fmax(std::fabs(exact), 1e-30));
      }
    std::printf("%-64s max |err| / sum|p| = %.3e   max |err| / |exact| = %.3e\n", what, worst_abs, worst_rel);
  };
  // (1) one instruction, exponents spread over 2^-8 .. 2^8
  for (int spread : {0, 4, 8, 12}) {
    for (int i = 0; i < 16 * K; ++i) {
      auto val = [&]() {
        const float m = 1.f + static_cast<float>(rng() % 128) / 128.f;
        const int e = spread ? static_cast<int>(rng() % (2 * spread + 1)) - spread : 0;
        return std::ldexp((rng() & 1) ? m : -m, e);
      };
      A[i] = val(); B[i] = val();
    }
    split();
    upload();
    (void)hipDeviceSynchronize(); dot_bf16<<<1, 64>>>(dP[0], dP[1], dP[2], dP[3], dP[4], dP[5], 32, K, 1, dC);
    char buf[96];
    std::snprintf(buf, sizeof buf, "(1) one v_mfma_f32_16x16x32_bf16, operand exponents within +-%d:", spread);
    report(buf, 32, true);
  }
  // (2) f32 dot products of K = 512
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : A) v = nd(rng);
  for (auto& v : B) v = nd(rng);
  split();
  upload();
  dot_f32<<<1, 64>>>(dA, dB, K, dC);
  report("(2) K = 512 f32 operands: 128 x v_mfma_f32_16x16x4_f32", K, false);
  dot_bf16<<<1, 64>>>(dP[0], dP[1], dP[2], dP[3], dP[4], dP[5], K, K, 1, dC);
  report("    bf16 x 1 (hi . hi)", K, false);
  dot_bf16<<<1, 64>>>(dP[0], dP[1], dP[2], dP[3], dP[4], dP[5], K, K, 3, dC);
  report("    bf16 x 3 (hi + lo: hh + hl + lh)", K, false);
  dot_bf16<<<1, 64>>>(dP[0], dP[1], dP[2], dP[3], dP[4], dP[5], K, K, 6, dC);
  report("    bf16 x 6 (hi + mid + lo: hh + hm + mh + hl + lh + mm)", K, false);
  return 0;
}

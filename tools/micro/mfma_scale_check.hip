// Operand model of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 inputs, checked on the hardware:
//   (1) one wave, random e4m3 bytes and random E8M0 scales against a host model in which lane l feeds row / column
//       l & 15, with g = l >> 4 the elements k = 16 g .. 16 g + 15 (VGPRs 0-3) and 64 + 16 g .. (VGPRs 4-7), and
//       SUPPLIES (byte op_sel of its scale VGPR) the scale of the consecutive block k = 32 g .. 32 g + 31;
//   (2) which outputs move when a single lane's scale is doubled (the scale -> (row, block) map, printed);
//   (3) which data VGPRs a scale lane owns (how the model of (1) was found: the first guess, 32 consecutive k per
//       lane, is wrong).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_scale_check.hip -o build/micro/mfma_scale_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>

using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int OPSEL>
__global__ void one_mfma(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* c) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OPSEL, sa[l], OPSEL, sb[l]);
  c[l] = acc;
}

static float e4m3_value(uint8_t b) {
  const int e = (b >> 3) & 15, m = b & 7;
  const float v = e == 0 ? std::ldexp(float(m) / 8.f, -6) : std::ldexp(1.f + float(m) / 8.f, e - 7);
  return (b & 0x80) ? -v : v;
}

int main() {
  std::mt19937 rng(1);
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  std::vector<int> SA(64), SB(64);
  i32x8 *dA, *dB;
  int *dSA, *dSB;
  f32x4* dC;
  hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dC, 64 * 16);
  std::vector<float> C(256);
  auto run = [&](int opsel) {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    if (opsel == 0) one_mfma<0><<<1, 64>>>(dA, dB, dSA, dSB, dC);
    else one_mfma<2><<<1, 64>>>(dA, dB, dSA, dSB, dC);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  };
  // (1) random operands
  for (int opsel : {0, 2}) {
    for (auto& v : A) { v = rng() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (auto& v : B) { v = rng() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (int l = 0; l < 64; ++l) {
      const int sa = 120 + rng() % 14, sb = 120 + rng() % 14;
      // the selected byte holds the scale, every other byte junk
      SA[l] = int((rng() & ~(0xffu << (8 * opsel))) | (unsigned(sa) << (8 * opsel)));
      SB[l] = int((rng() & ~(0xffu << (8 * opsel))) | (unsigned(sb) << (8 * opsel)));
    }
    run(opsel);
    double worst = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double t = 0, m = 0;
        for (int g = 0; g < 4; ++g) {
          const int la = i + 16 * g, lb = j + 16 * g;
          for (int p = 0; p < 32; ++p) {
            const int k = p < 16 ? 16 * g + p : 64 + 16 * g + (p - 16);      // what (3) below found
            const int sl = k / 32;                                          // the scale comes from lane row + 16 * (k / 32)
            const double s = std::ldexp(1.0, ((SA[i + 16 * sl] >> (8 * opsel)) & 0xff) - 127) *
                             std::ldexp(1.0, ((SB[j + 16 * sl] >> (8 * opsel)) & 0xff) - 127);
            const double pr = double(e4m3_value(A[la * 32 + p])) * e4m3_value(B[lb * 32 + p]) * s;
            t += pr;
            m += std::fabs(pr);
          }
        }
        const float got = C[(j + 16 * (i / 4)) * 4 + (i % 4)];                 // lane = col + 16 * (row / 4), reg = row % 4
        worst = std::fmax(worst, std::fabs(got - t) / m);
      }
    printf("(1) op_sel %d: random operands and scales vs the host model: worst |error| / sum |products| = %.3e\n", opsel, worst);
  }
  // (2) all ones; double one lane's A scale (then B scale): which outputs move, and by how much (128 = all of K)
  for (int side = 0; side < 2; ++side) {
    printf("(2) doubling the %s scale of lane l0 -> rows / columns that move (delta):\n", side ? "B" : "A");
    for (int l0 : {0, 1, 15, 16, 17, 31, 32, 47, 48, 63}) {
      for (auto& v : A) v = 0x38;
      for (auto& v : B) v = 0x38;
      for (int l = 0; l < 64; ++l) SA[l] = SB[l] = 127;
      (side ? SB : SA)[l0] = 128;
      run(0);
      printf("    l0 %2d:", l0);
      for (int x = 0; x < 16; ++x) {
        // row x (side A) / column x (side B): look at element (x, 0) / (0, x)
        const int i = side ? 0 : x, j = side ? x : 0;
        const float got = C[(j + 16 * (i / 4)) * 4 + (i % 4)];
        if (got != 128.f) printf(" %d(%+g)", x, got - 128.f);
      }
      printf("\n");
    }
  }
  // (3) which scale lane owns which VGPR of which data lane (row 0: lanes 0, 16, 32, 48): all ones, the A scale of lane
  // 16 s doubled, VGPR v of data lane 16 g zeroed -> row 0 reads 128 + 32 - 4 (not in the doubled block) or - 8 (in it)
  for (int side = 0; side < 2; ++side) {
    printf("(3) %s operand: scale lane 16 s owns these (data lane 16 g, VGPR v):\n", side ? "B" : "A");
    for (int sl = 0; sl < 4; ++sl) {
      printf("    s %d:", sl);
      for (int g = 0; g < 4; ++g)
        for (int v = 0; v < 8; ++v) {
          for (auto& x : A) x = 0x38;
          for (auto& x : B) x = 0x38;
          for (int l = 0; l < 64; ++l) SA[l] = SB[l] = 127;
          (side ? SB : SA)[16 * sl] = 128;
          for (int b = 0; b < 4; ++b) (side ? B : A)[(16 * g) * 32 + v * 4 + b] = 0;
          run(0);
          const float got = C[0];
          if (got == 128.f + 32 - 8) printf(" (%d,%d)", g, v);
          else if (got != 128.f + 32 - 4) printf(" (%d,%d:%g?)", g, v, got);
        }
      printf("\n");
    }
  }
  return 0;
}

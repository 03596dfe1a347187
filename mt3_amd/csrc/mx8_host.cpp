// Host side of the MXFP8 dense path: weights [N][K] (f32, output-major) -> OCP e4m3fn bytes + one E8M0 scale per 32
// consecutive K elements, by the same rule as the device quantiser (gemm_mx8.hip: scale = 2^(floor(log2 amax) - 7),
// round-to-nearest-even, nothing saturates).  Pure C++ (no device code), so the CPU tests can pin it against
// torch.float8_e4m3fn without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "common.h"
#include "mt3_hip.h"

namespace {

// |v| <= 448 -> e4m3fn bits (sign excluded), round-to-nearest-even; subnormals m / 8 * 2^-6
uint8_t e4m3_bits(float a) {
  if (!(a > 0.f)) return 0;
  if (a < 0.015625f) {                                   // below 2^-6: multiples of 2^-9 (8 = the smallest normal)
    return static_cast<uint8_t>(std::nearbyint(a * 512.f));
  }
  int e;
  const float m = std::frexp(a, &e);                     // a = m * 2^e, m in [0.5, 1)
  int q = static_cast<int>(std::nearbyint((m * 2.f - 1.f) * 8.f));   // 3 mantissa bits, ties to even
  int ex = e - 1 + 7;
  if (q == 8) {
    q = 0;
    ++ex;
  }
  const int bits = (ex << 3) | q;
  return static_cast<uint8_t>(bits > 0x7e ? 0x7e : bits);
}

}  // namespace

extern "C" int mt3_host_mx8_quantize(const float* h_w, int64_t rows, int64_t K, uint8_t* h_q, uint8_t* h_sc) {
  if (!h_w || !h_q || !h_sc || rows <= 0 || K <= 0 || K % 32)
    return mt3::fail(MT3_ERR_INVALID, "mt3_host_mx8_quantize: bad arguments (K = 32n)");
  const int64_t nb = K / 32;
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t b = 0; b < nb; ++b) {
      const float* v = h_w + r * K + b * 32;
      float amax = 0.f;
      for (int j = 0; j < 32; ++j) {
        if (!std::isfinite(v[j])) return mt3::fail(MT3_ERR_INVALID, "mt3_host_mx8_quantize: non-finite value in the matrix");
        amax = std::fmax(amax, std::fabs(v[j]));
      }
      uint32_t u;
      std::memcpy(&u, &amax, 4);
      const uint32_t e = (u >> 23) & 0xffu;
      const uint32_t byte = e > 7u ? e - 7u : 0u;
      const uint32_t ie = 261u - e;
      const uint32_t ib = (ie > 254u ? 254u : ie) << 23;
      float inv;
      std::memcpy(&inv, &ib, 4);
      h_sc[r * nb + b] = static_cast<uint8_t>(byte);
      for (int j = 0; j < 32; ++j) {
        const float s = v[j] * inv;                      // exact: a power of two
        h_q[r * K + b * 32 + j] = static_cast<uint8_t>(e4m3_bits(std::fabs(s)) | (std::signbit(s) ? 0x80 : 0));
      }
    }
  return MT3_OK;
}

// Does v_dot2c_f32_bf16 on gfx950 compute a.x*b.x + a.y*b.y + c?  (A/B against unpacked FMAs.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>
static uint32_t rnd_bf16() { float f = (rand() % 2001 - 1000) / 250.f; uint32_t u; std::memcpy(&u, &f, 4); return u >> 16; }
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* a, const uint32_t* b, float* dot, float* ref) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t x = a[i], y = b[i];
  dot[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, y), 1.0f, false);
  const float x0 = __builtin_bit_cast(float, x << 16), x1 = __builtin_bit_cast(float, x & 0xffff0000u);
  const float y0 = __builtin_bit_cast(float, y << 16), y1 = __builtin_bit_cast(float, y & 0xffff0000u);
  ref[i] = x0 * y0 + x1 * y1 + 1.0f;
}
// a dependent chain of 4 (the q.k of one 16-byte chunk), as the attention kernel issues it
__global__ void k4(const uint32_t* a, const uint32_t* b, float* dot, float* ref) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  float s = 0.f, r = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t x = a[(i * 4 + j) % 1024], y = b[(i * 4 + j) % 1024];
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, y), s, false);
    const float x0 = __builtin_bit_cast(float, x << 16), x1 = __builtin_bit_cast(float, x & 0xffff0000u);
    const float y0 = __builtin_bit_cast(float, y << 16), y1 = __builtin_bit_cast(float, y & 0xffff0000u);
    r += x0 * y0 + x1 * y1;
  }
  dot[i] = s;
  ref[i] = r;
}
int main() {
  const int n = 1024;
  uint32_t ha[n], hb[n];
  srand(1);
  for (int i = 0; i < n; ++i) {
    ha[i] = rnd_bf16() | (rnd_bf16() << 16);
    hb[i] = rnd_bf16() | (rnd_bf16() << 16);
  }
  uint32_t *a, *b; float *d, *r;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&d, n * 4); hipMalloc(&r, n * 4);
  hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, d, r);
  float hd[n], hr[n];
  hipMemcpy(hd, d, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, r, n * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) worst = fmax(worst, fabs(hd[i] - hr[i]));
  hipLaunchKernelGGL(k4, dim3(1), dim3(256), 0, 0, a, b, d + 0, r + 0);
  float cd[256], cr[256];
  hipMemcpy(cd, d, 1024, hipMemcpyDeviceToHost); hipMemcpy(cr, r, 1024, hipMemcpyDeviceToHost);
  double worst4 = 0;
  for (int i = 0; i < 256; ++i) worst4 = fmax(worst4, fabs(cd[i] - cr[i]));
  printf("chain of 4 dot2c: max abs diff %.3g  sample %.6f vs %.6f\n", worst4, cd[5], cr[5]);
  printf("dot2 vs fma: max abs diff %.3g   sample dot %.6f ref %.6f  (a=%08x b=%08x)\n", worst, hd[3], hr[3], ha[3], hb[3]);
  return 0;
}

// Decode-attention kernels through the C ABI at the bench shape (B = 256, H = 6): time per launch against the number
// of cached keys, bf16 and e4m3 caches, self (append) and cross (no append).  Four rotating cache sets (> 256 MB MALL
// in total) so that every launch streams from HBM.  duration = a + b * keys separates the fixed cost from the stream.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/micro/attn_probe.cpp -L mt3_amd -lmt3hip \
//         -Wl,-rpath,'$ORIGIN/../../mt3_amd' -o build/micro/attn_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mt3_hip.h"

int main() {
  const int B = 256, H = 6, cap = 1024, L = 4;
  void *kc[L], *vc[L], *sc[L], *k8[L], *v8[L];
  for (int l = 0; l < L; ++l) {
    hipMalloc(&kc[l], size_t(B) * H * cap * 64 * 2); hipMalloc(&vc[l], size_t(B) * H * cap * 64 * 2);
    hipMalloc(&k8[l], size_t(B) * H * cap * 64);     hipMalloc(&v8[l], size_t(B) * H * cap * 64);
    hipMalloc(&sc[l], size_t(B) * H * cap * 8);
    hipMemset(kc[l], 0x3c, size_t(B) * H * cap * 64 * 2); hipMemset(vc[l], 0x3c, size_t(B) * H * cap * 64 * 2);
    hipMemset(k8[l], 0x38, size_t(B) * H * cap * 64);     hipMemset(v8[l], 0x38, size_t(B) * H * cap * 64);
    std::vector<float> ones(size_t(B) * H * cap * 2, 1.f);
    hipMemcpy(sc[l], ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
  }
  void *qkv, *out; int* step;
  hipMalloc(&qkv, size_t(B) * 3 * H * 64 * 2); hipMemset(qkv, 0x3c, size_t(B) * 3 * H * 64 * 2);
  hipMalloc(&out, size_t(B) * H * 64 * 2); hipMalloc(&step, B * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* q = static_cast<const char*>(qkv);
  for (int fp8 = 0; fp8 < 2; ++fp8) {
    for (int n : {1, 65, 129, 257, 513, 769, 1024, -256}) {       // -256: cross attention over 256 keys, no append
      const bool cross = n < 0;
      const int keys = cross ? 256 : n;
      std::vector<int> hs(B, keys - 1);
      hipMemcpy(step, hs.data(), B * 4, hipMemcpyHostToDevice);
      auto launch = [&](int l) {
        if (fp8)
          return mt3_op_decode_attention_fp8(q, 3 * H * 64, k8[l], v8[l], sc[l], cap, cross ? nullptr : q + H * 64 * 2,
                                             cross ? nullptr : q + 2 * H * 64 * 2, 3 * H * 64, cross ? nullptr : step,
                                             keys, out, B, H, s);
        return mt3_op_decode_attention(MT3_BF16, q, 3 * H * 64, kc[l], vc[l], cap, cross ? nullptr : q + H * 64 * 2,
                                       cross ? nullptr : q + 2 * H * 64 * 2, 3 * H * 64, cross ? nullptr : step, keys,
                                       out, B, H, s);
      };
      for (int i = 0; i < 8; ++i) if (launch(i % L)) { printf("error: %s\n", mt3_last_error()); return 1; }
      hipStreamSynchronize(s);
      const int reps = 40;
      hipEventRecord(e0, s);
      for (int i = 0; i < reps; ++i) launch(i % L);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      const double bytes = double(B) * H * keys * (fp8 ? 2 * 64 + 8 : 2 * 64 * 2);
      printf("%s %-5s keys %4d : %6.2f us per launch (back to back)  %6.0f GB/s\n", fp8 ? "e4m3" : "bf16",
             cross ? "cross" : "self", keys, us, bytes / (us * 1e-6) / 1e9);
    }
  }
  return 0;
}

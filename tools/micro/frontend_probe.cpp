// Times the log-mel frontend through the C ABI (libmt3hip.so as built in-tree): 256 and 2048 full segments, HIP
// events, plus a checksum of the output so that two builds can be compared without a Python round trip.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/micro/frontend_probe.cpp -L mt3_amd -lmt3hip \
//         -Wl,-rpath,'$ORIGIN/../../mt3_amd' -o build/micro/frontend_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "mt3_hip.h"

int main() {
  mt3_frontend_config cfg{16000, 128, 512, 2048, 20.0f, 7600.0f};
  mt3_frontend* fe = nullptr;
  if (mt3_frontend_create(&cfg, &fe)) { printf("create: %s\n", mt3_last_error()); return 1; }
  const int S = 2048, N = 32768;
  std::vector<float> h(static_cast<size_t>(S) * N);
  unsigned s = 12345u;
  for (size_t i = 0; i < h.size(); ++i) {            // tones + noise, deterministic
    s = s * 1664525u + 1013904223u;
    const float noise = (static_cast<int>(s >> 9) - (1 << 22)) / static_cast<float>(1 << 22);
    h[i] = 0.5f * sinf(0.05f * static_cast<float>(i % N) * (1 + (i / N) % 7)) + 0.05f * noise;
  }
  float *d_a, *d_o;
  hipMalloc(&d_a, h.size() * 4); hipMalloc(&d_o, static_cast<size_t>(S) * 256 * 512 * 4);
  hipMemcpy(d_a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int n : {256, 2048}) {
    for (int i = 0; i < 3; ++i) mt3_frontend_logmel(fe, d_a, n, 256, nullptr, d_o, st);
    hipStreamSynchronize(st);
    const int reps = 20;
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) mt3_frontend_logmel(fe, d_a, n, 256, nullptr, d_o, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("logmel %4d segments: %8.1f us  -> %7.1f GB/s algorithmic (%.3f of 8 TB/s)\n", n, us,
           655360.0 * n / (us * 1e-6) / 1e9, 655360.0 * n / (us * 1e-6) / 8e12);
  }
  std::vector<float> o(static_cast<size_t>(4) * 256 * 512);
  hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
  double sum = 0, sq = 0;
  for (float v : o) { sum += v; sq += static_cast<double>(v) * v; }
  printf("checksum over 4 segments: sum %.6f  sumsq %.6f  first %.6f %.6f %.6f\n", sum, sq, o[0], o[777], o[123456]);
  return 0;
}

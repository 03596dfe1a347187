// What bounds the encoder's LDS-DMA staged GEMM tile (mt3k::gemm_glds_kernel)?  Builds the product kernel with parts
// left out (-DMT3_GLDS_PROBE=n: 1 no MFMA / fragment reads, 2 no DMA beyond the prologue, 4 no epilogue) and times
// encoder-shaped launches (M = 65536).  Build (4 binaries) and run: tools/micro/build_glds_probe.sh, then on the GPU
// box `for p in 0 1 2 4 6; do build/micro/glds_probe_$p; done`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "gemm.hip"

int main() {
  struct Case { const char* name; int N, K, epi, ldo; };
  const Case cases[] = {{"qkv   N=1152 K=512  STORE", 1152, 512, MT3_EPI_STORE, 1152},
                        {"geglu N=2048 K=512  STORE", 2048, 512, MT3_EPI_STORE, 2048},
                        {"      same, ldo 2048+64     ", 2048, 512, MT3_EPI_STORE, 2112},
                        {"geglu N=2048 K=512  GEGLU", 2048, 512, MT3_EPI_GEGLU, 1024},
                        {"      same, ldo 1024+64     ", 2048, 512, MT3_EPI_GEGLU, 1088},
                        {"wo    N=512  K=1024 RESID", 512, 1024, MT3_EPI_RESID, 512},
                        {"      same, ldo 512+32      ", 512, 1024, MT3_EPI_RESID, 544}};
  const int M = 65536;
  void *A, *W, *O;
  float* ss;
  hipMalloc(&A, size_t(M) * 1024 * 2); hipMalloc(&W, 2048 * 1024 * 2); hipMalloc(&O, size_t(M) * 2176 * 4);
  hipMalloc(&ss, size_t(M) * 64 * 4);
  hipMemset(A, 0, size_t(M) * 1024 * 2); hipMemset(W, 0, 2048 * 1024 * 2); hipMemset(O, 0, size_t(M) * 2176 * 4);
  hipMemset(ss, 0, size_t(M) * 64 * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Case& c : cases) {
    mt3k::GemmArgs g{};
    g.A = A; g.Wt = W; g.out = O; g.M = M; g.N = c.N; g.K = c.K; g.lda = c.K; g.ldo = c.ldo;
    const bool norm2 = c.epi != MT3_EPI_RESID;
    g.a_ss = norm2 ? ss : nullptr;
    for (int i = 0; i < 3; ++i) mt3k::launch_gemm(MT3_BF16, g, false, norm2 ? 2 : 0, c.epi, false, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) mt3k::launch_gemm(MT3_BF16, g, false, norm2 ? 2 : 0, c.epi, false, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * M * c.N * c.K / (us * 1e-6) / 1e12;
    printf("probe %d ns %d bk %d  %-28s %8.1f us per launch  (%6.0f TF/s if it were the whole GEMM)  %d tiles, %.2f us per tile-slot at 2 WG/CU\n",
           MT3_GLDS_PROBE, MT3_GLDS_NS, MT3_GLDS_BK, c.name, us, tf, (M / 128) * (c.N / 128), us * 512.0 / ((M / 128) * (c.N / 128)));
  }
  return 0;
}

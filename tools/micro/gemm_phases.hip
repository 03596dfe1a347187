// Where do the ~5 us of a decode-sized GEMM go?  Includes the product kernel with phase marks compiled in
// (s_memtime at entry / first barrier / operands in LDS / MFMAs issued / stores acknowledged) and prints the
// per-phase averages over the workgroups of back-to-back launches with hot caches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mt3_amd/csrc tools/micro/gemm_phases.hip \
//         mt3_amd/csrc/errors.cpp -o build/micro/gemm_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long* g_prof = nullptr;       // [blocks][8]
#define MT3_PROF_MARK(i)                                                                      \
  do {                                                                                        \
    if ((i) == 4) __builtin_amdgcn_s_waitcnt(0);       /* stores acknowledged */              \
    if (threadIdx.x == 0 && g_prof) g_prof[blockIdx.x * 8 + (i)] = wall_clock64();            \
  } while (0)
#include "gemm.hip"

int main() {
  struct Case { const char* name; int M, N, K; bool a_f32, norm; int epi; };
  const Case cases[] = {{"qkv   (f32 A, norm, N=1152, K=512)", 256, 1152, 512, true, true, MT3_EPI_STORE},
                        {"o-proj (bf16 A, resid, N=512, K=384)", 256, 512, 384, false, false, MT3_EPI_RESID},
                        {"wo    (bf16 A, resid, N=512, K=1024)", 256, 512, 1024, false, false, MT3_EPI_RESID},
                        {"geglu (f32 A, norm, N=2048, K=512)", 256, 2048, 512, true, true, MT3_EPI_GEGLU}};
  unsigned long long* prof;
  hipMalloc(&prof, 4096 * 8 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &prof, sizeof(prof));
  void *A, *W, *O;
  hipMalloc(&A, 256 * 1024 * 4); hipMalloc(&W, 2048 * 1024 * 2); hipMalloc(&O, 256 * 2048 * 4);
  hipMemset(A, 0, 256 * 1024 * 4); hipMemset(W, 0, 2048 * 1024 * 2); hipMemset(O, 0, 256 * 2048 * 4);
  hipStream_t s; hipStreamCreate(&s);
  for (const Case& c : cases) {
    mt3k::GemmArgs g{};
    g.A = A; g.Wt = W; g.out = O; g.aux = nullptr; g.M = c.M; g.N = c.N; g.K = c.K; g.lda = c.K;
    g.ldo = c.epi == MT3_EPI_GEGLU ? c.N / 2 : c.N; g.seq_len = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) mt3k::launch_gemm(MT3_BF16, g, c.a_f32, c.norm, c.epi, true, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) mt3k::launch_gemm(MT3_BF16, g, c.a_f32, c.norm, c.epi, true, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int blocks = (c.M / 32) * (c.N / (c.epi == MT3_EPI_GEGLU ? 64 : 32));
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
    double d[4] = {0, 0, 0, 0}, d_norm = 0, d_epi = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int b = 0; b < blocks; ++b) {
      for (int i = 0; i < 4; ++i) d[i] += double(h[b * 8 + i + 1] - h[b * 8 + i]);
      d_norm += double(h[b * 8 + 5] - h[b * 8 + 3]);
      d_epi += double(h[b * 8 + 4] - h[b * 8 + 5]);
      if (h[b * 8] < first) first = h[b * 8];
      if (h[b * 8 + 4] > last) last = h[b * 8 + 4];
    }
    // wall_clock64 ticks at 100 MHz: 10 ns per tick
    printf("%-40s %5.2f us per launch (stream, back to back) | %3d wgs | entry->barrier %.2f  loads+LDS %.2f  MFMA %.2f  "
           "norm-stat %.2f  stores+ack %.2f us | first entry -> last ack %.2f us\n",
           c.name, ms * 1e3 / reps, blocks, d[0] / blocks * 0.01, d[1] / blocks * 0.01, d[2] / blocks * 0.01,
           d_norm / blocks * 0.01, d_epi / blocks * 0.01, double(last - first) * 0.01);
  }
  return 0;
}

// Microbenchmark: in-graph cost of small dependent kernels on MI355X (what sets the ~4 us floor of the
// decode step's 75 launches?).  hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void k_dep(const int* idx, const float* tab, float* out) {   // two dependent loads, like embed
  const int i = idx[blockIdx.x];
  out[blockIdx.x * 256 + threadIdx.x] = tab[i * 256 + threadIdx.x] + 1.f;
}
__global__ void k_stream(const float4* in, float4* out, int n) {         // n float4 per block, streamed
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[(size_t)blockIdx.x * n + i] = in[(size_t)blockIdx.x * n + i];
}

template <typename F>
float time_graph(hipStream_t s, int chain, int reps, F enqueue) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < chain; ++i) enqueue(s);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (reps * chain);
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *p, *tab, *out; int* idx; float4 *big_in, *big_out;
  CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&tab, 2048 * 256 * 4)); CK(hipMalloc(&out, 2048 * 256 * 4));
  CK(hipMalloc(&idx, 2048 * 4)); CK(hipMemset(idx, 0, 2048 * 4)); CK(hipMemset(p, 0, 1 << 20));
  CK(hipMalloc(&big_in, 512u << 20)); CK(hipMalloc(&big_out, 512u << 20));
  const int chain = 64, reps = 50;
  for (int wg : {1, 256, 1536, 6144}) {
    for (int th : {64, 256}) {
      float e = time_graph(s, chain, reps, [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(th), 0, st); });
      float t = time_graph(s, chain, reps, [&](hipStream_t st) { hipLaunchKernelGGL(k_touch, dim3(wg), dim3(th), 0, st, p); });
      printf("grid %5d x %3d thr : empty %.2f us/kernel, touch(1 RMW per WG) %.2f us/kernel\n", wg, th, e, t);
    }
  }
  for (int wg : {256, 1536}) {
    float d = time_graph(s, chain, reps, [&](hipStream_t st) { hipLaunchKernelGGL(k_dep, dim3(wg), dim3(256), 0, st, idx, tab, out); });
    printf("grid %5d x 256 thr : dependent-load (embed-like) %.2f us/kernel\n", wg, d);
  }
  for (int kb : {4, 64, 256}) {          // KB per block, 1536 blocks
    const int n = kb * 1024 / 16;
    float d = time_graph(s, 16, 20, [&](hipStream_t st) { hipLaunchKernelGGL(k_stream, dim3(1536), dim3(256), 0, st, big_in, big_out, n); });
    printf("stream copy 1536 WG x %3d KB (%.0f MB r + w): %.2f us/kernel -> %.2f TB/s\n", kb, 1536.0 * kb / 1024, d,
           2.0 * 1536 * kb * 1024 / (d * 1e-6) / 1e12);
  }
  // same chain WITHOUT a graph (direct launches)
  {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("direct launches, empty 256x256: %.2f us/kernel\n", ms * 1e3f / 2000);
  }
  return 0;
}

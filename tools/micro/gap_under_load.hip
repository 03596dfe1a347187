// Microbenchmark for the NEXT step on the decode schedule ("Why four row groups ..."): a row group's
// step is 49 DEPENDENT launches, and in situ the time between two of them is ~4 us (200 us of the 637 us a group spends
// outside attention) against the 1.66 us in-graph floor of an idle chip.  What stretches it -- the memory system being
// saturated by the other groups' K/V streams (the release / acquire at a kernel boundary has to get through it), or the
// command processor serving four busy queues?  This probe separates the two:
//   chain  : a captured graph of N dependent small kernels (one 64 KB read-modify-write each, like a decode GEMM's
//            epilogue) replayed on a stream with a hardware queue of its own -> us per dependent launch;
//   load   : 0 .. 3 other such streams, each running either
//              (a) a streaming kernel that reads a 1 GiB buffer over and over (HBM saturated, few launches), or
//              (b) its own chain of dependent small kernels (queues busy, memory idle).
// Prints the chain's us per launch for every (kind of load, number of loaded streams).
// hipcc --offload-arch=gfx950 -O3 gap_under_load.hip -o gap_under_load   (sources only; not run in round 4: no GPU minutes left)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
      return 1;                                                               \
    }                                                                         \
  } while (0)

__global__ void k_small(float* p) {   // 64 workgroups x 256 threads x 4 B: 64 KB read + written
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  p[i] += 1.f;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void k_stream(const float4* __restrict__ in, float* __restrict__ sink, size_t n_per_block) {
  // every workgroup streams its slice once; the sum keeps the loads alive
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(in) + static_cast<size_t>(blockIdx.x) * n_per_block;
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

static int make_queue_stream(hipStream_t* s) {   // a stream with a hardware queue of its own (a CU mask of all CUs)
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
  for (int i = 0; i < n_cu; ++i) mask[i >> 5] |= 1u << (i & 31);
  CK(hipExtStreamCreateWithCUMask(s, static_cast<uint32_t>(mask.size()), mask.data()));
  return 0;
}

static int capture_chain(hipStream_t s, float* p, int n, hipGraphExec_t* out) {
  hipGraph_t g;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, p);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
  return 0;
}

int main() {
  constexpr int kChain = 49, kReps = 400, kLoads = 3;
  constexpr size_t kBig = 1ull << 30;
  hipStream_t probe, load[kLoads];
  if (make_queue_stream(&probe)) return 1;
  for (int i = 0; i < kLoads; ++i)
    if (make_queue_stream(&load[i])) return 1;
  float *p_probe, *p_load[kLoads], *sink;
  float4* big[kLoads];
  CK(hipMalloc(&p_probe, 64 * 256 * 4));
  CK(hipMemset(p_probe, 0, 64 * 256 * 4));
  CK(hipMalloc(&sink, 4096 * 4));
  for (int i = 0; i < kLoads; ++i) {
    CK(hipMalloc(&p_load[i], 64 * 256 * 4));
    CK(hipMemset(p_load[i], 0, 64 * 256 * 4));
    CK(hipMalloc(&big[i], kBig));
    CK(hipMemset(big[i], 0, kBig));
  }
  hipGraphExec_t chain, load_chain[kLoads];
  if (capture_chain(probe, p_probe, kChain, &chain)) return 1;
  for (int i = 0; i < kLoads; ++i)
    if (capture_chain(load[i], p_load[i], kChain, &load_chain[i])) return 1;
  CK(hipDeviceSynchronize());

  for (int kind = 0; kind < 2; ++kind) {
    for (int n_load = 0; n_load <= kLoads; ++n_load) {
      if (kind == 1 && n_load == 0) continue;
      std::atomic<bool> stop{false};
      std::vector<std::thread> feeders;
      for (int i = 0; i < n_load; ++i)
        feeders.emplace_back([&, i] {            // one host thread per loaded stream keeps ~2 launches queued
          (void)hipSetDevice(0);
          while (!stop.load()) {
            if (kind == 0) {
              // 1024 workgroups x 1 MiB each: one pass over the buffer, ~0.2 ms at HBM speed
              hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, load[i], big[i], sink, kBig / 16 / 1024);
              hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, load[i], big[i], sink, kBig / 16 / 1024);
            } else {
              (void)hipGraphLaunch(load_chain[i], load[i]);
              (void)hipGraphLaunch(load_chain[i], load[i]);
            }
            (void)hipStreamSynchronize(load[i]);
          }
        });
      // warm up, then time kReps replays of the chain
      for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(chain, probe));
      CK(hipStreamSynchronize(probe));
      hipEvent_t a, b;
      CK(hipEventCreate(&a));
      CK(hipEventCreate(&b));
      CK(hipEventRecord(a, probe));
      for (int r = 0; r < kReps; ++r) CK(hipGraphLaunch(chain, probe));
      CK(hipEventRecord(b, probe));
      CK(hipEventSynchronize(b));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, a, b));
      stop.store(true);
      for (auto& t : feeders) t.join();
      CK(hipDeviceSynchronize());
      printf("%d other stream(s) %s: %.2f us per dependent launch of the probe chain\n", n_load,
             kind == 0 ? "streaming 1 GiB buffers (HBM busy, queues quiet)" : "replaying their own chains (queues busy, HBM idle)",
             ms * 1e3f / (kReps * kChain));
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  }
  return 0;
}

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
#include <cstdlib>
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

// (second experiment) the product's shapes.  An attention-like launch: 1536 workgroups x 3 waves, each streaming 256 KB
// (one (batch, head) at 512 cached f32 keys) -- 18 of a CU's 32 wave slots, as dec_attn_kernel<float> leaves it;
// a GEMM-like chain kernel: 104 workgroups x 256 threads, each pulling 64 KB of "weights" that rotate through a 96 MB
// buffer (never L2-resident, as the f32 decode weights between two uses) before it writes its 4 KB tile
__global__ __launch_bounds__(192) void k_attn_like(const float4* __restrict__ in, float* __restrict__ sink) {
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(in) + static_cast<size_t>(blockIdx.x) * (256 * 1024 / 16);
  float acc = 0.f;
  for (int i = threadIdx.x; i < 256 * 1024 / 16; i += 192) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_gemm_like(const float4* __restrict__ w, float* __restrict__ out, int slice) {
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(w) + (static_cast<size_t>(slice) * 104 + blockIdx.x) * (64 * 1024 / 16);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) acc += src[i];
  out[blockIdx.x * 1024 + threadIdx.x * 4 + 0] += acc[0];
  out[blockIdx.x * 1024 + threadIdx.x * 4 + 1] += acc[1];
  out[blockIdx.x * 1024 + threadIdx.x * 4 + 2] += acc[2];
  out[blockIdx.x * 1024 + threadIdx.x * 4 + 3] += acc[3];
}

// a stream with a hardware queue of its own: a CU mask of all CUs, or (third experiment) of CUs [lo, hi) in eighths
static int make_queue_stream(hipStream_t* s, int lo8 = 0, int hi8 = 8) {
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
  for (int i = n_cu * lo8 / 8; i < n_cu * hi8 / 8; ++i) mask[i >> 5] |= 1u << (i & 31);
  CK(hipExtStreamCreateWithCUMask(s, static_cast<uint32_t>(mask.size()), mask.data()));
  return 0;
}

static int capture_chain(hipStream_t s, float* p, int n, hipGraphExec_t* out, const float4* weights = nullptr) {
  hipGraph_t g;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) {
    if (weights) hipLaunchKernelGGL(k_gemm_like, dim3(104), dim3(256), 0, s, weights, p, i % 14);   // 14 x 104 x 64 KB = 93 MB
    else hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, p);
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
  return 0;
}

int main(int argc, char** argv) {
  constexpr int kChain = 49, kReps = 200, kLoads = 3;
  constexpr size_t kBig = 1ull << 30;
  // gap_under_load [load_eighths [probe_on_the_rest]]: the loaded streams on the first load_eighths / 8 of the CUs (mask
  // bits), the probe chain on all CUs or only on the remaining ones; then only the attention-like experiments run
  const int load8 = argc > 1 ? atoi(argv[1]) : 8;
  const bool probe_rest = argc > 2 && atoi(argv[2]) != 0;
  const int attn_wgs = argc > 3 ? atoi(argv[3]) : 1536;      // 384 = a row group's quarter-batch launch
  hipStream_t probe, load[kLoads];
  if (make_queue_stream(&probe, probe_rest ? load8 : 0, 8)) return 1;
  for (int i = 0; i < kLoads; ++i)
    if (make_queue_stream(&load[i], 0, load8)) return 1;
  if (load8 != 8 || attn_wgs != 1536)
    printf("loaded streams on %d/8 of the CUs, probe chain on %s, attention-like launches of %d workgroups\n", load8,
           probe_rest ? "the other CUs" : "all CUs", attn_wgs);
  float *p_probe, *p_load[kLoads], *sink;
  float4* big[kLoads];
  CK(hipMalloc(&p_probe, 104 * 1024 * 4));
  CK(hipMemset(p_probe, 0, 104 * 1024 * 4));
  float4* weights;
  CK(hipMalloc(&weights, 14ull * 104 * 64 * 1024));
  CK(hipMemset(weights, 0, 14ull * 104 * 64 * 1024));
  CK(hipMalloc(&sink, 4096 * 4));
  for (int i = 0; i < kLoads; ++i) {
    CK(hipMalloc(&p_load[i], 64 * 256 * 4));
    CK(hipMemset(p_load[i], 0, 64 * 256 * 4));
    CK(hipMalloc(&big[i], kBig));
    CK(hipMemset(big[i], 0, kBig));
  }
  hipGraphExec_t chain, gemm_chain, load_chain[kLoads];
  if (capture_chain(probe, p_probe, kChain, &chain)) return 1;
  if (capture_chain(probe, p_probe, kChain, &gemm_chain, weights)) return 1;
  for (int i = 0; i < kLoads; ++i)
    if (capture_chain(load[i], p_load[i], kChain, &load_chain[i])) return 1;
  CK(hipDeviceSynchronize());

  for (int kind = 0; kind < 4; ++kind) {
    for (int n_load = 0; n_load <= kLoads; ++n_load) {
      if ((kind == 1 || kind == 2) && n_load == 0) continue;
      if ((load8 != 8 || attn_wgs != 1536) && kind < 2) continue;
      hipGraphExec_t timed_chain = kind == 3 ? gemm_chain : chain;
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
            } else if (kind == 1) {
              (void)hipGraphLaunch(load_chain[i], load[i]);
              (void)hipGraphLaunch(load_chain[i], load[i]);
            } else {                             // four attention-like launches of 1536 x 256 KB = 403 MB each
              for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_attn_like, dim3(attn_wgs), dim3(192), 0, load[i], big[i], sink);
            }
            (void)hipStreamSynchronize(load[i]);
          }
        });
      // warm up, then time kReps replays of the chain
      for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(timed_chain, probe));
      CK(hipStreamSynchronize(probe));
      hipEvent_t a, b;
      CK(hipEventCreate(&a));
      CK(hipEventCreate(&b));
      CK(hipEventRecord(a, probe));
      for (int r = 0; r < kReps; ++r) CK(hipGraphLaunch(timed_chain, probe));
      CK(hipEventRecord(b, probe));
      CK(hipEventSynchronize(b));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, a, b));
      stop.store(true);
      for (auto& t : feeders) t.join();
      CK(hipDeviceSynchronize());
      char attn_label[96];
      snprintf(attn_label, sizeof attn_label, "running attention-like launches (%d x 3 waves x 256 KB)", attn_wgs);
      const char* kKinds[4] = {"streaming 1 GiB buffers with every wave slot (HBM busy, queues quiet)",
                               "replaying their own chains (queues busy, HBM idle)", attn_label, attn_label};
      printf("%d other stream(s) %s: %.2f us per dependent launch of the %s chain\n", n_load, kKinds[kind],
             ms * 1e3f / (kReps * kChain), kind == 3 ? "GEMM-like (104 workgroups x 64 KB of cold weights)" : "small-kernel");
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  }
  return 0;
}

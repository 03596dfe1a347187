// Microbenchmark, the CU-split schedules of DESIGN.md section 7 a iii in miniature (measured: all slower than A): four "row groups", each alternating an
// attention-like launch (384 workgroups x 3 waves x 512 KB: a quarter-batch dec_attn_kernel<float> launch late in a decode) with three
// dependent GEMM-like launches (104 workgroups x 64 KB of cold weights) -- eight such layers per step.
//   A  one stream per group, every kernel on all CUs                              (the product's schedule today)
//   B  two streams per group -- attention on the first 6/8 of the CUs, dense launches on the other 2/8 -- chained by
//      events; a group's two queues are never busy at the same time, so at most four queues are (five busy queues are
//      2.2x slower than four: profiles/r4_ab_five_six_row_groups.txt)
//   C  as B with both streams on all CUs                                          (what the event hand-offs cost)
//   D  one stream per group again, the split done INSIDE the kernels: both kinds claim their work items from a counter,
//      attention-like workgroups leave at once when HW_ID says they sit on one of the two last CUs of a shader engine,
//      GEMM-like workgroups when they do not (profiles/r4_cu_mask_map.txt: that is the 6/8 - 2/8 split of B)
//   E  (round 6, VERDICT r5 #3c) ONE PERSISTENT launch per layer and group (256 workgroups, one per CU and group, all four
//      groups' workgroups resident): every workgroup streams its share of the layer's attention-like bytes, then 104 of
//      them run the three dependent dense stages, each waiting on a ready counter in memory (stage 0: the 256 workgroups'
//      attention shares; stage d: the 104 tiles of stage d - 1).  E0 loads a stage's 64 KB of weights AFTER its wait (what
//      removing three launch boundaries per layer buys by itself), E1 BEFORE it, into registers (what a dense stage
//      costs when its weight fetch hides behind the phase before it)
//   A' = A with the same dependent read added to the GEMM-like launches (the like-for-like baseline of E)
// Prints microseconds per group step for each.   hipcc --offload-arch=gfx950 -O3 cu_split_groups.hip -o cu_split_groups -lpthread
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(192) void k_attn_like(const f32x4_t* __restrict__ in, float* __restrict__ sink) {
  const f32x4_t* src = in + static_cast<size_t>(blockIdx.x) * (512 * 1024 / 16);
  float acc = 0.f;
  for (int i = threadIdx.x; i < 512 * 1024 / 16; i += 192) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_gemm_like(const f32x4_t* __restrict__ w, float* __restrict__ out, int slice) {
  const f32x4_t* src = w + (static_cast<size_t>(slice) * 104 + blockIdx.x) * (64 * 1024 / 16);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) acc += src[i];
  float* o = out + blockIdx.x * 1024 + threadIdx.x * 4;
  o[0] += acc[0], o[1] += acc[1], o[2] += acc[2], o[3] += acc[3];
}

// A': the GEMM-like launch of A that also consumes 4 KB of its producer's output (what E's stages do)
__global__ __launch_bounds__(256) void k_gemm_dep(const f32x4_t* __restrict__ w, float* __restrict__ out, const float* __restrict__ prev,
                                                  int slice) {
  const f32x4_t* src = w + (static_cast<size_t>(slice) * 104 + blockIdx.x) * (64 * 1024 / 16);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) acc += src[i];
  const f32x4_t a = *reinterpret_cast<const f32x4_t*>(prev + ((blockIdx.x * 7) % 104) * 1024 + threadIdx.x * 4);
  float* o = out + blockIdx.x * 1024 + threadIdx.x * 4;
  o[0] = acc[0] * a[0], o[1] = acc[1] * a[1], o[2] = acc[2] * a[2], o[3] = acc[3] * a[3];
}

// G (round 6): A' launched with hipExtAnyOrderLaunch -- no barrier bit between the launches of a stream, the order kept by
// the kernels themselves: every workgroup signals a per-group counter at its end, every kernel waits (one polling lane,
// bounded) until the counter has reached the number of workgroups launched before it.  The GEMM-like kernel reads its
// weights BEFORE the wait.  What a launch boundary costs against a flag, if the runtime honours the flag on gfx950.
__device__ __forceinline__ void g_wait(const int* ctr, int target, int* gave_up) {
  if (threadIdx.x == 0) {
    long spins = 0;
    // bounded: ~20 ms per wait, and once 64 waits have given up nobody waits any more (a broken run ends quickly)
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 20000 || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 64) {
        atomicAdd(gave_up, 1);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
__device__ __forceinline__ void g_signal(int* ctr) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(192) void k_attn_like_g(const f32x4_t* __restrict__ in, float* __restrict__ sink, int* ctr, int target,
                                                     int* gave_up) {
  g_wait(ctr, target, gave_up);
  const f32x4_t* src = in + static_cast<size_t>(blockIdx.x) * (512 * 1024 / 16);
  float acc = 0.f;
  for (int i = threadIdx.x; i < 512 * 1024 / 16; i += 192) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
  g_signal(ctr);
}
__global__ __launch_bounds__(256) void k_gemm_dep_g(const f32x4_t* __restrict__ w, float* __restrict__ out, const float* __restrict__ prev,
                                                    int slice, int* ctr, int target, int* gave_up) {
  const f32x4_t* src = w + (static_cast<size_t>(slice) * 104 + blockIdx.x) * (64 * 1024 / 16);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) acc += src[i];
  g_wait(ctr, target, gave_up);
  const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(prev + ((blockIdx.x * 7) % 104) * 1024 + threadIdx.x * 4));
  float* o = out + blockIdx.x * 1024 + threadIdx.x * 4;
  o[0] = acc[0] * a[0], o[1] = acc[1] * a[1], o[2] = acc[2] * a[2], o[3] = acc[3] * a[3];
  g_signal(ctr);
}

// G' : G without fences, by the guide's hand-off recipe (MI355X_MICROARCH.md, rows "handoff-flag" / "publish-large"): the
// producer's outputs leave as write-through (sc1) stores, are drained (vmcnt 0), then ONE relaxed device-scope add per
// workgroup; the consumer polls with a relaxed device-scope load and reads the producer's data with sc1 loads -- no
// buffer_wbl2 / buffer_inv anywhere (G's per-workgroup release fence writes an XCD's whole L2 back).
__device__ int g_poll_naps = 1;      // s_sleep 2 (~128 cycles) this many times between two polls (argv[2])
__device__ __forceinline__ void g2_wait(const int* ctr, int target, int* gave_up) {
  if (threadIdx.x == 0) {
    long spins = 0;
    const int naps = g_poll_naps;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(2);
      if (++spins > 20000 || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 64) {
        atomicAdd(gave_up, 1);
        break;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void g2_signal(int* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(192) void k_attn_like_g2(const f32x4_t* __restrict__ in, float* __restrict__ sink, int* ctr, int target,
                                                      int* gave_up) {
  g2_wait(ctr, target, gave_up);
  const f32x4_t* src = in + static_cast<size_t>(blockIdx.x) * (512 * 1024 / 16);
  float acc = 0.f;
  for (int i = threadIdx.x; i < 512 * 1024 / 16; i += 192) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
  g2_signal(ctr);
}
__global__ __launch_bounds__(256) void k_gemm_dep_g2(const f32x4_t* __restrict__ w, float* __restrict__ out, const float* __restrict__ prev,
                                                     int slice, int* ctr, int target, int* gave_up) {
  const f32x4_t* src = w + (static_cast<size_t>(slice) * 104 + blockIdx.x) * (64 * 1024 / 16);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) acc += src[i];
  g2_wait(ctr, target, gave_up);
  const float* pa = prev + ((blockIdx.x * 7) % 104) * 1024 + threadIdx.x * 4;
  float a[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = __hip_atomic_load(pa + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* o = out + blockIdx.x * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) __hip_atomic_store(o + j, acc[j] * a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  g2_signal(ctr);
}

// F (VERDICT r5 #3b, the out-projection inside the attention epilogue): the attention-like workgroups are (row, head) pairs,
// six per row; the LAST of a row's six to finish (a counter per row) projects the row itself -- a GEMV over the whole
// 1.4 MB out-projection matrix (896 x 384 f32), read by every one of the 64 rows -- instead of a 22-workgroup GEMM-like
// launch that reads the matrix once per 32-row tile.
__global__ __launch_bounds__(192) void k_attn_gemv_tail(const f32x4_t* __restrict__ in, float* __restrict__ sink,
                                                        const f32x4_t* __restrict__ w, float* __restrict__ out, int* row_cnt) {
  const f32x4_t* src = in + static_cast<size_t>(blockIdx.x) * (512 * 1024 / 16);
  float acc = 0.f;
  for (int i = threadIdx.x; i < 512 * 1024 / 16; i += 192) {
    const f32x4_t v = __builtin_nontemporal_load(src + i);
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(row_cnt + blockIdx.x / 6, 1) == 5;
  }
  __syncthreads();
  if (!last) return;
  // the row's projection: 896 x 384 f32 = 86,016 16-byte words, 448 per thread
  f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int i = threadIdx.x; i < 896 * 384 / 4; i += 192) a += w[i];
  float* o = out + (blockIdx.x / 6) * 1024 + threadIdx.x * 4;
  o[0] = a[0], o[1] = a[1], o[2] = a[2], o[3] = a[3];
}

// E: one PERSISTENT launch per layer and group: 256 workgroups (one per CU and group: with four groups at once all 1024
// workgroups are resident -- 16 waves per CU at <= 128 VGPRs -- so a waiting workgroup can never keep a producer from
// being dispatched.  The first form of this variant, a 696-workgroup grid whose dense workgroups spun on their producers'
// counters, DEADLOCKED with four groups in flight: dispatch is in order per XCD only, so consumers of group B fill XCD 5
// while B's producers wait for slots on XCD 2 behind consumers of group A whose producers wait on XCD 5).
// Every workgroup streams 3 of the layer's 768 attention-like half-items (256 KB each), then workgroups 0..103 run the
// three dependent dense stages: wait on the ready counter (stage 0: 256 finished workgroups; stage d: 104 tiles of stage
// d - 1), read a 4 KB slice of the producer's output, combine it with 64 KB of weights, publish.  PREFETCH: the weights
// of stage 0 are requested BEFORE the attention phase and those of stage d + 1 before the wait for stage d's tiles.
// NOFENCE (round 6, after G'): the same hand-offs by the guide's recipe -- write-through (sc1) stores of a stage's output,
// drained, one relaxed device-scope add; relaxed sc1 polls and sc1 loads of the producer's output; no release / acquire
// fence (buffer_wbl2 / buffer_inv sc1 act on an XCD's whole L2, which every other group is using)
template <bool PREFETCH, bool NOFENCE = false>
__global__ __launch_bounds__(256) void k_layer_fused(const f32x4_t* __restrict__ kv, float* __restrict__ sink,
                                                     const f32x4_t* __restrict__ w, float* __restrict__ b0, float* __restrict__ b1,
                                                     float* __restrict__ b2, float* __restrict__ b3, int slice, int* flags) {
  constexpr int kTiles = 104, kWgs = 256;
  const bool dense = blockIdx.x < kTiles;
  f32x4_t wr[16];
  auto load_weights = [&](int d) {
    const f32x4_t* src = w + (static_cast<size_t>(slice + d) % 14 * 104 + blockIdx.x) * (64 * 1024 / 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) wr[i] = src[threadIdx.x + i * 256];
  };
  if (PREFETCH && dense) load_weights(0);
  float acc = 0.f;
  for (int it = 0; it < 3; ++it) {
    const f32x4_t* src = kv + (static_cast<size_t>(it) * kWgs + blockIdx.x) * (256 * 1024 / 16);
    for (int i = threadIdx.x; i < 256 * 1024 / 16; i += 256) {
      const f32x4_t v = __builtin_nontemporal_load(src + i);
      acc += v[0] + v[1] + v[2] + v[3];
    }
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!NOFENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!dense) return;
  for (int d = 0; d < 3; ++d) {
    if (threadIdx.x == 0) {
      const int need = d == 0 ? kWgs : kTiles;
      // bounded: a wait that does not end must not hang the GPU -- after ~50 ms the workgroup goes on and counts itself
      long spins = 0;
      while (__hip_atomic_load(flags + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > 400000) {
          atomicAdd(flags + 7, 1);
          break;
        }
      }
      if (!NOFENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!PREFETCH) load_weights(d);
    f32x4_t a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) a4 += wr[i];
    const float* prev = d == 0 ? b0 : d == 1 ? b1 : b2;
    float* out = d == 0 ? b1 : d == 1 ? b2 : b3;
    const float* pa = prev + ((blockIdx.x * 7) % 104) * 1024 + threadIdx.x * 4;
    float* o = out + blockIdx.x * 1024 + threadIdx.x * 4;
    if constexpr (NOFENCE) {
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = __hip_atomic_load(pa + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int j = 0; j < 4; ++j) __hip_atomic_store(o + j, a4[j] * a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pa));
      o[0] = a4[0] * a[0], o[1] = a4[1] * a[1], o[2] = a4[2] * a[2], o[3] = a4[3] * a[3];
    }
    if (PREFETCH && d < 2) load_weights(d + 1);          // in flight while this stage is published and the next one awaited
    if (NOFENCE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!NOFENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(flags + 1 + d, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// this wave's CU is one of the two reserved for the dense launches (hardware CU ids run 0..7 or, with CU 0 harvested, 1..8)
__device__ __forceinline__ bool on_reserved_cu() {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID
  return (((hw >> 8) & 0xF) & 7u) >= 6u;
}
__global__ __launch_bounds__(192) void k_attn_claim(const f32x4_t* __restrict__ in, float* __restrict__ sink, int* counter, int items) {
  if (on_reserved_cu()) return;
  __shared__ int item;
  for (;;) {
    if (threadIdx.x == 0) item = atomicAdd(counter, 1);
    __syncthreads();
    const int it = item;
    __syncthreads();
    if (it >= items) return;
    const f32x4_t* src = in + static_cast<size_t>(it) * (512 * 1024 / 16);
    float acc = 0.f;
    for (int i = threadIdx.x; i < 512 * 1024 / 16; i += 192) {
      const f32x4_t v = __builtin_nontemporal_load(src + i);
      acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 123.456f) sink[it] = acc;
  }
}
__global__ __launch_bounds__(256) void k_gemm_claim(const f32x4_t* __restrict__ w, float* __restrict__ out, int slice, int* counter, int items) {
  if (!on_reserved_cu()) return;
  __shared__ int item;
  for (;;) {
    if (threadIdx.x == 0) item = atomicAdd(counter, 1);
    __syncthreads();
    const int it = item;
    __syncthreads();
    if (it >= items) return;
    const f32x4_t* src = w + (static_cast<size_t>(slice) * 104 + it) * (64 * 1024 / 16);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256) acc += src[i];
    float* o = out + it * 1024 + threadIdx.x * 4;
    o[0] += acc[0], o[1] += acc[1], o[2] += acc[2], o[3] += acc[3];
  }
}

static int make_stream(hipStream_t* s, int lo8, int hi8) {
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
  for (int i = n_cu * lo8 / 8; i < n_cu * hi8 / 8; ++i) mask[i >> 5] |= 1u << (i & 31);
  CK(hipExtStreamCreateWithCUMask(s, static_cast<uint32_t>(mask.size()), mask.data()));
  return 0;
}

int main(int argc, char** argv) {
  constexpr int G = 4, kSteps = 100, kLayers = 8, kDense = 3;
  f32x4_t *kv[G], *weights;
  float *out[G], *sink;
  CK(hipMalloc(&weights, 14ull * 104 * 64 * 1024));
  CK(hipMemset(weights, 0, 14ull * 104 * 64 * 1024));
  CK(hipMalloc(&sink, 4096 * 4));
  for (int g = 0; g < G; ++g) {
    CK(hipMalloc(&kv[g], 8ull * 384 * 512 * 1024));          // eight layers' worth of "cache": 1.6 GB per group
    CK(hipMemset(kv[g], 0, 8ull * 384 * 512 * 1024));
    CK(hipMalloc(&out[g], 104 * 1024 * 4));
    CK(hipMemset(out[g], 0, 104 * 1024 * 4));
  }
  const char* names[15] = {"A one stream per group, all CUs", "B attention on 6/8 of the CUs, dense launches on the other 2/8",
                          "C two streams per group, both on all CUs",
                          "D one stream per group, work-claiming kernels that keep to 6/8 (attention) and 2/8 (dense) of the CUs",
                          "A' = A with a dependent 4 KB read in every GEMM-like launch",
                          "E0 one launch per layer and group, dense stages wait on ready counters, weights loaded AFTER the wait",
                          "E1 as E0, weights prefetched into registers BEFORE the wait",
                          "A'' = A' with the FIRST dense launch of a layer at the out-projection's size (22 workgroups, 1.4 MB)",
                          "F  = A'' with that launch replaced by a per-row GEMV in the tail of the attention-like kernel",
                           "G  = A' launched with hipExtAnyOrderLaunch, order kept by a per-group counter (weights read before the wait)",
                           "G' = G without fences: write-through stores, drained, relaxed counter; sc1 loads of the producer's data",
                           "H  = G' with the barrier bit kept on the attention-like launches (run-ahead bounded to a layer's three dense launches)",
                           "A'x = A' launched through hipExtLaunchKernelGGL with flags 0 (what the launch API itself costs the host)",
                           "E0' = E0 without fences (sc1 stores drained + relaxed counter, sc1 loads)", "E1' = E1 without fences"};
  int* row_cnt[G];
  for (int g = 0; g < G; ++g) CK(hipMalloc(&row_cnt[g], (kSteps + 10) * kLayers * 64 * 4));
  float* bufs[G][4];
  int* flags[G];
  for (int g = 0; g < G; ++g) {
    for (int b = 0; b < 4; ++b) {
      CK(hipMalloc(&bufs[g][b], 104 * 1024 * 4));
      CK(hipMemset(bufs[g][b], 0, 104 * 1024 * 4));
    }
    CK(hipMalloc(&flags[g], (kSteps + 10) * kLayers * 8 * 4));
  }
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  if (argc > 2) {
    const int naps = atoi(argv[2]);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_poll_naps), &naps, sizeof(int)));
    printf("(G' / H: %d naps of s_sleep 2 between polls)\n", naps);
  }          // run one variant only (0..8); default: all but B, C, D
  // one claim counter per launch of variant D
  constexpr int kLaunches = (kSteps + 10) * kLayers * (1 + kDense);
  int* counters[G];
  for (int g = 0; g < G; ++g) CK(hipMalloc(&counters[g], kLaunches * 4));
  int* gctr[G];
  for (int g = 0; g < G; ++g) CK(hipMalloc(&gctr[g], 8));
  for (int variant = 0; variant < 15; ++variant) {
    for (int g = 0; g < G; ++g) CK(hipMemset(row_cnt[g], 0, (kSteps + 10) * kLayers * 64 * 4));
    if (only >= 0 ? variant != only : (variant >= 1 && variant <= 3)) continue;     // B, C, D: round 4's results stand
    for (int g = 0; g < G; ++g) CK(hipMemset(counters[g], 0, kLaunches * 4));
    for (int g = 0; g < G; ++g) CK(hipMemset(gctr[g], 0, 8));
    for (int g = 0; g < G; ++g) CK(hipMemset(flags[g], 0, (kSteps + 10) * kLayers * 8 * 4));
    hipStream_t sa[G], sd[G];
    hipEvent_t e1[G], e2[G];
    // creation order: the four attention streams first, then the four dense streams
    for (int g = 0; g < G; ++g)
      if (make_stream(&sa[g], 0, variant == 1 ? 6 : 8)) return 1;
    for (int g = 0; g < G; ++g) {
      if (variant == 0 || variant >= 3) sd[g] = sa[g];
      else if (make_stream(&sd[g], variant == 1 ? 6 : 0, 8)) return 1;
      CK(hipEventCreateWithFlags(&e1[g], hipEventDisableTiming));
      CK(hipEventCreateWithFlags(&e2[g], hipEventDisableTiming));
    }
    static int launched_total[G];
    for (int g = 0; g < G; ++g) launched_total[g] = 0;
    double host_enqueue_us = 0.0;
    auto run = [&](int g, int steps, int c0) {
      (void)hipSetDevice(0);
      const auto t_enq = std::chrono::steady_clock::now();
      int& launched = launched_total[g];
      int slice = g * 3;
      int* ctr = counters[g] + c0;
      int* fl = flags[g] + 2 * c0;                   // (c0 counts 4 launches per layer; 8 flag words per fused launch)
      for (int t = 0; t < steps; ++t)
        for (int l = 0; l < kLayers; ++l) {
          if (variant == 13 || variant == 14) {
            const f32x4_t* kvl = kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16);
            if (variant == 13)
              hipLaunchKernelGGL((k_layer_fused<false, true>), dim3(256), dim3(256), 0, sa[g], kvl, sink, weights, bufs[g][0],
                                 bufs[g][1], bufs[g][2], bufs[g][3], slice % 14, fl);
            else
              hipLaunchKernelGGL((k_layer_fused<true, true>), dim3(256), dim3(256), 0, sa[g], kvl, sink, weights, bufs[g][0],
                                 bufs[g][1], bufs[g][2], bufs[g][3], slice % 14, fl);
            fl += 8;
            slice += 3;
            continue;
          }
          if (variant == 5 || variant == 6) {
            const f32x4_t* kvl = kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16);
            if (variant == 5)
              hipLaunchKernelGGL(k_layer_fused<false>, dim3(256), dim3(256), 0, sa[g], kvl, sink, weights, bufs[g][0],
                                 bufs[g][1], bufs[g][2], bufs[g][3], slice % 14, fl);
            else
              hipLaunchKernelGGL(k_layer_fused<true>, dim3(256), dim3(256), 0, sa[g], kvl, sink, weights, bufs[g][0],
                                 bufs[g][1], bufs[g][2], bufs[g][3], slice % 14, fl);
            fl += 8;
            slice += 3;
            continue;
          }
          if (variant >= 7) {
            const f32x4_t* kvl = kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16);
            if (variant == 7) {
              hipLaunchKernelGGL(k_attn_like, dim3(384), dim3(192), 0, sa[g], kvl, sink);
              hipLaunchKernelGGL(k_gemm_dep, dim3(22), dim3(256), 0, sa[g], weights, bufs[g][1], bufs[g][0], slice % 14);
            } else {
              hipLaunchKernelGGL(k_attn_gemv_tail, dim3(384), dim3(192), 0, sa[g], kvl, sink, weights + (slice % 14) * (104 * 64 * 1024 / 16),
                                 bufs[g][1], row_cnt[g] + (c0 / 4 + t * kLayers + l) * 64);
            }
            ++slice;
            for (int d = 1; d < kDense; ++d) {
              hipLaunchKernelGGL(k_gemm_dep, dim3(104), dim3(256), 0, sa[g], weights, bufs[g][d + 1], bufs[g][d], slice % 14);
              ++slice;
            }
            continue;
          }
          if (variant == 10 || variant == 11) {
            const f32x4_t* kvl = kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16);
            hipExtLaunchKernelGGL(k_attn_like_g2, dim3(384), dim3(192), 0, sa[g], nullptr, nullptr,
                                  variant == 10 ? hipExtAnyOrderLaunch : 0, kvl, sink, gctr[g], variant == 10 ? launched : 0, gctr[g] + 1);
            launched += 384;
            for (int d = 0; d < kDense; ++d) {
              hipExtLaunchKernelGGL(k_gemm_dep_g2, dim3(104), dim3(256), 0, sa[g], nullptr, nullptr, hipExtAnyOrderLaunch, weights,
                                    bufs[g][d + 1], bufs[g][d], slice % 14, gctr[g], launched, gctr[g] + 1);
              launched += 104;
              ++slice;
            }
            continue;
          }
          if (variant == 9) {
            const f32x4_t* kvl = kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16);
            hipExtLaunchKernelGGL(k_attn_like_g, dim3(384), dim3(192), 0, sa[g], nullptr, nullptr, hipExtAnyOrderLaunch, kvl, sink,
                                  gctr[g], launched, gctr[g] + 1);
            launched += 384;
            for (int d = 0; d < kDense; ++d) {
              hipExtLaunchKernelGGL(k_gemm_dep_g, dim3(104), dim3(256), 0, sa[g], nullptr, nullptr, hipExtAnyOrderLaunch, weights,
                                    bufs[g][d + 1], bufs[g][d], slice % 14, gctr[g], launched, gctr[g] + 1);
              launched += 104;
              ++slice;
            }
            continue;
          }
          if (variant == 12) {
            hipExtLaunchKernelGGL(k_attn_like, dim3(384), dim3(192), 0, sa[g], nullptr, nullptr, 0,
                                  kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16), sink);
            for (int d = 0; d < kDense; ++d) {
              hipExtLaunchKernelGGL(k_gemm_dep, dim3(104), dim3(256), 0, sa[g], nullptr, nullptr, 0, weights, bufs[g][d + 1],
                                    static_cast<const float*>(bufs[g][d]), slice % 14);
              ++slice;
            }
            continue;
          }
          if (variant == 4) {
            hipLaunchKernelGGL(k_attn_like, dim3(384), dim3(192), 0, sa[g], kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16), sink);
            for (int d = 0; d < kDense; ++d) {
              hipLaunchKernelGGL(k_gemm_dep, dim3(104), dim3(256), 0, sa[g], weights, bufs[g][d + 1], bufs[g][d], slice % 14);
              ++slice;
            }
            continue;
          }
          if (variant == 3) {
            // 512 workgroups: about three quarters of them land on allowed CUs and claim the 384 items
            hipLaunchKernelGGL(k_attn_claim, dim3(512), dim3(192), 0, sa[g], kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16), sink, ctr++, 384);
            for (int d = 0; d < kDense; ++d) {
              // 1024 workgroups: the quarter that lands on reserved CUs claims the 104 tiles
              hipLaunchKernelGGL(k_gemm_claim, dim3(1024), dim3(256), 0, sa[g], weights, out[g], slice % 14, ctr++, 104);
              ++slice;
            }
            continue;
          }
          hipLaunchKernelGGL(k_attn_like, dim3(384), dim3(192), 0, sa[g], kv[g] + static_cast<size_t>(l) * 384 * (512 * 1024 / 16), sink);
          if (variant != 0) {
            (void)hipEventRecord(e1[g], sa[g]);
            (void)hipStreamWaitEvent(sd[g], e1[g], 0);
          }
          for (int d = 0; d < kDense; ++d) {
            hipLaunchKernelGGL(k_gemm_like, dim3(104), dim3(256), 0, sd[g], weights, out[g], slice % 14);
            ++slice;
          }
          if (variant != 0) {
            (void)hipEventRecord(e2[g], sd[g]);
            (void)hipStreamWaitEvent(sa[g], e2[g], 0);
          }
        }
      if (g == 0 && steps == kSteps)
        host_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq).count();
      (void)hipStreamSynchronize(sa[g]);
      (void)hipStreamSynchronize(sd[g]);
    };
    {   // warm-up
      std::vector<std::thread> th;
      for (int g = 0; g < G; ++g) th.emplace_back(run, g, 10, 0);
      for (auto& t : th) t.join();
    }
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) th.emplace_back(run, g, kSteps, 10 * kLayers * (1 + kDense));
    for (auto& t : th) t.join();
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    long timed_out = 0;
    if (variant >= 9) {
      for (int g = 0; g < G; ++g) {
        int h2[2];
        CK(hipMemcpy(h2, gctr[g], 8, hipMemcpyDeviceToHost));
        timed_out += h2[1];
        if (h2[0] != launched_total[g]) printf("   group %d: counter %d, launched %d\n", g, h2[0], launched_total[g]);
      }
    }
    if (variant == 5 || variant == 6 || variant == 13 || variant == 14) {
      std::vector<int> h((kSteps + 10) * kLayers * 8);
      for (int g = 0; g < G; ++g) {
        CK(hipMemcpy(h.data(), flags[g], h.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 7; i < h.size(); i += 8) timed_out += h[i];
      }
    }
    printf("%s: %.1f us per group step (%d layers of 1 attention-like + %d GEMM-like launches; four groups at once)%s\n",
           names[variant], us / kSteps, kLayers, kDense, timed_out ? "  [WAITS TIMED OUT: result void]" : "");
    if (timed_out) printf("   %ld waits gave up after ~50 ms\n", timed_out);
    printf("   host: group 0's thread spent %.1f us per group step enqueueing\n", host_enqueue_us / kSteps);
    fflush(stdout);
    for (int g = 0; g < G; ++g) {
      (void)hipStreamDestroy(sa[g]);
      if (variant == 1 || variant == 2) (void)hipStreamDestroy(sd[g]);
      (void)hipEventDestroy(e1[g]);
      (void)hipEventDestroy(e2[g]);
    }
  }
  return 0;
}

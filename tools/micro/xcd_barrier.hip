// Microbenchmark behind DESIGN.md section 3 ("why the decode step is not one persistent kernel"): what does ONE
// all-CU hand-off cost inside a persistent launch on MI355X, against the ~1.45 us of a dependent kernel boundary?
//   flat    : one monotonic agent-scope counter, release fence before arrive, relaxed poll, acquire fence after
//   xcd     : XCD-hierarchical (per-XCC arrival counter -> XCC leader -> top counter -> per-XCC generation flag)
//   xcdonly : barrier among the workgroups of ONE XCD only (the "row block per XCD" idea), same fences
// Placement-independent protocols only (agent-scope fences); XCC id is used for speed, never for correctness.
// hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip -o xcd_barrier && ./xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}
__device__ __forceinline__ unsigned poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every spin is BOUNDED (a stranded workgroup must not hang the GPU): gives up after ~5 ms and counts the failure
__device__ __forceinline__ void wait_ge(const unsigned* p, unsigned want, unsigned* fails) {
  for (int spins = 0; poll(p) < want; ++spins) {
    __builtin_amdgcn_s_sleep(1);
    if (spins > 200000) {
      atomicAdd(fails, 1u);
      return;
    }
  }
}

// mode 0 flat, 1 xcd-hierarchical, 2 xcd-only.  ctr[0] = top counter, ctr[16 + 16*x] = XCC x arrival counter,
// ctr[256 + 16*x] = XCC x generation; n_xcc_wg[x] = workgroups resident on XCC x (census pass).
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* ctr, const unsigned* n_xcc_wg, float* payload,
                                                      int rounds, int mode, int nwg) {
  const int x = xcc_id();
  const unsigned mine = n_xcc_wg[x];
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    // a little "work" + a published value so the fences have something to order
    payload[blockIdx.x * 256 + threadIdx.x] = acc + r;
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (mode == 0) {
        __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_ge(&ctr[0], static_cast<unsigned>(r) * nwg, &ctr[1000]);
      } else {
        const unsigned t = __hip_atomic_fetch_add(&ctr[16 + 16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == static_cast<unsigned>(r) * mine - 1) {            // last arriver of this XCC
          if (mode == 1) {
            __hip_atomic_fetch_add(&ctr[0], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_ge(&ctr[0], static_cast<unsigned>(r) * nwg, &ctr[1000]);
          }
          __hip_atomic_store(&ctr[256 + 16 * x], static_cast<unsigned>(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          wait_ge(&ctr[256 + 16 * x], static_cast<unsigned>(r), &ctr[1000]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    acc += payload[((blockIdx.x + 1) % nwg) * 256 + threadIdx.x];   // read a neighbour's published value
  }
  payload[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void census_kernel(unsigned* n_xcc_wg) {
  if (threadIdx.x == 0) atomicAdd(&n_xcc_wg[xcc_id()], 1u);
}

int main() {
  unsigned *ctr, *census;
  float* payload;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&census, 64)); CK(hipMalloc(&payload, 2048 * 256 * 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int rounds = 200;
  for (int nwg : {256, 512, 1024}) {
    CK(hipMemsetAsync(census, 0, 64, s));
    hipLaunchKernelGGL(census_kernel, dim3(nwg), dim3(256), 0, s, census);
    unsigned h[8]; CK(hipMemcpyAsync(h, census, 32, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    printf("grid %4d: workgroups per XCC %u %u %u %u %u %u %u %u\n", nwg, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 4096, s));
        CK(hipMemsetAsync(payload, 0, static_cast<size_t>(nwg) * 256 * 4, s));
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(256), 0, s, ctr, census, payload, rounds, mode, nwg);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        unsigned fails = 0; CK(hipMemcpy(&fails, ctr + 1000, 4, hipMemcpyDeviceToHost));
        if (fails) printf("  (!) %u bounded spins gave up: numbers below are not a barrier cost\n", fails);
      }
      printf("  %-8s: %.2f us per hand-off (%d rounds, whole launch %.1f us)\n",
             mode == 0 ? "flat" : mode == 1 ? "xcd" : "xcdonly", best * 1e3f / rounds, rounds, best * 1e3f);
    }
  }
  // reference: a chain of dependent trivial kernels on the same stream
  CK(hipEventRecord(a, s));
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(census_kernel, dim3(256), dim3(256), 0, s, census);
  CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("dependent kernel boundary (400 trivial 256-WG launches): %.2f us each\n", ms * 1e3f / 400);
  return 0;
}

// Can the HBM idle time of the decode step's small latency-bound kernels be used to pull the NEXT attention launch's
// K/V rows into the 256 MB Infinity Cache?  Four questions, one binary (decode attention through the C ABI at the
// bench shape B = 256, H = 6, bf16 caches with capacity 1024):
//   1. how fast is the attention kernel when its rows are already cache-resident (one set re-used) vs streamed (4 sets)
//   2. prefetch(set i) ; attention(set i) on one stream: what the pair costs against its two parts
//   3. the prefetch on a second stream, released by a device flag, next to a chain of five ~3.5 us spin kernels
//   4. the prefetch as extra workgroups of those five kernels (no second stream, no flag)
//   hipcc --offload-arch=gfx950 -O2 -I include tools/micro/mall_probe.hip -L mt3_amd -lmt3hip \
//         -Wl,-rpath,'$ORIGIN/../../mt3_amd' -o build/micro/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mt3_hip.h"

namespace {

constexpr int B = 256, H = 6, CAP = 1024, L = 4;
constexpr size_t kRowBytes = 128;                                    // 64 bf16
using i32x4 = __attribute__((ext_vector_type(4))) int;

// Reads the first `keys` rows of every (b, h) slab of K and V, 16 bytes per lane, nothing kept.
// part / parts: which slice of the (b, h) slabs this launch covers.  gate: spin until *flag >= want (bounded).
__device__ __forceinline__ void prefetch_body(const char* k, const char* v, int keys, int wg, int nwg, int part, int parts,
                                              int* sink, bool nt = false) {
  const int slabs = B * H;
  const int s0 = slabs * part / parts, s1 = slabs * (part + 1) / parts;
  const int per_slab = keys * int(kRowBytes) / 16;                   // 16-byte pieces per slab and tensor
  const long total = long(s1 - s0) * per_slab;
  int acc = 0;
  for (long i = long(wg) * blockDim.x + threadIdx.x; i < total; i += long(nwg) * blockDim.x) {
    const long slab = s0 + i / per_slab, off = (i % per_slab) * 16;
    const i32x4* pk = reinterpret_cast<const i32x4*>(k + slab * CAP * kRowBytes + off);
    const i32x4* pv = reinterpret_cast<const i32x4*>(v + slab * CAP * kRowBytes + off);
    const i32x4 a = nt ? __builtin_nontemporal_load(pk) : *pk;
    const i32x4 b = nt ? __builtin_nontemporal_load(pv) : *pv;
    acc ^= a.x ^ b.y;
  }
  if (acc == 0x12345678) *sink = acc;
}

__global__ void __launch_bounds__(256) prefetch_kernel(const char* k, const char* v, int keys, int part, int parts,
                                                       const int* flag, int want, int* sink, bool nt) {
  if (flag) {
    int budget = 1 << 16;                                            // bounded: ~10 ms at worst, never a hang
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && --budget > 0)
      __builtin_amdgcn_s_sleep(8);
  }
  prefetch_body(k, v, keys, blockIdx.x, gridDim.x, part, parts, sink, nt);
}

// A stand-in for one small decode GEMM: `spin_wgs` workgroups that sit for `ticks` of the 100 MHz clock; the first
// one bumps the flag; workgroups beyond spin_wgs (if any) prefetch.
__global__ void __launch_bounds__(256) spin_kernel(int spin_wgs, long ticks, int* flag, int set_flag, const char* k,
                                                   const char* v, int keys, int part, int parts, int* sink) {
  if (int(blockIdx.x) < spin_wgs) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && set_flag)
      __hip_atomic_store(flag, set_flag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  } else {
    prefetch_body(k, v, keys, blockIdx.x - spin_wgs, gridDim.x - spin_wgs, part, parts, sink);
  }
}

}  // namespace

int main() {
  void *kc[L], *vc[L];
  for (int l = 0; l < L; ++l) {
    hipMalloc(&kc[l], size_t(B) * H * CAP * kRowBytes); hipMalloc(&vc[l], size_t(B) * H * CAP * kRowBytes);
    hipMemset(kc[l], 0x3c, size_t(B) * H * CAP * kRowBytes); hipMemset(vc[l], 0x3c, size_t(B) * H * CAP * kRowBytes);
  }
  void *qkv, *out; int *step, *flag, *sink;
  hipMalloc(&qkv, size_t(B) * 3 * H * 64 * 2); hipMemset(qkv, 0x3c, size_t(B) * 3 * H * 64 * 2);
  hipMalloc(&out, size_t(B) * H * 64 * 2); hipMalloc(&step, B * 4); hipMalloc(&flag, 4); hipMalloc(&sink, 4);
  hipStream_t s, s2; hipStreamCreate(&s); hipStreamCreate(&s2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* q = static_cast<const char*>(qkv);
  const int reps = 40;
  auto attn = [&](int l, int keys) {
    return mt3_op_decode_attention(MT3_BF16, q, 3 * H * 64, kc[l], vc[l], CAP, q + H * 64 * 2, q + 2 * H * 64 * 2, 3 * H * 64,
                                   step, keys, out, B, H, s);
  };
  auto pref = [&](hipStream_t st, int l, int keys, int wgs, int part, int parts, const int* fl, int want, bool nt = false) {
    prefetch_kernel<<<wgs, 256, 0, st>>>(static_cast<const char*>(kc[l]), static_cast<const char*>(vc[l]), keys, part, parts,
                                         fl, want, sink, nt);
  };
  auto timed = [&](auto&& body) {
    for (int i = 0; i < 8; ++i) body(i);
    hipDeviceSynchronize();
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) body(8 + i);
    hipEventRecord(e1, s);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return double(ms) * 1e3 / reps;
  };
  for (int keys : {129, 257, 513}) {
    std::vector<int> hs(B, keys - 1);
    hipMemcpy(step, hs.data(), B * 4, hipMemcpyHostToDevice);
    const double mb = double(B) * H * keys * 2 * kRowBytes / 1e6;
    if (attn(0, keys)) { printf("error: %s\n", mt3_last_error()); return 1; }
    const double t_stream = timed([&](int i) { attn(i % L, keys); });
    const double t_res = timed([&](int i) { attn(0, keys); });
    printf("1. keys %4d (%6.1f MB): streamed (4 sets) %6.2f us = %5.0f GB/s   resident (1 set) %6.2f us = %5.0f GB/s\n", keys,
           mb, t_stream, mb / t_stream * 1e3, t_res, mb / t_res * 1e3);
    for (int v = 0; v < 4; ++v) {
      const int wgs = v & 1 ? 2048 : 512;
      const bool nt = v & 2;
      const double t_pref = timed([&](int i) { pref(s, i % L, keys, wgs, 0, 1, nullptr, 0, nt); });
      const double t_pair = timed([&](int i) { pref(s, i % L, keys, wgs, 0, 1, nullptr, 0, nt); attn(i % L, keys); });
      printf("2. keys %4d  prefetch alone (%4d wgs%s) %6.2f us = %5.0f GB/s   prefetch ; attention %6.2f us  -> attention after "
             "prefetch %6.2f us\n", keys, wgs, nt ? ", nt" : "", t_pref, mb / t_pref * 1e3, t_pair, t_pair - t_pref);
    }
    // 3./4. five stand-in small kernels then the attention
    const long ticks = 350;                                          // 3.5 us at 100 MHz
    const double t_chain = timed([&](int i) {
      for (int j = 0; j < 5; ++j) spin_kernel<<<32, 256, 0, s>>>(32, ticks, flag, 0, nullptr, nullptr, 0, 0, 1, sink);
    });
    const double t_base = timed([&](int i) {
      for (int j = 0; j < 5; ++j) spin_kernel<<<32, 256, 0, s>>>(32, ticks, flag, 0, nullptr, nullptr, 0, 0, 1, sink);
      attn(i % L, keys);
    });
    printf("3. keys %4d  five spin kernels %6.2f us; + attention %6.2f us\n", keys, t_chain, t_base);
    for (int frac : {4, 2, 1}) {                                     // prefetch 1/4, 1/2, all of the slabs
      hipMemset(flag, 0, 4);
      int serial = 0;
      const double t_two = timed([&](int i) {
        ++serial;
        pref(s2, i % L, keys, 512, 0, frac, flag, serial);
        for (int j = 0; j < 5; ++j)
          spin_kernel<<<32, 256, 0, s>>>(32, ticks, flag, j == 0 ? serial : 0, nullptr, nullptr, 0, 0, 1, sink);
        attn(i % L, keys);
      });
      const double t_piggy = timed([&](int i) {
        for (int j = 0; j < 5; ++j)
          spin_kernel<<<32 + 480, 256, 0, s>>>(32, ticks, flag, 0, static_cast<const char*>(kc[i % L]),
                                               static_cast<const char*>(vc[i % L]), keys, j, 5 * frac, sink);
        attn(i % L, keys);
      });
      printf("   keys %4d  prefetching 1/%d of the rows: second stream %6.2f us   extra workgroups %6.2f us   (no prefetch %6.2f)\n",
             keys, frac, t_two, t_piggy, t_base);
    }
  }
  return 0;
}

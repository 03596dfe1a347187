// Which compute units does bit i of a hipExtStreamCreateWithCUMask mask name?  (Groundwork for placing attention and dense
// launches on different CUs inside one queue, DESIGN.md section 7 a iii: a kernel can only find out where it runs from
// HW_ID / XCC_ID, so the reserved set has to be described in those terms.)  Launches many one-wave workgroups on streams
// whose masks hold 32 consecutive bits each and prints, per mask range, the (xcc, se, cu) triples the workgroups reported.
// hipcc --offload-arch=gfx950 -O3 cu_mask_map.hip -o cu_mask_map
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <set>
#include <vector>
#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

__global__ void k_where(unsigned* hw, unsigned* xcc) {
  // s_getreg_b32: simm16 = id | offset << 6 | (size - 1) << 11;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20 (gfx940+)
  const unsigned h = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  const unsigned x = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  // stay a little so that the dispatcher has to spread the grid
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) {
    hw[blockIdx.x] = h;
    xcc[blockIdx.x] = x;
  }
}

int main() {
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  constexpr int kGrid = 4096;
  unsigned *d_hw, *d_xcc;
  CK(hipMalloc(&d_hw, kGrid * 4));
  CK(hipMalloc(&d_xcc, kGrid * 4));
  std::vector<unsigned> hw(kGrid), xcc(kGrid);
  printf("%d CUs; HW_ID fields printed as xcc.se.cu (xcc = XCC_ID[3:0], se = HW_ID[15:13], cu = HW_ID[11:8]; sh = HW_ID[12] folded into cu as +16)\n", n_cu);
  for (int lo = -32; lo < n_cu; lo += 32) {
    std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
    if (lo < 0) {
      for (int i = 0; i < n_cu; ++i) mask[i >> 5] |= 1u << (i & 31);
    } else {
      for (int i = lo; i < lo + 32 && i < n_cu; ++i) mask[i >> 5] |= 1u << (i & 31);
    }
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>(mask.size()), mask.data()));
    hipLaunchKernelGGL(k_where, dim3(kGrid), dim3(64), 0, s, d_hw, d_xcc);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hw.data(), d_hw, kGrid * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(xcc.data(), d_xcc, kGrid * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> seen, xs;
    for (int i = 0; i < kGrid; ++i) {
      const unsigned x = xcc[i] & 0xF, se = (hw[i] >> 13) & 0x7, cu = ((hw[i] >> 8) & 0xF) + 16 * ((hw[i] >> 12) & 1);
      seen.insert((x << 16) | (se << 8) | cu);
      xs.insert(x);
    }
    if (lo < 0) printf("all bits: %zu distinct (xcc, se, cu) on %zu XCCs\n", seen.size(), xs.size());
    else {
      printf("bits %3d..%3d: %3zu distinct CUs on %zu XCC(s):", lo, lo + 31, seen.size(), xs.size());
      int n = 0;
      for (unsigned v : seen)
        if (n++ < 40) printf(" %u.%u.%u", v >> 16, (v >> 8) & 0xFF, v & 0xFF);
      printf("\n");
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}

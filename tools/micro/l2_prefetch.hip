// Microbenchmark (round 5; VERDICT r4 "next" #5 i): can the tail of an HBM-streaming launch
// (decode attention) pull the NEXT launch's f32 weight columns from the Infinity Cache into the L2 of the XCD that will
// read them, so that the latency-bound dense launch behind it finds L2 hits instead of MALL hits?
//
// The decode step in miniature, one row group: per "layer" a streaming kernel S (384 workgroups, non-temporal reads of 50 MB,
// like the cross-attention launch of a 64-row group) followed by a dense-like kernel G shaped like the fold launch
// (gemm_kernel<float, 32, 32, 256, ...>: 128 workgroups = 2 row tiles x 64 column tiles, each reading its 32-column
// weight tile [32][1536] f32 in SIX DEPENDENT K slices of 32 KB through LDS, column tiles dealt to XCDs in runs --
// GemmArgs::n_major -- so an XCD's L2 sees 1/8 of the 12.6 MB matrix).  Eight layers with their own matrices (100 MB:
// more than the 32 MB of L2, less than the 256 MB Infinity Cache), captured as one graph and replayed, so that a
// matrix is out of L2 and in the MALL when its launch comes round again -- the product's situation.
// Variants: S without / with a prefetch tail in which the workgroups of XCD x touch the lines of the weight columns XCD x
// will read (plain loads, results discarded).  Reported: us per S, per G (differential) and per layer.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_prefetch.hip -o build/micro/l2_prefetch
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                      \
    }                                                                \
  } while (0)

constexpr int kN = 2048, kK = 1536, kSlice = 256, kTileN = 32, kRowTiles = 2;
constexpr int kColTiles = kN / kTileN;   // 64
constexpr int kXcds = 8;

// column tile of workgroup b: XCD (b % 8) owns the run of 8 column tiles [8 * xcd, 8 * xcd + 8), both row tiles
__device__ __forceinline__ int col_tile_of(int b) {
  const int xcd = b % kXcds, k = b / kXcds;          // k = 0 .. 15: 8 column tiles x 2 row tiles
  return xcd * (kColTiles / kXcds) + (k % (kColTiles / kXcds));
}

// the workgroups of XCD x = blockIdx % 8 touch the lines of the weight rows XCD x will read in the next dense launch:
// column tiles [8x, 8x + 8) = rows [256 x, 256 x + 256) of W [N][K], 1.57 MB = 12288 lines of 128 bytes, shared by the
// gridDim / 8 workgroups of the XCD (plain loads, never waited for, never used: the line lands in this XCD's L2)
__device__ __forceinline__ void prefetch_next(const float* w, int nthreads) {
  const int xcd = blockIdx.x % kXcds, j = blockIdx.x / kXcds, per_xcd = gridDim.x / kXcds;
  const char* base = reinterpret_cast<const char*>(w) + static_cast<size_t>(xcd) * (kColTiles / kXcds) * kTileN * kK * 4;
  const int lines = (kColTiles / kXcds) * kTileN * kK * 4 / 128;
  const int share = (lines + per_xcd - 1) / per_xcd;
  unsigned acc = 0;
  for (int l = j * share + threadIdx.x; l < (j + 1) * share && l < lines; l += nthreads)
    acc += *reinterpret_cast<const unsigned*>(base + static_cast<size_t>(l) * 128);
  // ordinary loads kept alive by an empty asm (an inline-asm load with a dead output register is NOT safe: the compiler
  // hands the register to the next live value while the load is still in flight)
  asm volatile("" ::"v"(acc));
}

__global__ __launch_bounds__(256) void dense_like(const float* __restrict__ W, float* __restrict__ out,
                                                  const float* __restrict__ prefetch_w) {
  __shared__ float tile[kTileN * kSlice];            // 32 KB
  const int ct = col_tile_of(blockIdx.x);
  const float* w = W + static_cast<size_t>(ct) * kTileN * kK;
  float acc = 0.f;
  for (int k0 = 0; k0 < kK; k0 += kSlice) {          // six dependent slices: load -> LDS -> barrier -> use
    for (int i = threadIdx.x; i < kTileN * kSlice / 4; i += 256) {
      const int row = (i * 4) / kSlice, col = (i * 4) % kSlice;
      *reinterpret_cast<float4*>(tile + row * kSlice + col) =
          *reinterpret_cast<const float4*>(w + static_cast<size_t>(row) * kK + k0 + col);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kTileN * kSlice; i += 256 * 8) acc += tile[i];
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (prefetch_w) prefetch_next(prefetch_w, 256);    // a dense launch pulling the NEXT dense launch's matrix
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// streaming kernel: each workgroup reads `per_wg` 16-byte chunks with non-temporal loads (the K/V stream); with
// prefetch_w != nullptr its tail touches this XCD's share of the next dense launch's weight columns
__global__ __launch_bounds__(192) void stream_like(const u32x4* __restrict__ kv, size_t per_wg, float* __restrict__ out,
                                                   const float* __restrict__ prefetch_w, int at_start) {
  const u32x4* p = kv + static_cast<size_t>(blockIdx.x) * per_wg;
  unsigned acc = 0;
  if (prefetch_w && at_start) prefetch_next(prefetch_w, 192);
  for (size_t i = threadIdx.x; i < per_wg; i += 192 * 4) {
    u32x4 a = __builtin_nontemporal_load(p + i);
    u32x4 b = i + 192 < per_wg ? __builtin_nontemporal_load(p + i + 192) : a;
    u32x4 c = i + 384 < per_wg ? __builtin_nontemporal_load(p + i + 384) : a;
    u32x4 d = i + 576 < per_wg ? __builtin_nontemporal_load(p + i + 576) : a;
    acc += a.x + b.y + c.z + d.w;
  }
  if (prefetch_w && !at_start) prefetch_next(prefetch_w, 192);
  if (acc == 0x12345678u) out[blockIdx.x] = 1.f;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int layers = 8, s_wgs = 384;
  const size_t kv_bytes = 50u << 20, w_bytes = static_cast<size_t>(kN) * kK * 4;
  std::vector<float*> W(layers);
  for (auto& w : W) {
    CK(hipMalloc(&w, w_bytes));
    CK(hipMemset(w, 0, w_bytes));
  }
  u32x4* kv;
  CK(hipMalloc(&kv, kv_bytes * layers));
  CK(hipMemset(kv, 0, kv_bytes * layers));
  float* out;
  CK(hipMalloc(&out, 1 << 20));
  const size_t per_wg = kv_bytes / 16 / s_wgs;

  // mode 0: no prefetch; 1: the streaming launch's TAIL pulls the next dense launch's matrix; 2: its HEAD does;
  // 3: (dense-only chains) every dense launch's tail pulls the NEXT layer's matrix
  auto run = [&](int mode, bool with_dense, bool with_stream) -> float {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int l = 0; l < layers; ++l) {
      if (with_stream)
        hipLaunchKernelGGL(stream_like, dim3(s_wgs), dim3(192), 0, s, kv + static_cast<size_t>(l) * (kv_bytes / 16), per_wg, out,
                           (mode == 1 || mode == 2) ? W[l] : nullptr, mode == 2 ? 1 : 0);
      if (with_dense)
        hipLaunchKernelGGL(dense_like, dim3(kRowTiles * kColTiles), dim3(256), 0, s, W[l], out + 4096,
                           mode == 3 ? W[(l + 1) % layers] : nullptr);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int reps = 200;
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    return ms * 1e3f / (reps * layers);             // us per layer
  };
  const float s_only = run(0, false, true), s_tail = run(1, false, true), s_head = run(2, false, true);
  const float d_only = run(0, true, false), d_chain = run(3, true, false);
  const float both = run(0, true, true), both_tail = run(1, true, true), both_head = run(2, true, true);
  printf("per layer (8 layers x 12.6 MB f32 matrices = 100 MB: out of the L2s, inside the Infinity Cache; 50 MB streamed per layer)\n");
  printf("  streaming launch alone                   %.2f us   (prefetch in its tail: %.2f, in its head: %.2f)\n", s_only, s_tail, s_head);
  printf("  dense-like launches back to back         %.2f us   (each pulling the next one's matrix in its tail: %.2f)\n", d_only, d_chain);
  printf("  stream + dense                           %.2f us   -> dense part %.2f us\n", both, both - s_only);
  printf("  stream (prefetch in its tail) + dense    %.2f us   -> dense part %.2f us, pair %+.2f us\n", both_tail, both_tail - s_tail, both_tail - both);
  printf("  stream (prefetch in its head) + dense    %.2f us   -> dense part %.2f us, pair %+.2f us\n", both_head, both_head - s_head, both_head - both);
  return 0;
}

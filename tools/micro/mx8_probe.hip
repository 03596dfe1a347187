// The MXFP8 encoder GEMM (mt3k::gemm_mx8_kernel) outside Python: (1) correctness of the quantisers and of every
// epilogue against a double-precision product of the DEQUANTISED operands, (2) encoder-shaped timings (M = 65536).
// -DMT3_MX8_PROBE=n leaves parts out (1 no fragment reads / MFMAs, 2 no DMA beyond the prologue, 4 no epilogue),
// -DMT3_MX8_NS=2|3 picks the ring depth.  Build: tools/micro/build_glds_probe.sh (cross-compiles in the CPU container).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "gemm_mx8.hip"

namespace {

float e4m3_value(uint8_t b) {
  const int e = (b >> 3) & 15, m = b & 7;
  const float v = e == 0 ? std::ldexp(float(m) / 8.f, -6) : std::ldexp(1.f + float(m) / 8.f, e - 7);
  return (b & 0x80) ? -v : v;
}
float e8m0_value(uint8_t b) { return std::ldexp(1.f, int(b) - 127); }
void dequant(const std::vector<uint8_t>& q, const std::vector<uint8_t>& sc, int rows, int K, std::vector<float>& out) {
  out.resize(size_t(rows) * K);
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) out[size_t(r) * K + k] = e4m3_value(q[size_t(r) * K + k]) * e8m0_value(sc[size_t(r) * (K / 32) + k / 32]);
}
template <typename T>
T* dev(const std::vector<T>& h) {
  T* d;
  hipMalloc(&d, h.size() * sizeof(T));
  hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
template <typename T>
std::vector<T> host(const T* d, size_t n) {
  std::vector<T> h(n);
  hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
  return h;
}
float bf16_value(uint16_t b) {
  uint32_t u = uint32_t(b) << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// the scaled MFMA aligns its 128 products before adding: 1.2-1.8e-4 of sum |products| measured (mfma_scale_check.hip)
constexpr double kMfmaTol = 2.5e-4;
double gelu(double x) { return 0.5 * x * (1.0 + std::tanh(0.7978845608028654 * (x + 0.044715 * x * x * x))); }

int check(int M, int N, int K) {
  std::mt19937 rng(M * 31 + N * 7 + K);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> a(size_t(M) * K), w(size_t(N) * K), x0(size_t(M) * N);
  for (int r = 0; r < M; ++r) {
    const float row_gain = std::exp(nd(rng));                        // rows of very different magnitude
    for (int k = 0; k < K; ++k) a[size_t(r) * K + k] = nd(rng) * row_gain * (k % 97 == 3 ? 30.f : 1.f);   // + outlier columns
  }
  for (auto& v : w) v = nd(rng) * 0.05f;
  for (auto& v : x0) v = nd(rng);
  std::vector<uint8_t> wq(w.size()), wsc(size_t(N) * K / 32), aq_h(a.size()), asc_h(size_t(M) * K / 32);
  mt3_host_mx8_quantize(w.data(), N, K, wq.data(), wsc.data());
  mt3_host_mx8_quantize(a.data(), M, K, aq_h.data(), asc_h.data());
  float* d_a = dev(a);
  uint8_t *d_aq, *d_asc, *d_wq = dev(wq), *d_wsc = dev(wsc);
  float* d_ss;
  hipMalloc(&d_aq, a.size()); hipMalloc(&d_asc, size_t(M) * K / 32); hipMalloc(&d_ss, size_t(M) * (K / 16) * 4);
  int bad = 0;
  if (mt3_op_mx8_quantize(d_a, 1, M, K, d_aq, d_asc, d_ss, nullptr)) { printf("quantize: %s\n", mt3_last_error()); return 1; }
  const auto aq = host(d_aq, a.size());
  const auto asc = host(d_asc, size_t(M) * K / 32);
  const auto ss = host(d_ss, size_t(M) * (K / 16));
  size_t dq = 0, ds = 0;
  for (size_t i = 0; i < aq.size(); ++i) dq += e4m3_value(aq[i]) != e4m3_value(aq_h[i]);
  for (size_t i = 0; i < asc.size(); ++i) ds += asc[i] != asc_h[i];
  double ss_err = 0;
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < K / 16; ++c) {
      double t = 0;
      for (int j = 0; j < 16; ++j) t += double(a[size_t(r) * K + c * 16 + j]) * a[size_t(r) * K + c * 16 + j];
      ss_err = std::fmax(ss_err, std::fabs(ss[size_t(r) * (K / 16) + c] - t) / (t + 1e-30));
    }
  printf("M %d N %d K %d: device vs host quantiser: %zu element / %zu scale mismatches; partial sums rel err %.2e\n", M, N, K, dq,
         ds, ss_err);
  bad += dq != 0 || ds != 0 || ss_err > 1e-6;
  std::vector<float> ad, wd;
  dequant(aq, asc, M, K, ad);
  dequant(wq, wsc, N, K, wd);
  std::vector<double> prod(size_t(M) * N), mag(size_t(M) * N);
  for (int r = 0; r < M; ++r)
    for (int n = 0; n < N; ++n) {
      double t = 0, m = 0;
      for (int k = 0; k < K; ++k) {
        const double p = double(ad[size_t(r) * K + k]) * wd[size_t(n) * K + k];
        t += p;
        m += std::fabs(p);
      }
      prod[size_t(r) * N + n] = t;
      mag[size_t(r) * N + n] = m;
    }
  std::vector<double> rs(M);
  for (int r = 0; r < M; ++r) {
    double t = 0;
    for (int k = 0; k < K; ++k) t += double(a[size_t(r) * K + k]) * a[size_t(r) * K + k];
    rs[r] = 1.0 / std::sqrt(t / K + 1e-6);
  }
  // STORE with the fused RMSNorm, bf16 out
  uint16_t* d_o;
  hipMalloc(&d_o, size_t(M) * N * 2);
  if (mt3_op_gemm_mx8(d_aq, d_asc, d_wq, d_wsc, d_o, M, N, K, MT3_EPI_STORE, 0, d_ss, nullptr, nullptr, nullptr, nullptr)) {
    printf("store: %s\n", mt3_last_error());
    return 1;
  }
  {
    const auto o = host(d_o, size_t(M) * N);
    double worst = 0;
    for (size_t i = 0; i < o.size(); ++i) {
      const double want = prod[i] * rs[i / N], tol = std::fabs(want) / 256 + mag[i] * rs[i / N] * kMfmaTol;
      worst = std::fmax(worst, std::fabs(bf16_value(o[i]) - want) / tol);
    }
    printf("  STORE (fused norm, bf16): worst error / tolerance (bf16 half-ulp + 2.5e-4 of the magnitude sum) = %.3f\n", worst);
    bad += !(worst <= 1.0);
  }
  // HEADS (no norm): [2][B][H][seq][64]
  if (N % 128 == 0 && M % 64 == 0) {
    const int seq = 64, H = N / 128, B = M / seq;
    hipMemset(d_o, 0, size_t(M) * N * 2);
    if (mt3_op_gemm_mx8(d_aq, d_asc, d_wq, d_wsc, d_o, M, N, K, MT3_EPI_HEADS, seq, nullptr, nullptr, nullptr, nullptr, nullptr)) {
      printf("heads: %s\n", mt3_last_error());
      return 1;
    }
    const auto o = host(d_o, size_t(M) * N);
    double worst = 0;
    for (int r = 0; r < M; ++r)
      for (int n = 0; n < N; ++n) {
        const int kv = n / (H * 64), hh = (n % (H * 64)) / 64, d = n % 64, bb = r / seq, tt = r % seq;
        const size_t at = ((((size_t(kv) * B + bb) * H + hh) * seq) + tt) * 64 + d;
        const double want = prod[size_t(r) * N + n], tol = std::fabs(want) / 256 + mag[size_t(r) * N + n] * kMfmaTol;
        worst = std::fmax(worst, std::fabs(bf16_value(o[at]) - want) / tol);
      }
    printf("  HEADS: worst error / tolerance = %.3f\n", worst);
    bad += !(worst <= 1.0);
  }
  // RESID: x += product; MXFP8 copy + partial sums of the new rows
  {
    float* d_x = dev(x0);
    uint8_t *d_xq, *d_xsc;
    float* d_xss;
    hipMalloc(&d_xq, size_t(M) * N); hipMalloc(&d_xsc, size_t(M) * N / 32); hipMalloc(&d_xss, size_t(M) * (N / 16) * 4);
    if (mt3_op_gemm_mx8(d_aq, d_asc, d_wq, d_wsc, d_x, M, N, K, MT3_EPI_RESID, 0, nullptr, d_xq, d_xsc, d_xss, nullptr)) {
      printf("resid: %s\n", mt3_last_error());
      return 1;
    }
    const auto x = host(d_x, size_t(M) * N);
    const auto xq = host(d_xq, size_t(M) * N);
    const auto xsc = host(d_xsc, size_t(M) * N / 32);
    const auto xss = host(d_xss, size_t(M) * (N / 16));
    double worst = 0, sse = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      const double want = x0[i] + prod[i], tol = std::fabs(want) * 2e-7 + mag[i] * kMfmaTol + 1e-7;
      worst = std::fmax(worst, std::fabs(x[i] - want) / tol);
    }
    std::vector<uint8_t> q2(x.size()), sc2(xsc.size());
    mt3_host_mx8_quantize(x.data(), M, N, q2.data(), sc2.data());
    size_t mq = 0, ms = 0;
    for (size_t i = 0; i < xq.size(); ++i) mq += e4m3_value(xq[i]) != e4m3_value(q2[i]);
    for (size_t i = 0; i < xsc.size(); ++i) ms += xsc[i] != sc2[i];
    for (int r = 0; r < M; ++r)
      for (int c = 0; c < N / 16; ++c) {
        double t = 0;
        for (int j = 0; j < 16; ++j) t += double(x[size_t(r) * N + c * 16 + j]) * x[size_t(r) * N + c * 16 + j];
        sse = std::fmax(sse, std::fabs(xss[size_t(r) * (N / 16) + c] - t) / (t + 1e-30));
      }
    printf("  RESID: worst error / tolerance = %.3f; MXFP8 copy of the new rows vs host quantiser: %zu / %zu mismatches; sums rel err %.2e\n",
           worst, mq, ms, sse);
    bad += !(worst <= 1.0) || mq || ms || sse > 1e-6;
  }
  // GEGLU (fused norm): rows of W interleaved gate / linear in 16s; output only as MXFP8
  if (N % 256 == 0) {
    uint8_t *d_hq, *d_hsc;
    hipMalloc(&d_hq, size_t(M) * N / 2); hipMalloc(&d_hsc, size_t(M) * N / 64);
    if (mt3_op_gemm_mx8(d_aq, d_asc, d_wq, d_wsc, nullptr, M, N, K, MT3_EPI_GEGLU, 0, d_ss, d_hq, d_hsc, nullptr, nullptr)) {
      printf("geglu: %s\n", mt3_last_error());
      return 1;
    }
    const auto hq = host(d_hq, size_t(M) * N / 2);
    const auto hsc = host(d_hsc, size_t(M) * N / 64);
    std::vector<float> hd, want(size_t(M) * N / 2);
    dequant(hq, hsc, M, N / 2, hd);
    for (int r = 0; r < M; ++r)
      for (int u = 0; u < N / 2; ++u) {
        const int q = u / 16, j = u % 16;
        want[size_t(r) * (N / 2) + u] = float(gelu(prod[size_t(r) * N + 32 * q + j] * rs[r]) * (prod[size_t(r) * N + 32 * q + 16 + j] * rs[r]));
      }
    std::vector<uint8_t> q2(want.size()), sc2(hsc.size());
    mt3_host_mx8_quantize(want.data(), M, N / 2, q2.data(), sc2.data());
    size_t ms = 0, exact = 0;
    double worst = 0;
    for (size_t i = 0; i < hsc.size(); ++i) ms += hsc[i] != sc2[i];
    for (int r = 0; r < M; ++r)
      for (int u = 0; u < N / 2; ++u) {
        const size_t i = size_t(r) * (N / 2) + u;
        const double step = std::fmax(std::fabs(want[i]) / 8, e8m0_value(hsc[size_t(r) * (N / 64) + u / 32]) / 512);   // one e4m3 step
        worst = std::fmax(worst, std::fabs(hd[i] - want[i]) / step);
        exact += e4m3_value(hq[i]) == e4m3_value(q2[i]);
      }
    printf("  GEGLU: worst |dequantised - exact| in e4m3 steps = %.3f (0.5 + what the MFMA tolerance moves near zero crossings); "
           "%zu of %zu bytes equal the quantised exact result; %zu of %zu block scales differ\n", worst, exact, hq.size(), ms, hsc.size());
    bad += exact < hq.size() * 99 / 100 || ms > hsc.size() / 50;
  }
  return bad;
}

}  // namespace

int main(int argc, char** argv) {
  int bad = 0;
  if (!(MT3_MX8_PROBE)) {
    bad += check(256, 512, 512);
    bad += check(384, 256, 384);
    bad += check(200, 768, 1024);      // M not a multiple of the tile
    printf(bad ? "CHECKS FAILED (%d)\n" : "checks ok\n", bad);
  }
  if (argc > 1 && argv[1][0] == 'c') return bad;
  struct Case { const char* name; int N, K, epi; bool norm; };
  const Case cases[] = {{"qkv   N=1152 K=512  STORE", 1152, 512, MT3_EPI_STORE, true},
                        {"geglu N=2048 K=512  GEGLU", 2048, 512, MT3_EPI_GEGLU, true},
                        {"out   N=512  K=384  RESID", 512, 384, MT3_EPI_RESID, false},
                        {"wo    N=512  K=1024 RESID", 512, 1024, MT3_EPI_RESID, false},
                        {"kv    N=768  K=512  HEADS", 768, 512, MT3_EPI_HEADS, false}};
  const int M = 65536;
  uint8_t *A, *Asc, *W, *Wsc, *Oq, *Osc;
  void* O;
  float *ss, *oss;
  hipMalloc(&A, size_t(M) * 1024); hipMalloc(&Asc, size_t(M) * 32); hipMalloc(&W, 2048 * 1024); hipMalloc(&Wsc, 2048 * 32);
  hipMalloc(&O, size_t(M) * 2048 * 4); hipMalloc(&Oq, size_t(M) * 1024); hipMalloc(&Osc, size_t(M) * 32);
  hipMalloc(&ss, size_t(M) * 64 * 4); hipMalloc(&oss, size_t(M) * 64 * 4);
  hipMemset(A, 0x38, size_t(M) * 1024); hipMemset(Asc, 127, size_t(M) * 32); hipMemset(W, 0x30, 2048 * 1024); hipMemset(Wsc, 120, 2048 * 32);
  hipMemset(O, 0, size_t(M) * 2048 * 4); hipMemset(ss, 0, size_t(M) * 64 * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Case& c : cases) {
    auto run = [&]() {
      return mt3_op_gemm_mx8(A, Asc, W, Wsc, O, M, c.N, c.K, c.epi, 256, c.norm ? ss : nullptr, Oq, Osc, oss, s);
    };
    if (run()) { printf("%s: %s\n", c.name, mt3_last_error()); return 1; }
    for (int i = 0; i < 2; ++i) run();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * M * c.N * c.K / (us * 1e-6) / 1e12;
    printf("probe %d ns %d  %-28s %8.1f us per launch  (%6.0f TF/s if it were the whole GEMM)\n", MT3_MX8_PROBE, MT3_MX8_NS, c.name, us, tf);
  }
  // the activation quantiser at the attention-output shape
  {
    void* src;
    hipMalloc(&src, size_t(M) * 512 * 4);
    hipMemset(src, 0x3c, size_t(M) * 512 * 4);
    for (int f32 = 0; f32 < 2; ++f32) {
      const int K = f32 ? 512 : 384;
      for (int i = 0; i < 2; ++i) mt3_op_mx8_quantize(src, f32, M, K, Oq, Osc, f32 ? oss : nullptr, s);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      for (int i = 0; i < 10; ++i) mt3_op_mx8_quantize(src, f32, M, K, Oq, Osc, f32 ? oss : nullptr, s);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("quantise %s [65536][%d]: %.1f us\n", f32 ? "f32 (+ sums)" : "bf16", K, ms * 100);
    }
  }
  return bad;
}

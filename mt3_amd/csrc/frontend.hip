// Fused log-mel frontend for gfx950: framing + Hann + rFFT-2048 + |.| + HTK mel + safe log
// in ONE kernel, one pass over HBM.
//
// Replaces (per segment) spectral_ops.compute_logmel (mt3/spectral_ops.py:76-88) =
// stft (:35-48) -> compute_mag (:52-54) -> compute_mel (:58-73) -> safe_log (:29-32),
// called through spectrograms.compute_spectrogram (mt3/spectrograms.py:64-73).
//
// Layout / roofline: HBM-bound by design.  Algorithmic bytes per segment =
// 32768*4 (audio in) + 256*512*4 (log-mel out) = 655,360 B.
//   * hop 128 / window 2048 = 16x sample reuse: a workgroup stages the
//     G*128 + 1920 samples of a G-frame tile into LDS ONCE (coalesced float4)
//     instead of materialising overlapping frames in HBM (2.1 MB/segment).
//   * each of the 4 waves owns a frame at a time: 16 points per lane, radix
//     16 x 4 x 16 complex FFT-1024 with the exchange in a padded LDS buffer
//     (row stride 65 -> conflict-free 16-point gathers), twiddles/window in VGPRs.
//   * mel projection in its band-sparse form (1934 non-zeros, <= 10 per mel bin):
//     ~1 MFLOP per segment instead of the 269 MFLOP dense [1025x512] product, which
//     on the fp32 matrix pipe (157 TF peak) would cost 17x the kernel's HBM time.
//   * output rows of short segments (frame >= n_frames) are written as 0.0: the
//     reference zero-pads AFTER the log (mt3/models.py:48-98).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"
#include "frontend_core.h"
#include "frontend_tables.h"
#include "mt3_hip.h"

namespace {

using mt3fe::cpx;

constexpr int kFramesPerBlock = 16;                          // frames_per_segment must be a multiple of this
constexpr int kFramesPerWave = 4;                            // a wave walks 4 frames of its workgroup's tile
constexpr int kHop = 128;
constexpr int kMelBins = 512;
constexpr int kMagStride = 1028;
constexpr int kMaxBandWeights = 2048;       // >= sum of band lengths (1934 for the reference's mel matrix)
// logf(1e-5f) as numpy's float32 evaluates it (spectral_ops.safe_log's floor), and ln 2
constexpr float kLogFloor = -11.512925148010254f, kLn2 = 0.6931471805599453f;

// safe_log of a mel value: v_log_f32 (log2, 1 ulp) * ln 2 instead of logf's expansion; denormal inputs (which the
// instruction would flush) are clamped
__device__ __forceinline__ float safe_log_fast(float m) {
  // (a positive DENORMAL mel -- below 1.2e-38, i.e. numerical dust of a silent frame -- is read as the smallest
  // normal number: v_log_f32 would flush it to zero)
  const float l = __builtin_amdgcn_logf(fmaxf(m, 1.17549435e-38f)) * kLn2;
  return m <= 0.f ? kLogFloor : l;
}

template <int I>
__device__ __forceinline__ void mel_group(const int* k0, const float* wpad, int lane, const float* mag, float* dst) {
  const int j = lane + 64 * I;
  dst[j] = safe_log_fast(mt3fe::mel_bin_padded<mt3fe::kGroupMaxBand[I]>(k0, wpad + mt3fe::group_base(I), j, mag));
}

// The 4 waves of a workgroup work on different frames and only meet at the initial staging barrier;
// inside the frame loop every LDS hand-off is between lanes of ONE wave (private xchg / mag regions).
// A wave executes its LDS instructions in order, so it suffices to (a) wait for the wave's own
// outstanding LDS writes and (b) stop the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct FrontendDev {
  const float* hann;
  const cpx* tw1024;
  const cpx* tw2048;
  const int* k0;
  const int* cnt;
  const int* off;
  const float* w;
  int n_w;
};

// kWaves = 4 (product): 16-frame tiles (3,968 staged samples: the 1,920-sample halo is re-read by every tile, 1.94x on
// the input side, PMC traffic 1.178x algorithmic).  kWaves = 8 (round 3, opt-in debug knob): 32-frame tiles -- halo and
// constant tables staged half as often, PMC traffic 1.091x, the same 8 waves per CU (135 KB of LDS: ONE 512-thread
// workgroup per CU instead of two of 256) -- measured 17-19 % SLOWER (186.9 against 156.0 us per 256 segments): the
// kernel is latency-bound, not HBM-bound, and a single workgroup per CU has nobody to overlap its staging barrier and
// its prologue with.  Outputs are bit-identical.
// Per-segment frame counts that the caller holds in HOST memory travel as a kernel ARGUMENT (round 4): 1024 counts of 16
// bits per launch, calls with more segments are issued in pieces.  (Rounds 2-3 copied them into a ring of device slots
// behind a mutex and waited for the whole device when the ring wrapped -- a hidden synchronisation that also broke stream
// capture; an argument needs no buffer, no lifetime and no lock.)
constexpr int kCountsPerLaunch = 1024;
struct FrameCounts {
  uint16_t n[kCountsPerLaunch];
};

template <int kWaves>
__global__ __launch_bounds__(kWaves * 64) void logmel_kernel(FrontendDev t, const float* __restrict__ audio,
                                                             const int* __restrict__ n_frames, int frames_per_segment,
                                                             float* __restrict__ out, int use_counts, FrameCounts counts) {
  constexpr int kTileFrames = kWaves * kFramesPerWave;                        // G
  constexpr int kTileSamples = kTileFrames * kHop + (mt3fe::kFft - kHop);     // 3968 / 6016
  constexpr int kThreads = kWaves * 64;
  __shared__ __attribute__((aligned(16))) float s_samples[kTileSamples];
  __shared__ __attribute__((aligned(16))) cpx s_xchg[kWaves][mt3fe::kXchg];   // also holds Z in natural order
  __shared__ __attribute__((aligned(16))) float s_mag[kWaves][kMagStride];
  __shared__ int s_k0[kMelBins];                                        // first spectrum bin of every mel band
  __shared__ float s_w[mt3fe::kPaddedWeights];                                 // group-padded, transposed band weights

  const int tiles = frames_per_segment / kTileFrames;
  const int seg = blockIdx.x / tiles;
  const int f0 = (blockIdx.x % tiles) * kTileFrames;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = use_counts ? static_cast<int>(counts.n[seg]) : (n_frames ? n_frames[seg] : frames_per_segment);
  const int valid = n * kHop;                                    // samples of this segment that exist
  const float* seg_audio = audio + static_cast<size_t>(seg) * frames_per_segment * kHop;

  // stage the tile's samples once (16x reuse); zeros past the end of the segment (pad_end=True)
  for (int i = tid; i < kTileSamples / 4; i += kThreads) {
    const int idx = f0 * kHop + 4 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < valid) v = *reinterpret_cast<const float4*>(seg_audio + idx);
    *reinterpret_cast<float4*>(&s_samples[4 * i]) = v;
  }
  for (int i = tid; i < kMelBins; i += kThreads) s_k0[i] = t.k0[i];
  for (int i = tid; i < mt3fe::kPaddedWeights; i += kThreads) s_w[i] = t.w[i];
  mt3fe::LaneConst lc;
  mt3fe::load_lane_const(lc, lane, t.hann, t.tw1024, t.tw2048);
  __syncthreads();

  cpx* xchg = s_xchg[wave];
  float* mag = s_mag[wave];
  for (int r = 0; r < kFramesPerWave; ++r) {
    const int fl = wave + kWaves * r;                            // frame inside the tile
    mt3fe::stage_a(lc, lane, s_samples + fl * kHop, mt3fe::kFft, xchg);
    wave_lds_sync();
    mt3fe::stage_b(lc, lane, xchg);
    wave_lds_sync();
    cpx z[16];
    mt3fe::stage_c(lane, xchg, z);
    wave_lds_sync();
    mt3fe::publish_z(lane, z, xchg);
    wave_lds_sync();
    mt3fe::untangle_mag(lc, lane, z, xchg, mag);
    wave_lds_sync();
    const int f = f0 + fl;
    float* dst = out + (static_cast<size_t>(seg) * frames_per_segment + f) * kMelBins;
    if (f < n) {                                                 // spectral_ops.safe_log(mel)
      mel_group<0>(s_k0, s_w, lane, mag, dst);
      mel_group<1>(s_k0, s_w, lane, mag, dst);
      mel_group<2>(s_k0, s_w, lane, mag, dst);
      mel_group<3>(s_k0, s_w, lane, mag, dst);
      mel_group<4>(s_k0, s_w, lane, mag, dst);
      mel_group<5>(s_k0, s_w, lane, mag, dst);
      mel_group<6>(s_k0, s_w, lane, mag, dst);
      mel_group<7>(s_k0, s_w, lane, mag, dst);
    } else {
#pragma unroll
      for (int i = 0; i < kMelBins / 64; ++i) dst[lane + 64 * i] = 0.f;
    }
    // the next round's stage_a overwrites xchg: every lane's zlin reads happened before the
    // barrier that precedes the mel step, and mag is only rewritten after the next 4 barriers.
  }
}

template <typename T>
int upload(const std::vector<T>& h, void** d) {
  MT3_HIP_CHECK(hipMalloc(d, h.size() * sizeof(T)));
  MT3_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return MT3_OK;
}

}  // namespace

struct mt3_frontend {
  mt3_frontend_config cfg;
  mt3fe::HostTables host;
  void* d_hann = nullptr;
  void* d_tw1024 = nullptr;
  void* d_tw2048 = nullptr;
  void* d_k0 = nullptr;
  void* d_cnt = nullptr;
  void* d_off = nullptr;
  void* d_w = nullptr;           // group-padded transposed band weights (mt3fe::kPaddedWeights floats)
  std::vector<float> wpad;
  std::mutex table_mutex;        // first-call table upload: calls may come from several host threads
  bool on_device = false;
};

extern "C" {

int mt3_frontend_create(const mt3_frontend_config* cfg, mt3_frontend** out) {
  if (!cfg || !out) return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_create: null argument");
  if (cfg->fft_size != mt3fe::kFft || cfg->hop_width != kHop || cfg->num_mel_bins != kMelBins)
    return mt3::fail(MT3_ERR_INVALID,
                     "mt3_frontend_create: this build supports fft_size=2048, hop_width=128, num_mel_bins=512 "
                     "(the only configuration the reference uses: spectrograms.py:23-29)");
  mt3_frontend* fe = new (std::nothrow) mt3_frontend();
  if (!fe) return mt3::fail(MT3_ERR_INVALID, "out of host memory");
  fe->cfg = *cfg;
  if (cfg->table_dtype != 0 && cfg->table_dtype != 1) {
    delete fe;
    return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_create: table_dtype must be 0 (float32, as TensorFlow builds them) or 1 (float64)");
  }
  fe->host = mt3fe::build_tables(cfg->sample_rate, cfg->fft_size, cfg->num_mel_bins, cfg->lo_hz, cfg->hi_hz,
                                 cfg->table_dtype == 0);
  if (static_cast<int>(fe->host.w.size()) > kMaxBandWeights) {
    delete fe;
    return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_create: mel band table too large for the kernel's LDS budget");
  }
  if (!mt3fe::bands_fit(fe->host)) {
    delete fe;
    return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_create: a mel band is longer than the kernel's unrolled bound "
                                      "(this build is specialised to 512 bins over 20 .. 7600 Hz at 16 kHz)");
  }
  fe->wpad = mt3fe::build_padded_weights(fe->host);
  *out = fe;
  return MT3_OK;
}

void mt3_frontend_destroy(mt3_frontend* fe) {
  if (!fe) return;
  void* ptrs[] = {fe->d_hann, fe->d_tw1024, fe->d_tw2048, fe->d_k0, fe->d_cnt, fe->d_off, fe->d_w};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete fe;
}

int mt3_frontend_mel_matrix(const mt3_frontend* fe, float* h_out, int64_t* nnz) {
  if (!fe) return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_mel_matrix: null frontend");
  if (h_out) std::memcpy(h_out, fe->host.mel_dense.data(), fe->host.mel_dense.size() * sizeof(float));
  if (nnz) *nnz = fe->host.nnz;
  return MT3_OK;
}

static int upload_all(mt3_frontend* fe) {
  int rc;
  if ((rc = upload(fe->host.hann, &fe->d_hann))) return rc;
  if ((rc = upload(fe->host.tw1024, &fe->d_tw1024))) return rc;
  if ((rc = upload(fe->host.tw2048, &fe->d_tw2048))) return rc;
  if ((rc = upload(fe->host.k0, &fe->d_k0))) return rc;
  if ((rc = upload(fe->host.cnt, &fe->d_cnt))) return rc;
  if ((rc = upload(fe->host.off, &fe->d_off))) return rc;
  if ((rc = upload(fe->wpad, &fe->d_w))) return rc;
  return MT3_OK;
}

static int ensure_device_tables(mt3_frontend* fe) {
  std::lock_guard<std::mutex> lock(fe->table_mutex);
  if (fe->on_device) return MT3_OK;
  const int rc = upload_all(fe);
  if (rc != MT3_OK) {
    // a failed first call leaves nothing behind: the next call starts from scratch instead of leaking the tables
    // that did get uploaded (the error message of the failing step is kept)
    void** ptrs[] = {&fe->d_hann, &fe->d_tw1024, &fe->d_tw2048, &fe->d_k0, &fe->d_cnt, &fe->d_off, &fe->d_w};
    for (void** p : ptrs) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    return rc;
  }
  fe->on_device = true;
  return MT3_OK;
}

// h_n != nullptr: host-resident counts of exactly these n_segments (<= kCountsPerLaunch), passed by value
static int launch_logmel(mt3_frontend* fe, const float* d_audio, int32_t n_segments, int32_t frames_per_segment,
                         const int* d_n, float* d_logmel, hipStream_t s, const int32_t* h_n = nullptr) {
  FrontendDev t{static_cast<const float*>(fe->d_hann), static_cast<const cpx*>(fe->d_tw1024),
                static_cast<const cpx*>(fe->d_tw2048), static_cast<const int*>(fe->d_k0),
                static_cast<const int*>(fe->d_cnt),    static_cast<const int*>(fe->d_off),
                static_cast<const float*>(fe->d_w), static_cast<int>(fe->host.w.size())};
  // 16-frame tiles, four waves (r3 also measured 32-frame tiles on eight waves: HBM traffic 1.178 -> 1.091 x the
  // algorithmic bytes, but 17-19 % slower -- one 135 KB workgroup per CU)
  FrameCounts fc;
  if (h_n)
    for (int i = 0; i < n_segments; ++i) fc.n[i] = static_cast<uint16_t>(h_n[i]);
  else
    fc.n[0] = 0;                       // (unused; the rest of the argument is never read)
  hipLaunchKernelGGL(logmel_kernel<4>, dim3(n_segments * (frames_per_segment / 16)), dim3(256), 0, s, t, d_audio, d_n,
                     frames_per_segment, d_logmel, h_n ? 1 : 0, fc);
  MT3_HIP_CHECK(hipGetLastError());
  return MT3_OK;
}

static int check_logmel_args(mt3_frontend* fe, const float* d_audio, int32_t frames_per_segment, float* d_logmel) {
  if (!fe || !d_audio || !d_logmel) return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_logmel: null argument");
  if (frames_per_segment <= 0 || frames_per_segment % kFramesPerBlock != 0)
    return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_logmel: frames_per_segment must be a positive multiple of 16");
  return MT3_OK;
}

int mt3_frontend_logmel(mt3_frontend* fe, const float* d_audio, int32_t n_segments, int32_t frames_per_segment,
                        const int32_t* h_n_frames, float* d_logmel, void* stream) {
  int rc = check_logmel_args(fe, d_audio, frames_per_segment, d_logmel);
  if (rc) return rc;
  if (n_segments <= 0) return MT3_OK;
  if ((rc = ensure_device_tables(fe))) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!h_n_frames) return launch_logmel(fe, d_audio, n_segments, frames_per_segment, nullptr, d_logmel, s);
  if (frames_per_segment > 65535) return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_logmel: frames_per_segment > 65535");
  for (int i = 0; i < n_segments; ++i)
    if (h_n_frames[i] < 0 || h_n_frames[i] > frames_per_segment)
      return mt3::fail(MT3_ERR_INVALID, "mt3_frontend_logmel: n_frames out of range");
  // the counts ride in the launches themselves, kCountsPerLaunch segments apiece: the caller's buffer is free when the
  // call returns, nothing is copied, allocated, locked or waited for
  const size_t seg_in = static_cast<size_t>(frames_per_segment) * kHop, seg_out = static_cast<size_t>(frames_per_segment) * kMelBins;
  for (int s0 = 0; s0 < n_segments; s0 += kCountsPerLaunch) {
    const int n = n_segments - s0 < kCountsPerLaunch ? n_segments - s0 : kCountsPerLaunch;
    rc = launch_logmel(fe, d_audio + s0 * seg_in, n, frames_per_segment, nullptr, d_logmel + s0 * seg_out, s, h_n_frames + s0);
    if (rc) return rc;
  }
  return MT3_OK;
}

int mt3_frontend_logmel_dev(mt3_frontend* fe, const float* d_audio, int32_t n_segments, int32_t frames_per_segment,
                            const int32_t* d_n_frames, float* d_logmel, void* stream) {
  int rc = check_logmel_args(fe, d_audio, frames_per_segment, d_logmel);
  if (rc) return rc;
  if (n_segments <= 0) return MT3_OK;
  if ((rc = ensure_device_tables(fe))) return rc;
  return launch_logmel(fe, d_audio, n_segments, frames_per_segment, d_n_frames, d_logmel,
                       static_cast<hipStream_t>(stream));
}

}  // extern "C"

// Host-side construction of the frontend's constant tables (window, twiddles,
// HTK mel filterbank in dense and band-sparse form).  Plain C++ (no HIP), shared
// by frontend.hip and the CPU emulation test.
//
// Mel matrix: tf.signal.linear_to_mel_weight_matrix(512, 1025, 16000, 20.0, 7600.0)
// as called at mt3/spectral_ops.py:69-70 [TF op restated from its documentation]:
// HTK mel m(f) = 1127 ln(1 + f/700); 514 band edges linspace'd in mel between
// m(lo) and m(hi); weight[k][j] = max(0, min(lower slope, upper slope)) for linear
// bins k >= 1, DC row zero.
//
// Table arithmetic (round 5).  TensorFlow builds both the Hann window and the mel matrix in FLOAT32 -- tf.signal.stft's
// window_fn and linear_to_mel_weight_matrix default to dtype=tf.float32 and the reference passes none
// (mt3/spectral_ops.py:42-47,69-71) -- so `tf32 = true` (the default of mt3_frontend_config.table_dtype) evaluates them
// in float32 in TF's op order [restated from the TF sources from memory; the test oracle's mel_weight_matrix_tf32 /
// hann_periodic_tf32 are the same rule in numpy]: linspace as start + delta * i, _hertz_to_mel as 1127 * log(1 + f / 700) with a plain log,
// the slopes as f32 quotients, the window as 0.5 - 0.5 * cos(2 pi f32 * k / N) -- every +, -, *, / one IEEE float32
// operation, and the two transcendental functions CORRECTLY ROUNDED to float32 (evaluated in double, rounded once).  That
// last choice is deliberate: float32 `log` is not correctly rounded in any of the libraries at hand (numpy's differs
// from the correctly rounded value at 10 % of its arguments, glibc's logf and Eigen's plog -- what TensorFlow runs -- at
// others), one ulp of a mel value near 2800 is 4.5e-5 of a triangle's width, and so three float32 evaluations of this
// very formula (numpy's log, glibc's, the correctly rounded one) land up to 9.1e-5 apart in a weight -- MORE than the
// 6.8e-5 between any of them and the float64 evaluation.  TensorFlow's own table is one more point of that cloud; the
// correctly rounded log is its centre, needs no particular libm, and is what the oracle's tf32 restatement uses too, so
// product and oracle tables are bit-identical (tests/test_frontend_emulation.py, no GPU needed).  `tf32 = false` keeps
// rounds 1-4's evaluation in double (rounded to float at the end).  The FFT twiddles are the kernel's own business
// (TF's FFT is a library call) and stay double-evaluated.
#ifndef MT3_FRONTEND_TABLES_H_
#define MT3_FRONTEND_TABLES_H_

#include <cmath>
#include <cstdint>
#include <vector>

#include "frontend_core.h"

namespace mt3fe {

struct HostTables {
  int fft = 2048, bins = 1025, mel = 512;
  std::vector<float> hann;        // [fft] periodic Hann
  std::vector<float> tw1024;      // [1024][2]  exp(-2 pi i j/1024)
  std::vector<float> tw2048;      // [1025][2]  exp(-2 pi i j/2048)
  std::vector<float> mel_dense;   // [bins][mel]
  std::vector<int32_t> k0, cnt, off;   // per mel bin: first spectrum bin, count, offset into w
  std::vector<float> w;           // band weights / 2 (exact), concatenated per mel bin: the kernel's magnitudes are 2|X|
  int64_t nnz = 0;
  int max_cnt = 0;
};

inline double hz_to_mel(double f) { return 1127.0 * std::log1p(f / 700.0); }

// float32, one rounding per operation, no contraction (the translation units that include this header are built with
// -ffp-contract=off or carry the pragma below)
inline float log_f32_cr(float x) { return static_cast<float>(std::log(static_cast<double>(x))); }   // correctly rounded
inline float hz_to_mel_f32(float f) { return 1127.0f * log_f32_cr(1.0f + f / 700.0f); }
inline std::vector<float> linspace_f32(float start, float stop, int n) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  std::vector<float> v(n);
  const float delta = (stop - start) / static_cast<float>(n - 1);
  for (int i = 0; i < n; ++i) v[i] = start + delta * static_cast<float>(i);
  v[0] = start;
  v[n - 1] = stop;
  return v;
}

inline void tf32_window_and_mel(HostTables& t, int sample_rate, float lo_hz, float hi_hz) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float two_pi = static_cast<float>(2.0 * 3.14159265358979323846);
  for (int i = 0; i < t.fft; ++i) {
    const float arg = two_pi * static_cast<float>(i) / static_cast<float>(t.fft);
    t.hann[i] = 0.5f - 0.5f * static_cast<float>(std::cos(static_cast<double>(arg)));     // cos correctly rounded
  }
  const float nyquist = static_cast<float>(sample_rate) / 2.0f;
  const std::vector<float> lin = linspace_f32(0.0f, nyquist, t.bins);
  const std::vector<float> edges = linspace_f32(hz_to_mel_f32(lo_hz), hz_to_mel_f32(hi_hz), t.mel + 2);
  for (int k = 1; k < t.bins; ++k) {
    const float m = hz_to_mel_f32(lin[k]);
    for (int j = 0; j < t.mel; ++j) {
      const float lower = (m - edges[j]) / (edges[j + 1] - edges[j]);
      const float upper = (edges[j + 2] - m) / (edges[j + 2] - edges[j + 1]);
      t.mel_dense[static_cast<size_t>(k) * t.mel + j] = std::fmax(0.0f, std::fmin(lower, upper));
    }
  }
}

inline HostTables build_tables(int sample_rate, int fft, int mel_bins, double lo_hz, double hi_hz, bool tf32) {
  HostTables t;
  t.fft = fft;
  t.bins = fft / 2 + 1;
  t.mel = mel_bins;
  const double pi = 3.14159265358979323846;
  t.hann.resize(fft);
  for (int i = 0; i < fft; ++i) t.hann[i] = static_cast<float>(0.5 - 0.5 * std::cos(2.0 * pi * i / fft));
  const int half = fft / 2;
  t.tw1024.resize(2 * half);
  for (int j = 0; j < half; ++j) {
    t.tw1024[2 * j] = static_cast<float>(std::cos(2.0 * pi * j / half));
    t.tw1024[2 * j + 1] = static_cast<float>(-std::sin(2.0 * pi * j / half));
  }
  t.tw2048.resize(2 * (half + 1));
  for (int j = 0; j <= half; ++j) {
    t.tw2048[2 * j] = static_cast<float>(std::cos(2.0 * pi * j / fft));
    t.tw2048[2 * j + 1] = static_cast<float>(-std::sin(2.0 * pi * j / fft));
  }
  // dense mel matrix
  const double nyquist = sample_rate / 2.0;
  std::vector<double> edges(mel_bins + 2);
  const double m_lo = hz_to_mel(lo_hz), m_hi = hz_to_mel(hi_hz);
  for (int i = 0; i < mel_bins + 2; ++i) edges[i] = m_lo + (m_hi - m_lo) * i / (mel_bins + 1);
  t.mel_dense.assign(static_cast<size_t>(t.bins) * mel_bins, 0.f);
  for (int k = 1; k < t.bins; ++k) {
    const double f = nyquist * k / (t.bins - 1);
    const double m = hz_to_mel(f);
    for (int j = 0; j < mel_bins; ++j) {
      const double lower = (m - edges[j]) / (edges[j + 1] - edges[j]);
      const double upper = (edges[j + 2] - m) / (edges[j + 2] - edges[j + 1]);
      const double v = std::fmax(0.0, std::fmin(lower, upper));
      t.mel_dense[static_cast<size_t>(k) * mel_bins + j] = static_cast<float>(v);
    }
  }
  if (tf32) tf32_window_and_mel(t, sample_rate, static_cast<float>(lo_hz), static_cast<float>(hi_hz));
  // band-sparse form: for each mel bin the contiguous run of spectrum bins with weight > 0
  t.k0.assign(mel_bins, 0);
  t.cnt.assign(mel_bins, 0);
  t.off.assign(mel_bins, 0);
  for (int j = 0; j < mel_bins; ++j) {
    int first = -1, last = -1;
    for (int k = 0; k < t.bins; ++k)
      if (t.mel_dense[static_cast<size_t>(k) * mel_bins + j] != 0.f) {
        if (first < 0) first = k;
        last = k;
      }
    t.off[j] = static_cast<int32_t>(t.w.size());
    if (first >= 0) {
      t.k0[j] = first;
      t.cnt[j] = last - first + 1;
      for (int k = first; k <= last; ++k) {
        const float v = t.mel_dense[static_cast<size_t>(k) * mel_bins + j];
        t.w.push_back(0.5f * v);
        if (v != 0.f) ++t.nnz;
      }
      if (t.cnt[j] > t.max_cnt) t.max_cnt = t.cnt[j];
    }
  }
  return t;
}

// the kernel's form of the band weights (frontend_core.h: mel_bin_padded)
inline bool bands_fit(const HostTables& t) {
  if (t.mel != 512) return false;
  for (int j = 0; j < t.mel; ++j)
    if (t.cnt[j] > kGroupMaxBand[j / 64]) return false;
  return true;
}
inline std::vector<float> build_padded_weights(const HostTables& t) {
  std::vector<float> wpad(kPaddedWeights, 0.f);
  for (int j = 0; j < t.mel; ++j)
    for (int q = 0; q < t.cnt[j]; ++q) wpad[group_base(j / 64) + q * 64 + (j & 63)] = t.w[t.off[j] + q];
  return wpad;
}

}  // namespace mt3fe
#endif  // MT3_FRONTEND_TABLES_H_

// CPU emulation of mt3_amd/csrc/frontend.hip's logmel_kernel: the SAME per-lane
// functions (frontend_core.h) and the SAME tables (frontend_tables.h), with the
// wave's 64 lanes run in a loop between "barriers" and plain arrays as LDS.
// Built with g++ by tests/test_frontend_emulation.py; verifies the FFT index
// algebra, the untangle and the band-sparse mel against the numpy oracle
// without a GPU.  Test infrastructure only.
#include <cmath>
#include <cstring>
#include <vector>

#include "frontend_core.h"
#include "frontend_tables.h"

using mt3fe::cpx;

// mel_bin_padded<MAXC> of group i (the kernel unrolls the groups at compile time)
static float padded_bin(int i, const int* k0, const float* wp, int j, const float* mag) {
  using namespace mt3fe;
  switch (i) {
    case 0: return mel_bin_padded<kGroupMaxBand[0]>(k0, wp + group_base(0), j, mag);
    case 1: return mel_bin_padded<kGroupMaxBand[1]>(k0, wp + group_base(1), j, mag);
    case 2: return mel_bin_padded<kGroupMaxBand[2]>(k0, wp + group_base(2), j, mag);
    case 3: return mel_bin_padded<kGroupMaxBand[3]>(k0, wp + group_base(3), j, mag);
    case 4: return mel_bin_padded<kGroupMaxBand[4]>(k0, wp + group_base(4), j, mag);
    case 5: return mel_bin_padded<kGroupMaxBand[5]>(k0, wp + group_base(5), j, mag);
    case 6: return mel_bin_padded<kGroupMaxBand[6]>(k0, wp + group_base(6), j, mag);
    default: return mel_bin_padded<kGroupMaxBand[7]>(k0, wp + group_base(7), j, mag);
  }
}

// tf32: the tables as mt3_frontend_config.table_dtype = 0 builds them (float32 in TensorFlow's op order); else float64
static const mt3fe::HostTables& tables(int tf32) {
  static const mt3fe::HostTables T64 = mt3fe::build_tables(16000, 2048, 512, 20.0, 7600.0, false);
  static const mt3fe::HostTables T32 = mt3fe::build_tables(16000, 2048, 512, 20.0, 7600.0, true);
  return tf32 ? T32 : T64;
}

extern "C" int emul_logmel(const float* audio, int n_frames, int frames_per_segment, float* out, int tf32) {
  const mt3fe::HostTables& T = tables(tf32);
  const int hop = 128, G = 16, tile = G * hop + 1920;
  const int valid = n_frames * hop;
  const std::vector<float> WP = mt3fe::build_padded_weights(T);     // the kernel's group-padded table
  if (!mt3fe::bands_fit(T)) return 1;
  std::vector<mt3fe::LaneConst> lc(64);
  for (int l = 0; l < 64; ++l)
    mt3fe::load_lane_const(lc[l], l, T.hann.data(), reinterpret_cast<const cpx*>(T.tw1024.data()),
                           reinterpret_cast<const cpx*>(T.tw2048.data()));
  std::vector<float> samples(tile);
  std::vector<cpx> xchg(mt3fe::kXchg);
  std::vector<float> mag(1028);
  for (int f0 = 0; f0 < frames_per_segment; f0 += G) {
    for (int i = 0; i < tile; ++i) {
      const int idx = f0 * hop + i;
      samples[i] = idx < valid ? audio[idx] : 0.f;
    }
    for (int fl = 0; fl < G; ++fl) {
      cpx z[64][16];
      for (int l = 0; l < 64; ++l) mt3fe::stage_a(lc[l], l, samples.data() + fl * hop, 2048, xchg.data());
      for (int l = 0; l < 64; ++l) mt3fe::stage_b(lc[l], l, xchg.data());
      for (int l = 0; l < 64; ++l) mt3fe::stage_c(l, xchg.data(), z[l]);
      for (int l = 0; l < 64; ++l) mt3fe::publish_z(l, z[l], xchg.data());
      for (int l = 0; l < 64; ++l) mt3fe::untangle_mag(lc[l], l, z[l], xchg.data(), mag.data());
      const int f = f0 + fl;
      float* dst = out + static_cast<size_t>(f) * 512;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 8; ++i) {
          const int j = l + 64 * i;
          if (f < n_frames) {
            const float m = padded_bin(i, T.k0.data(), WP.data(), j, mag.data());
            dst[j] = logf(m <= 0.f ? 1e-5f : m);
          } else {
            dst[j] = 0.f;
          }
        }
    }
  }
  return 0;
}

extern "C" int emul_mel_dense(float* out, int tf32) {
  const mt3fe::HostTables& T = tables(tf32);
  std::memcpy(out, T.mel_dense.data(), T.mel_dense.size() * sizeof(float));
  return static_cast<int>(T.nnz);
}

extern "C" int emul_hann(float* out, int tf32) {
  const mt3fe::HostTables& T = tables(tf32);
  std::memcpy(out, T.hann.data(), T.hann.size() * sizeof(float));
  return static_cast<int>(T.hann.size());
}

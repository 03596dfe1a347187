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

extern "C" int emul_logmel(const float* audio, int n_frames, int frames_per_segment, float* out) {
  static const mt3fe::HostTables T = mt3fe::build_tables(16000, 2048, 512, 20.0, 7600.0);
  const int hop = 128, G = 16, tile = G * hop + 1920;
  const int valid = n_frames * hop;
  const mt3fe::MelTables mel{T.k0.data(), T.cnt.data(), T.off.data(), T.w.data()};
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
            const float m = mt3fe::mel_bin(mel, j, mag.data());
            dst[j] = logf(m <= 0.f ? 1e-5f : m);
          } else {
            dst[j] = 0.f;
          }
        }
    }
  }
  return 0;
}

extern "C" int emul_mel_dense(float* out) {
  static const mt3fe::HostTables T = mt3fe::build_tables(16000, 2048, 512, 20.0, 7600.0);
  std::memcpy(out, T.mel_dense.data(), T.mel_dense.size() * sizeof(float));
  return static_cast<int>(T.nnz);
}

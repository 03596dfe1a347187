// Per-lane building blocks of the fused log-mel frontend kernel (frontend.hip).
//
// Everything here is __host__ __device__ and free of wave intrinsics: the HIP
// kernel calls these once per lane between wave-level barriers, and
// tests/host/frontend_emul.cpp calls the very same functions in a loop over the
// 64 "lanes" of a wave with a plain array standing in for LDS, so the FFT index
// algebra is verified on the CPU build box (no GPU there).
//
// Math (reference: mt3/spectral_ops.py:35-54 `stft`/`compute_mag`; tf.signal.stft
// with frame_length = fft_length = 2048, periodic Hann):
//   real FFT-2048 of the windowed frame via ONE complex FFT-1024 of
//   z[m] = x[2m] + i x[2m+1], then the even/odd untangle.
//   FFT-1024 = radix 16 x 4 x 16 over a wave of 64 lanes, 16 points per lane:
//     stage A: lane t holds z[t + 64a], a=0..15   -> DFT16 over a, twiddle W1024^(t*ka)
//     stage B: (ka, c): 4 points t = c + 16b      -> DFT4 over b,  twiddle W64^(c*kb)
//     stage C: lane (ka = l%16, kb = l/16): c=0..15 -> DFT16 over c
//   output: lane l holds Z[l + 64*kc], kc = 0..15  (same striding as the input).
//   "LDS" exchange buffer: 16 rows (ka) of 64 complex, row stride kRowStride = 65.
#ifndef MT3_FRONTEND_CORE_H_
#define MT3_FRONTEND_CORE_H_

#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MT3_HD __host__ __device__ __forceinline__
#define MT3_UNROLL _Pragma("unroll")
#else
#define MT3_HD inline
#define MT3_UNROLL
#endif

// device build: the raw v_sqrt_f32 (1 ulp) instead of sqrtf's ~10-instruction denormal-safe expansion -- 16 per frame
// and lane in a kernel that is VALU-issue-bound; the host emulation keeps libm
#if defined(__HIP_DEVICE_COMPILE__)
#define MT3_FE_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define MT3_FE_SQRT(x) sqrtf(x)
#endif

namespace mt3fe {

constexpr int kFft = 2048;
constexpr int kHalf = 1024;        // complex FFT size
constexpr int kBins = 1025;        // rfft bins
constexpr int kRowStride = 65;     // complex elements per ka row in the exchange buffer
constexpr int kXchg = 16 * kRowStride;

struct cpx {
  float re, im;
};
MT3_HD cpx operator+(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
MT3_HD cpx operator-(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
#if defined(__HIP_DEVICE_COMPILE__)
// device: two packed f32 instructions (v_pk_mul_f32 + v_pk_fma_f32, half-swaps and the sign riding on op_sel / neg)
// instead of two multiplies and two FMAs
typedef float mt3_f2 __attribute__((ext_vector_type(2)));
MT3_HD cpx cmul(cpx a, cpx b) {
  const mt3_f2 av = {a.re, a.im}, as = {a.im, a.re};
  const mt3_f2 br = {b.re, b.re}, bi = {-b.im, b.im};
  const mt3_f2 r = __builtin_elementwise_fma(as, bi, av * br);
  return {r.x, r.y};
}
#else
MT3_HD cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
#endif
MT3_HD cpx mul_neg_i(cpx a) { return {a.im, -a.re}; }   // a * (-i)
MT3_HD cpx mul_pos_i(cpx a) { return {-a.im, a.re}; }   // a * (+i)

// forward 4-point DFT (kernel e^{-2 pi i nk/4}), in place
MT3_HD void dft4(cpx& x0, cpx& x1, cpx& x2, cpx& x3) {
  const cpx s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = x1 - x3;
  x0 = s02 + s13;
  x1 = d02 + mul_neg_i(d13);
  x2 = s02 - s13;
  x3 = d02 + mul_pos_i(d13);
}

// forward 16-point DFT in place: v[n] -> V[k], natural order in and out.
// n = 4*n1 + n2, k = k1 + 4*k2:  V[k1+4k2] = sum_n2 W4^(n2 k2) W16^(n2 k1) sum_n1 v[4n1+n2] W4^(n1 k1)
MT3_HD void dft16(cpx (&v)[16]) {
  // W16^m = exp(-2 pi i m / 16), m = 0..9
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r = 0.70710678118654752f;
  const cpx w[10] = {{1.f, 0.f}, {c1, -s1}, {r, -r},  {s1, -c1}, {0.f, -1.f},
                     {-s1, -c1}, {-r, -r},  {-c1, -s1}, {-1.f, 0.f}, {-c1, s1}};
  // step 1: for each n2, DFT4 over n1 of v[4*n1 + n2] -> t[n2][k1] stored back at v[4*k1 + n2]
MT3_UNROLL
  for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  // step 2: twiddle t[n2][k1] *= W16^(n2*k1)
MT3_UNROLL
  for (int k1 = 1; k1 < 4; ++k1) {
MT3_UNROLL
    for (int n2 = 1; n2 < 4; ++n2) v[4 * k1 + n2] = cmul(v[4 * k1 + n2], w[n2 * k1]);
  }
  // step 3: for each k1, DFT4 over n2 of v[4*k1 + n2] -> V[k1 + 4*k2] (stored at v[4*k1 + k2])
MT3_UNROLL
  for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  // reorder: out[k1 + 4*k2] = v[4*k1 + k2]  (a 4x4 transpose)
  cpx o[16];
MT3_UNROLL
  for (int k1 = 0; k1 < 4; ++k1) {
MT3_UNROLL
    for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
  }
MT3_UNROLL
  for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// Per-lane constants, loaded once per kernel (they do not depend on the frame).
struct LaneConst {
  float win[32];    // Hann at samples 2*(lane+64a) (even index) and +1: win[2a], win[2a+1]
  cpx twA[16];      // W1024^(lane*ka)
  cpx twB[4];       // W64^((lane%16)*kb)
  cpx twU[16];      // W2048^(lane + 64*m)   (untangle)
};

// tw1024[j] = exp(-2 pi i j / 1024), j in [0,1024); tw2048[j] = exp(-2 pi i j / 2048), j in [0,1024]
MT3_HD void load_lane_const(LaneConst& c, int lane, const float* hann, const cpx* tw1024, const cpx* tw2048) {
  MT3_UNROLL
  for (int a = 0; a < 16; ++a) {
    const int m = lane + 64 * a;
    c.win[2 * a] = hann[2 * m];
    c.win[2 * a + 1] = hann[2 * m + 1];
    c.twA[a] = tw1024[(lane * a) & 1023];
    c.twU[a] = tw2048[lane + 64 * a];
  }
  MT3_UNROLL
  for (int kb = 0; kb < 4; ++kb) c.twB[kb] = tw1024[(16 * (lane & 15) * kb) & 1023];
}

// ---- stage A: `frame` points at the frame's first sample inside the staged audio tile
// (at least `valid` samples are real; samples at index >= valid read as zero -- this is the
// pad_end=True zero padding of tf.signal.stft relative to the END OF THE SEGMENT).
MT3_HD void stage_a(const LaneConst& c, int lane, const float* frame, int valid, cpx* xchg) {
  cpx v[16];
  MT3_UNROLL
  for (int a = 0; a < 16; ++a) {
    const int m = lane + 64 * a;
    const float xe = (2 * m < valid) ? frame[2 * m] : 0.f;
    const float xo = (2 * m + 1 < valid) ? frame[2 * m + 1] : 0.f;
    v[a] = {xe * c.win[2 * a], xo * c.win[2 * a + 1]};
  }
  dft16(v);
  MT3_UNROLL
  for (int ka = 0; ka < 16; ++ka) xchg[ka * kRowStride + lane] = cmul(v[ka], c.twA[ka]);
}

// ---- stage B: lane -> c = lane%16, g = lane/16; butterflies for ka = g + 4i, i = 0..3 (in place)
MT3_HD void stage_b(const LaneConst& c, int lane, cpx* xchg) {
  const int cc = lane & 15, g = lane >> 4;
  MT3_UNROLL
  for (int i = 0; i < 4; ++i) {
    cpx* row = xchg + (g + 4 * i) * kRowStride + cc;
    cpx x0 = row[0], x1 = row[16], x2 = row[32], x3 = row[48];
    dft4(x0, x1, x2, x3);
    row[0] = x0;                       // kb = 0: twiddle 1
    row[16] = cmul(x1, c.twB[1]);
    row[32] = cmul(x2, c.twB[2]);
    row[48] = cmul(x3, c.twB[3]);
  }
}

// ---- stage C: lane -> ka = lane%16, kb = lane/16; z[kc] = Z[lane + 64*kc]
MT3_HD void stage_c(int lane, const cpx* xchg, cpx (&z)[16]) {
  const cpx* p = xchg + (lane & 15) * kRowStride + (lane >> 4) * 16;
  MT3_UNROLL
  for (int i = 0; i < 16; ++i) z[i] = p[i];
  dft16(z);
}

// ---- publish Z in natural order so that every lane can fetch its mirror bin
MT3_HD void publish_z(int lane, const cpx (&z)[16], cpx* zlin /*[1024]*/) {
  MT3_UNROLL
  for (int m = 0; m < 16; ++m) zlin[lane + 64 * m] = z[m];
}

// ---- untangle: 2 |X[k]| for k = lane + 64*m (m = 0..15), plus k = 1024 on lane 0.
// X[k] = E + W2048^k O with E = (a + b')/2, O = -i (a - b')/2, a = Z[k], b' = conj(Z[N/2 - k]).  The two halvings
// are exact scalings by a power of two, so they are left out here (x2 = 2 X[k] exactly, sqrt(4 y) = 2 sqrt(y)
// exactly) and the band weights of the mel tables carry the factor 1/2 instead (HostTables::w is stored halved):
// 4 multiplies per bin less, bit-identical mel sums.
MT3_HD void untangle_mag(const LaneConst& c, int lane, const cpx (&z)[16], const cpx* zlin, float* mag /*[1025]*/) {
  MT3_UNROLL
  for (int m = 0; m < 16; ++m) {
    const int k = lane + 64 * m;
    const cpx a = z[m];
    const cpx zb = zlin[(kHalf - k) & (kHalf - 1)];
    const cpx b = {zb.re, -zb.im};                       // conj(Z[N/2 - k])
    const cpx e2 = a + b;
    const cpx d = a - b;
    const cpx o2 = {d.im, -d.re};                        // -i (a - b)
    const cpx x2 = e2 + cmul(c.twU[m], o2);
    mag[k] = MT3_FE_SQRT(x2.re * x2.re + x2.im * x2.im);
  }
  if (lane == 0) {
    const float v = 2.f * (z[0].re - z[0].im);           // X[1024] = Re Z0 - Im Z0
    mag[kHalf] = v < 0.f ? -v : v;
  }
}

// Sparse mel projection tables: mel bin j sums mag[k0[j] .. k0[j]+cnt[j]) * w[off[j] + i].
struct MelTables {
  const int* k0;
  const int* cnt;
  const int* off;
  const float* w;
};

MT3_HD float mel_bin(const MelTables& t, int j, const float* mag) {
  float acc = 0.f;
  const int n = t.cnt[j];
  const float* w = t.w + t.off[j];
  const float* m = mag + t.k0[j];
  for (int i = 0; i < n; ++i) acc += m[i] * w[i];
  return acc;
}

// The same sum with a COMPILE-TIME trip count MAXC >= cnt[j] (the longest band of the bin's group of 64): branch-free,
// immediate LDS offsets, terms past the band get weight 0 (fma(m, 0, acc) == acc, so the result is bit-identical to
// mel_bin; reads past the band stay inside the mag / weight arrays: k0 + 10 <= 983 < 1028).
template <int MAXC>
MT3_HD float mel_bin_fixed(const MelTables& t, int j, const float* mag) {
  float acc = 0.f;
  const int n = t.cnt[j];
  const float* w = t.w + t.off[j];
  const float* m = mag + t.k0[j];
  float wv[MAXC], mv[MAXC];
  MT3_UNROLL
  for (int i = 0; i < MAXC; ++i) {      // unconditional loads first (a "load or zero" select would become a branch
    wv[i] = w[i];                       // around every load on the device)
    mv[i] = m[i];
  }
  MT3_UNROLL
  for (int i = 0; i < MAXC; ++i) acc += mv[i] * (i < n ? wv[i] : 0.f);
  return acc;
}

// longest band (spectrum bins per mel bin) inside each group of 64 mel bins, for the reference's mel matrix
// (512 bins, 20 .. 7600 Hz over 1025 FFT bins); mt3_frontend_create checks the tables against it
constexpr int kGroupMaxBand[8] = {2, 2, 3, 3, 4, 6, 8, 10};
// offset of group I inside the padded weight table: 64 * (sum of the bounds of the groups before it)
constexpr int group_base(int I) {
  int b = 0;
  for (int i = 0; i < I; ++i) b += 64 * kGroupMaxBand[i];
  return b;
}
constexpr int kPaddedWeights = group_base(8);   // 64 * 38 = 2432 floats

// Group-padded form of the same tables (what the kernel keeps in LDS): the bands of the 64 mel bins of group i are
// zero-padded to the group's longest band MAXC and stored TRANSPOSED, wpad[base_i + q * 64 + (j & 63)] = weight q of
// bin j -- no per-bin count / offset, no select per term, conflict-free lane-consecutive reads; adding the zero
// terms leaves the sum bit-identical to mel_bin (fma(m, 0, acc) == acc).
template <int MAXC>
MT3_HD float mel_bin_padded(const int* k0, const float* wpad_group, int j, const float* mag) {
  const float* m = mag + k0[j];
  const float* w = wpad_group + (j & 63);
  float wv[MAXC], mv[MAXC];
  MT3_UNROLL
  for (int i = 0; i < MAXC; ++i) {
    wv[i] = w[i * 64];
    mv[i] = m[i];
  }
  float acc = 0.f;
  MT3_UNROLL
  for (int i = 0; i < MAXC; ++i) acc += mv[i] * wv[i];
  return acc;
}

}  // namespace mt3fe
#endif  // MT3_FRONTEND_CORE_H_

// Launch-shape knobs behind mt3_debug_set_knob (include/mt3_hip_debug.h; defined in errors.cpp).
// 0 = the measured default.  The product never reads an environment variable.
#ifndef MT3_KNOBS_H_
#define MT3_KNOBS_H_

namespace mt3k {

struct Knobs {
  int dec_attn_waves;       // waves per decode-attention workgroup (2 / 3 / 4)
  int dec_attn_fp8_waves;
  int no_lds_dma_gemm;      // encoder GEMMs on the register-staged tile
  int f32_split_k;          // decode-sized f32 GEMM tiles with EIGHT waves, K-groups split two ways (measured slower: off)
  int xcd_n_major;          // decode-sized GEMM tiles dealt to the XCDs by weight-column slice: 0 = when the weight
                            // matrix exceeds 3 MB (more than an XCD's L2 keeps next to everything else), 1 = always, 2 = never
  int no_k768_split;        // K = 768 decode tiles always as one slice (one workgroup per CU), however many workgroups
  int no_glds_256;          // encoder GEMMs always on the 128 x 128 LDS-DMA tile (never the 256 x 128 one)
  int fold_wide_tile;       // the two-source fold launch on 32 x 64 tiles instead of 32 x 32 (measured slower: off)
  int frontend_32_frame_tiles;   // log-mel kernel on 32-frame tiles (512 threads; less halo traffic, measured slower: off)
  int enc_attn_4_waves;     // encoder attention (bf16, T = 256) with four waves per workgroup instead of eight
  int glds_frag_db;         // encoder GEMMs on the 128-row LDS-DMA tile with fragment double-buffering (4-stage ring)
  int geglu_narrow_tile;    // decode GEGLU launch on 32 x 32 two-wave tiles (two per CU) instead of 32 x 64
  int prefetch2;            // decode-sized multi-slice tiles with TWO K slices in flight (measured slower: off)
};
extern Knobs g_knobs;

}  // namespace mt3k
#endif  // MT3_KNOBS_H_

// Thread-local error string behind mt3_last_error().
#include <string>

#include "common.h"
#include "knobs.h"
#include "mt3_hip_debug.h"

namespace {
thread_local std::string g_last_error;
}

namespace mt3 {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace mt3

namespace mt3k {
Knobs g_knobs = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}

extern "C" {
const char* mt3_last_error(void) { return g_last_error.c_str(); }
int mt3_abi_version(void) { return 2; }   // 2: mt3_engine_config.options; profiling flags moved to mt3_hip_debug.h

int mt3_debug_set_knob(int32_t knob, int32_t value) {
  switch (knob) {
    case MT3_DEBUG_KNOB_DEC_ATTN_WAVES:
    case MT3_DEBUG_KNOB_DEC_ATTN_FP8_WAVES:
      if (value != 0 && (value < 2 || value > 4)) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_set_knob: waves must be 0, 2, 3 or 4");
      (knob == MT3_DEBUG_KNOB_DEC_ATTN_WAVES ? mt3k::g_knobs.dec_attn_waves : mt3k::g_knobs.dec_attn_fp8_waves) = value;
      return MT3_OK;
    case MT3_DEBUG_KNOB_NO_LDS_DMA_GEMM:
      mt3k::g_knobs.no_lds_dma_gemm = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_F32_SPLIT_K:
      mt3k::g_knobs.f32_split_k = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_XCD_N_MAJOR:
      if (value < 0 || value > 2) return mt3::fail(MT3_ERR_INVALID, "mt3_debug_set_knob: XCD_N_MAJOR is 0 (auto), 1 or 2");
      mt3k::g_knobs.xcd_n_major = value;
      return MT3_OK;
    case MT3_DEBUG_KNOB_NO_K768_SPLIT:
      mt3k::g_knobs.no_k768_split = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_NO_GLDS_256:
      mt3k::g_knobs.no_glds_256 = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_FOLD_WIDE_TILE:
      mt3k::g_knobs.fold_wide_tile = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_FRONTEND_32_FRAME_TILES:
      mt3k::g_knobs.frontend_32_frame_tiles = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_ENC_ATTN_4_WAVES:
      mt3k::g_knobs.enc_attn_4_waves = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_GLDS_FRAG_DB:
      mt3k::g_knobs.glds_frag_db = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_GEGLU_NARROW_TILE:
      mt3k::g_knobs.geglu_narrow_tile = value != 0;
      return MT3_OK;
    case MT3_DEBUG_KNOB_PREFETCH2:
      mt3k::g_knobs.prefetch2 = value != 0;
      return MT3_OK;
    default:
      return mt3::fail(MT3_ERR_INVALID, "mt3_debug_set_knob: unknown knob");
  }
}
}
